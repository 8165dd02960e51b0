set -o pipefail
(time timeout -k 10 900 python bench.py --steps 20 --warmup 5) > gpurun_out/b_default.log 2>&1; echo "default rc=$?"
(time timeout -k 10 600 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 10 --warmup 2 --single) > gpurun_out/b_gpus2.log 2>&1; echo "gpus2 rc=$?"
(time timeout -k 10 600 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 6 --warmup 2 --single --row-sharded --workload zipf_v2m_d128 --batch-size 262144 --max-batches 4) > gpurun_out/b_gpus2_sharded.log 2>&1; echo "gpus2 sharded rc=$?"
(time timeout -k 10 600 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 6 --warmup 2 --single --workload zipf_v400k_d300 --batch-size 131072 --max-batches 4) > gpurun_out/b_gpus2_c4.log 2>&1; echo "gpus2 c4 rc=$?"
(time timeout -k 10 600 python bench.py --single --row-sharded --workload zipf_v2m_d128 --batch-size 1048576 --steps 20 --warmup 5 --max-batches 8 --no-cpu-baseline) > gpurun_out/b_c5_sharded1.log 2>&1; echo "c5 sharded rc=$?"
(time timeout -k 10 600 python bench.py --single --row-sharded --cols-replicated --workload zipf_v2m_d128 --batch-size 1048576 --steps 20 --warmup 5 --max-batches 8 --no-cpu-baseline) > gpurun_out/b_c5_rowsharded1.log 2>&1; echo "c5 rowsharded rc=$?"
tail -4 gpurun_out/b_default.log | cut -c1-600
for f in gpurun_out/b_gpus2*.log gpurun_out/b_c5_*.log; do echo == $f; grep -v "^$" $f | tail -5 | cut -c1-900; done
