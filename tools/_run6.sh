set -o pipefail
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import glove_ref as ref, glove_ref_c
from helpers import make_batch
for (B,V,d,opt) in ((131072,10000,64,"Adagrad"),(1024,10000,64,"Adam")):
    row,col,w,y=make_batch(1,B,V)
    t=ref.Tables(V,d,opt,dtype=np.float32,seed=1)
    hp=ref.Hyper(learning_rate=0.05 if opt=="Adagrad" else 0.001)
    a=glove_ref_c.CPort(t,B)
    ix=glove_ref_c.BatchIndex(row,col,d)
    for th in (1,4,8,16,32,64,128):
        a.step_mt(ix,row,col,w,y,hp,threads=th)
        t0=time.perf_counter(); n=0
        while time.perf_counter()-t0 < 0.5:
            a.step_mt(ix,row,col,w,y,hp,threads=th); n+=1
        print(opt, B, th, "threads: %.2f M nnz/s" % (n*B/(time.perf_counter()-t0)/1e6), flush=True)
PY
(time timeout -k 10 600 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 6 --warmup 2 --single --row-sharded --workload zipf_v2m_d128 --batch-size 262144 --max-batches 4) > gpurun_out/b_gpus2_sharded.log 2>&1; echo "gpus2 sharded rc=$?"
(time timeout -k 10 600 python bench.py --single --row-sharded --workload zipf_v2m_d128 --batch-size 1048576 --steps 20 --warmup 5 --max-batches 8 --no-cpu-baseline) > gpurun_out/b_c5_sharded1.log 2>&1; echo "c5 sharded rc=$?"
(time timeout -k 10 600 python bench.py --single --force-dense --exchange rows --workload zipf_v400k_d300 --batch-size 131072 --steps 20 --warmup 5 --max-batches 8 --no-cpu-baseline) > gpurun_out/b_c4_rows1.log 2>&1; echo "c4 rows rc=$?"
(time timeout -k 10 600 python bench.py --single --force-dense --exchange dense --workload zipf_v400k_d300 --batch-size 131072 --steps 20 --warmup 5 --max-batches 8 --no-cpu-baseline) > gpurun_out/b_c4_dense1.log 2>&1; echo "c4 dense rc=$?"
(time timeout -k 10 600 python bench.py --single --workload zipf_v400k_d300 --batch-size 131072 --steps 20 --warmup 5 --max-batches 8 --no-cpu-baseline) > gpurun_out/b_c4_sparse131k.log 2>&1; echo "c4 sparse rc=$?"
for f in gpurun_out/b_gpus2_sharded.log gpurun_out/b_c5_sharded1.log gpurun_out/b_c4_rows1.log gpurun_out/b_c4_dense1.log gpurun_out/b_c4_sparse131k.log; do echo == $f; grep -v "^$" $f | grep -v Gloo | tail -5 | cut -c1-400; python3 - $f <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        d = json.loads(line); print("   ", round(d["value"]/1e9,3), "G nnz/s", round(d["ms_per_step"]*1e3,1), "us", d["config"]["parallelism"], {k: round(v,1) for k,v in d["roofline"]["kernel_us"].items()})
PY
done
