set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_q1 -o q -- python3 $R/bench.py --single --no-cpu-baseline --steps 96 --warmup 24 > $R/gpurun_out/q1.json 2> $R/gpurun_out/q1.err
python3 $R/tools/attic/queue_overlap.py $R/gpurun_out/prof_q1/q_results.db > $R/gpurun_out/q1.txt
export GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_q2 -o q -- python3 $R/bench.py --single --no-cpu-baseline --steps 96 --warmup 24 > $R/gpurun_out/q2.json 2> $R/gpurun_out/q2.err
python3 $R/tools/attic/queue_overlap.py $R/gpurun_out/prof_q2/q_results.db > $R/gpurun_out/q2.txt
rm -rf $R/gpurun_out/prof_q1 $R/gpurun_out/prof_q2
cd $R
python bench.py --single --no-cpu-baseline --steps 96 --warmup 24 > gpurun_out/q3.json 2> gpurun_out/q3.err
unset GPU_MAX_HW_QUEUES
python bench.py --single --no-cpu-baseline --steps 96 --warmup 24 > gpurun_out/q4.json 2> gpurun_out/q4.err
GPU_MAX_HW_QUEUES=8 python bench.py --single --no-cpu-baseline --steps 96 --warmup 24 > gpurun_out/q5.json 2> gpurun_out/q5.err
python bench.py --single --no-cpu-baseline --steps 96 --warmup 24 > gpurun_out/q6.json 2> gpurun_out/q6.err
