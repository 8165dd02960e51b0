set -e
R=$GRAFT_REPO_ROOT
python bench.py --row-sharded --collectives --exercise-exchange --single --no-cpu-baseline --steps 96 --warmup 30 > gpurun_out/shx_dealt.json 2> gpurun_out/shx_dealt.err
python bench.py --row-sharded --collectives --exercise-exchange --static-index --single --no-cpu-baseline --steps 48 --warmup 6 > gpurun_out/shx_static.json 2> gpurun_out/shx_static.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_shx4 -o shx -- python3 $R/bench.py --row-sharded --collectives --exercise-exchange --single --no-cpu-baseline --steps 72 --warmup 30 > $R/gpurun_out/shx_prof.json 2> $R/gpurun_out/shx_prof.err
python3 $R/tools/attic/queue_overlap.py $R/gpurun_out/prof_shx4/shx_results.db > $R/gpurun_out/shx4_queues.txt
