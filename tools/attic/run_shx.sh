set -e
python -m pytest tests -x -q -m gpu -k "shard or dealt or reshuffl" > gpurun_out/shx_tests.log 2>&1 || { tail -n 30 gpurun_out/shx_tests.log; exit 1; }
tail -n 3 gpurun_out/shx_tests.log
python bench.py --row-sharded --collectives --exercise-exchange --single --no-cpu-baseline --steps 96 --warmup 30 > gpurun_out/shx_dealt.json 2> gpurun_out/shx_dealt.err
python bench.py --row-sharded --collectives --exercise-exchange --static-index --single --no-cpu-baseline --steps 48 --warmup 6 > gpurun_out/shx_static.json 2> gpurun_out/shx_static.err
