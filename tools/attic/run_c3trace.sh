set -e
GLOVE_SEG_TRACE=1 python bench.py --single --no-cpu-baseline --workload text8_v50k_d300 --batch-size 131072 --steps 240 --warmup 24 > gpurun_out/c3_seg.json 2> gpurun_out/c3_seg.err
