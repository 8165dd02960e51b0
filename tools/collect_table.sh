#!/bin/bash
# Runs on the GPU box: the bench lines behind DESIGN.md §6 (one log per row under gpurun_out/table/).
out=gpurun_out/table
mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $out/$name.log 2>&1 || echo "$name FAILED"; }
timeout -k 10 300 python bench.py > $out/default.log 2>&1 || echo "default FAILED"
run b1024 --batch-size 1024 --steps 4000 --warmup 200
run b1m --batch-size 1048576
run adam1024 --optimizer Adam --batch-size 1024 --steps 4000 --warmup 200
run c3_131k --workload text8_v50k_d300
run c3_1m --workload text8_v50k_d300 --batch-size 1048576
run c4_1m --workload zipf_v400k_d300 --batch-size 1048576 --steps 60 --warmup 10
run c5_sparse --workload zipf_v2m_d128 --batch-size 1048576 --steps 60 --warmup 10
run c5_rowsharded --workload zipf_v2m_d128 --batch-size 1048576 --row-sharded --steps 60 --warmup 10
run dyn_131k --dynamic
run dyn_1024 --dynamic --batch-size 1024 --steps 2000 --warmup 200
run dense --force-dense
timeout -k 10 300 python tools/bench_trainer.py > $out/trainer.log 2>&1 || echo "trainer FAILED"
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/table/*.log")):
    for line in open(f):
        if line.startswith('{"metric"'):
            d = json.loads(line)
            r = d["roofline"]
            print("%-14s %8.3f G nnz/s  %8.2f us/step  alg %7.0f GB/s frac %.3f  %s" % (
                os.path.basename(f)[:-4], d["value"] / 1e9, d["ms_per_step"] * 1e3, r["achieved"], r["frac"],
                {k: round(v, 2) for k, v in r["kernel_us"].items()}))
            if "cpu_baseline" in d:
                print("   cpu_baseline", d["cpu_baseline"])
PY
tail -5 $out/trainer.log
