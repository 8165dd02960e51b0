#!/bin/bash
# Runs on the GPU box: the bench lines behind DESIGN.md §6 (one log per row under gpurun_out/table/).
out=gpurun_out/table
rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python bench.py --single --no-cpu-baseline "$@" > $out/$name.log 2>&1 || echo "$name FAILED"; }
timeout -k 10 600 python bench.py > $out/default.log 2>&1 || echo "default FAILED"
run b1024 --workload text8_d64 --batch-size 1024 --steps 4000 --warmup 200
run b1m --workload text8_d64 --batch-size 1048576
run adam1024 --workload text8_d64 --optimizer Adam --batch-size 1024 --steps 4000 --warmup 200 --learning-rate 0.001
run c3_131k_two_launch --workload text8_v50k_d300 --step-form 1
run c3_1m --workload text8_v50k_d300 --batch-size 1048576 --steps 60 --warmup 10
run c4_1m_two_launch --workload zipf_v400k_d300 --batch-size 1048576 --steps 40 --warmup 10 --step-form 1
run c4_1m_three_launch --workload zipf_v400k_d300 --batch-size 1048576 --steps 40 --warmup 10 --step-form 3
run c4_1m_twin --workload zipf_v400k_d300 --batch-size 1048576 --steps 40 --warmup 10
run c4_131k --workload zipf_v400k_d300 --batch-size 131072 --steps 100 --warmup 10
run c5_two_launch --workload zipf_v2m_d128 --batch-size 1048576 --steps 40 --warmup 10 --step-form 1
run c5_three_launch --workload zipf_v2m_d128 --batch-size 1048576 --steps 40 --warmup 10 --step-form 3
run c5_twin --workload zipf_v2m_d128 --batch-size 1048576 --steps 40 --warmup 10
run c5_sharded_world1 --workload zipf_v2m_d128 --batch-size 1048576 --row-sharded --steps 40 --warmup 10 --max-batches 8
run c5_rowsharded_world1 --workload zipf_v2m_d128 --batch-size 1048576 --row-sharded --cols-replicated --steps 40 --warmup 10 --max-batches 8
run static_131k --static-index
run static_1024 --static-index --batch-size 1024 --steps 2000 --warmup 200
run dp1_dense --force-dense --exchange dense
run dp1_rows --force-dense --exchange rows
run c4_131k_dp1_dense --workload zipf_v400k_d300 --batch-size 131072 --force-dense --exchange dense --steps 60 --warmup 10 --max-batches 8
run c4_131k_dp1_rows --workload zipf_v400k_d300 --batch-size 131072 --force-dense --exchange rows --steps 60 --warmup 10 --max-batches 8
run c4_1m_dp1_dense --workload zipf_v400k_d300 --batch-size 1048576 --force-dense --exchange dense --steps 30 --warmup 5 --max-batches 8
timeout -k 10 300 python tools/bench_trainer.py > $out/trainer.log 2>&1 || echo "trainer FAILED"
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/table/*.log")):
    for line in open(f):
        if line.startswith('{"metric"'):
            d = json.loads(line)
            r = d["roofline"]
            print("%-24s %8.3f G nnz/s  %8.2f us/step  alg %7.0f GB/s frac %.3f (%.3f of 6.29)  %s  %s" % (
                os.path.basename(f)[:-4], d["value"] / 1e9, d["ms_per_step"] * 1e3, r["achieved"], r["frac"],
                r["frac_of_measured_stream_ceiling"], {k: round(v, 1) for k, v in r["kernel_us"].items()}, d["config"]["parallelism"]))
            for c in d.get("configs", []):
                rc = c["roofline"]
                print("   %-32s %8.3f G nnz/s  %8.2f us/step  frac %.3f  %s" % (c["name"], c["value"] / 1e9, c["ms_per_step"] * 1e3, rc["frac"], {k: round(v, 1) for k, v in rc["kernel_us"].items()}))
            if "cpu_baseline" in d:
                cb = d["cpu_baseline"]
                print("   cpu_baseline", {k: (round(v["value"] / 1e6, 2), v["cores"]) for k, v in cb["legs"].items() if isinstance(v, dict)}, cb["cpu_model"], cb["host_cpus"], cb["usable_cores"])
PY
tail -6 $out/trainer.log
