set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forms or plan_build or full_size or adagrad_single" > gpurun_out/t_forms.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_forms.log
for f in 1 2 3; do
  timeout -k 10 300 python bench.py --single --no-cpu-baseline --workload zipf_v400k_d300 --batch-size 1048576 --steps 40 --warmup 10 --step-form $f > gpurun_out/c4_form$f.log 2>&1; echo "c4 form $f rc=$?"
  timeout -k 10 300 python bench.py --single --no-cpu-baseline --workload text8_v50k_d300 --batch-size 131072 --steps 100 --warmup 10 --step-form $f > gpurun_out/c3_form$f.log 2>&1; echo "c3 form $f rc=$?"
  timeout -k 10 300 python bench.py --single --no-cpu-baseline --step-form $f > gpurun_out/t8_form$f.log 2>&1; echo "t8 form $f rc=$?"
  timeout -k 10 300 python bench.py --single --no-cpu-baseline --workload zipf_v2m_d128 --batch-size 1048576 --steps 40 --warmup 10 --step-form $f > gpurun_out/c5_form$f.log 2>&1; echo "c5 form $f rc=$?"
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/*_form*.log")):
    for line in open(f):
        if line.startswith('{"metric"'):
            d = json.loads(line); r = d["roofline"]
            print("%-14s %8.3f G nnz/s %8.2f us/step alg %7.0f GB/s frac %.3f %s" % (os.path.basename(f)[:-4], d["value"]/1e9, d["ms_per_step"]*1e3, r["achieved"], r["frac"], {k: round(v,2) for k,v in r["kernel_us"].items()}))
PY
