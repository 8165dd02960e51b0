set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forms or plan_build or full_size or adagrad_single or randomized" > gpurun_out/t_forms.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/t_forms.log
for per in 16 32; do
for f in 3 2; do
  GLOVE_FUSE_PER=$per timeout -k 10 300 python bench.py --single --no-cpu-baseline --workload zipf_v400k_d300 --batch-size 1048576 --steps 40 --warmup 10 --step-form $f > gpurun_out/c4_form${f}_per$per.log 2>&1; echo "c4 form $f rc=$?"
  GLOVE_FUSE_PER=$per timeout -k 10 300 python bench.py --single --no-cpu-baseline --workload text8_v50k_d300 --batch-size 131072 --steps 100 --warmup 10 --step-form $f > gpurun_out/c3_form${f}_per$per.log 2>&1; echo "c3 form $f rc=$?"
  GLOVE_FUSE_PER=$per timeout -k 10 300 python bench.py --single --no-cpu-baseline --step-form $f > gpurun_out/t8_form${f}_per$per.log 2>&1; echo "t8 form $f rc=$?"
done
done
GLOVE_FUSE_PER=16 timeout -k 10 300 python bench.py --single --no-cpu-baseline --workload zipf_v400k_d300 --batch-size 1048576 --steps 40 --warmup 10 --step-form 3 --chunk-cap 8 > gpurun_out/c4_form3_cap8.log 2>&1
GLOVE_FUSE_PER=16 timeout -k 10 300 python bench.py --single --no-cpu-baseline --workload zipf_v400k_d300 --batch-size 1048576 --steps 40 --warmup 10 --step-form 3 --chunk-cap 32 > gpurun_out/c4_form3_cap32.log 2>&1
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/*_form*_*.log")):
    for line in open(f):
        if line.startswith('{"metric"'):
            d = json.loads(line); r = d["roofline"]
            print("%-20s %8.3f G nnz/s %8.2f us/step  chunks %d" % (os.path.basename(f)[:-4], d["value"]/1e9, d["ms_per_step"]*1e3, r["chunks_per_step"]))
PY
