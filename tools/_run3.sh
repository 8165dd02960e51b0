set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forms or plan_build or full_size or adagrad_single or randomized" > gpurun_out/t_forms.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/t_forms.log
run() { name=$1; shift; timeout -k 10 300 python bench.py --single --no-cpu-baseline "$@" > gpurun_out/x_$name.log 2>&1 || echo "$name FAILED"; }
C4="--workload zipf_v400k_d300 --batch-size 1048576 --steps 40 --warmup 10"
C3="--workload text8_v50k_d300 --batch-size 131072 --steps 100 --warmup 10"
GLOVE_RECORDS=0 run c4_f1_cap16_norec $C4 --step-form 1 --chunk-cap 16
GLOVE_RECORDS=1 run c4_f1_cap16_rec $C4 --step-form 1 --chunk-cap 16
GLOVE_RECORDS=0 run c4_f1_cap32_norec $C4 --step-form 1 --chunk-cap 32
GLOVE_RECORDS=1 run c4_f1_cap32_rec $C4 --step-form 1 --chunk-cap 32
GLOVE_RECORDS=0 run c4_f1_cap8_norec $C4 --step-form 1 --chunk-cap 8
for cap in 8 16 32; do for per in 4 16; do
GLOVE_RECORDS=1 GLOVE_FUSE_PER=$per run c4_f3_cap${cap}_per$per $C4 --step-form 3 --chunk-cap $cap
done; done
GLOVE_RECORDS=1 GLOVE_FUSE_PER=4 run c4_f2_cap16_per4 $C4 --step-form 2 --chunk-cap 16
GLOVE_RECORDS=0 run c3_f1_cap16_norec $C3 --step-form 1 --chunk-cap 16
GLOVE_RECORDS=0 run c3_f1_cap32_norec $C3 --step-form 1 --chunk-cap 32
GLOVE_RECORDS=1 run c3_f1_cap16_rec $C3 --step-form 1 --chunk-cap 16
GLOVE_RECORDS=1 GLOVE_FUSE_PER=4 run c3_f3_cap16_per4 $C3 --step-form 3 --chunk-cap 16
GLOVE_RECORDS=1 GLOVE_FUSE_PER=2 run c3_f3_cap16_per2 $C3 --step-form 3 --chunk-cap 16
GLOVE_RECORDS=1 GLOVE_FUSE_PER=2 run c3_f3_cap32_per2 $C3 --step-form 3 --chunk-cap 32
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/x_*.log")):
    for line in open(f):
        if line.startswith('{"metric"'):
            d = json.loads(line); r = d["roofline"]
            print("%-24s %8.3f G nnz/s %8.2f us/step  chunks %d  %s" % (os.path.basename(f)[2:-4], d["value"]/1e9, d["ms_per_step"]*1e3, r["chunks_per_step"], {k: round(v,1) for k,v in r["kernel_us"].items()}))
PY
