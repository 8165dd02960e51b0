set -o pipefail
export TMPDIR=/tmp
C4="--workload zipf_v400k_d300 --batch-size 1048576 --steps 20 --warmup 5"
GLOVE_RECORDS=1 GLOVE_FUSE_PER=4 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_c4f3 -- python3 bench.py --single --no-cpu-baseline $C4 --step-form 3 --chunk-cap 32 > gpurun_out/kt_c4f3.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt_c4f3/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "glove::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(("pass_fused" if "true>" in r["Kernel_Name"].split("(")[0] and "sidepass" in r["Kernel_Name"] else "pass" if "sidepass" in r["Kernel_Name"] else "apply" if "apply" in r["Kernel_Name"] else "other"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size"]) for r in rows]
# find the steady-state triples pass_fused, pass_fused, apply
tri = [(seq[i][1], seq[i+1][1], seq[i+2][1], seq[i][2], seq[i+1][2], seq[i+2][2]) for i in range(len(seq) - 2) if seq[i][0] == "pass_fused" and seq[i+1][0] == "pass_fused" and seq[i+2][0] == "apply"]
tri = tri[5:]
import statistics as st
print("triples", len(tri), "K1 %.1f K2 %.1f K3 %.1f us; grids %s" % (st.median(t[0] for t in tri), st.median(t[1] for t in tri), st.median(t[2] for t in tri), tri[0][3:]))
PY
run() { name=$1; shift; timeout -k 10 300 python bench.py --single --no-cpu-baseline "$@" > gpurun_out/x_$name.log 2>&1 || echo "$name FAILED"; }
C4="--workload zipf_v400k_d300 --batch-size 1048576 --steps 40 --warmup 10"
C5="--workload zipf_v2m_d128 --batch-size 1048576 --steps 40 --warmup 10"
for per in 1 2 8; do
GLOVE_RECORDS=1 GLOVE_FUSE_PER=$per run c4_f3_cap32_per$per $C4 --step-form 3 --chunk-cap 32
done
GLOVE_RECORDS=1 run c5_f1_cap16 $C5 --step-form 1 --chunk-cap 16
for per in 2 4; do
GLOVE_RECORDS=1 GLOVE_FUSE_PER=$per run c5_f3_cap16_per$per $C5 --step-form 3 --chunk-cap 16
GLOVE_RECORDS=1 GLOVE_FUSE_PER=$per run c5_f3_cap8_per$per $C5 --step-form 3 --chunk-cap 8
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/x_*.log")):
    for line in open(f):
        if line.startswith('{"metric"'):
            d = json.loads(line); r = d["roofline"]
            print("%-24s %8.3f G nnz/s %8.2f us/step  chunks %d" % (os.path.basename(f)[2:-4], d["value"]/1e9, d["ms_per_step"]*1e3, r["chunks_per_step"]))
PY
