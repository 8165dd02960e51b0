#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of one bench configuration,
# condensed into the small text/JSON summaries that get committed under profiles/.
# Usage: tools/collect_profiles.sh <tag> <name> <workload> <batch> [steps] [extra bench args...]
#   e.g. tools/collect_profiles.sh r02 c4 zipf_v400k_d300 1048576 40
# writes gpurun_out/profiles_<tag>/<tag>_<name>_{kernel_stats.txt,pmc.txt,traffic.json}
set -o pipefail
tag=${1:-r02}; name=${2:-text8_d64_b131072}; wl=${3:-text8_d64}; batch=${4:-131072}; steps=${5:-200}
shift 5 2>/dev/null || shift $#
extra="$*"
out=gpurun_out/profiles_$tag
raw=gpurun_out/raw_${tag}_$name
mkdir -p $out $raw
export TMPDIR=/tmp
B="python3 bench.py --single --no-cpu-baseline --workload $wl --batch-size $batch $extra"
psteps=$(( steps < 40 ? steps : 40 ))
echo "[$name] kernel trace"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/stats -- $B --steps $steps --warmup 10 > $raw/stats.log 2>&1 || { tail -5 $raw/stats.log; exit 1; }
echo "[$name] pmc fetch"; date
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $raw/fetch -- $B --no-graph --steps $psteps --warmup 5 > $raw/fetch.log 2>&1 || { tail -5 $raw/fetch.log; exit 1; }
echo "[$name] pmc write"; date
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $raw/write -- $B --no-graph --steps $psteps --warmup 5 > $raw/write.log 2>&1 || { tail -5 $raw/write.log; exit 1; }
echo "[$name] pmc sq"; date
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $raw/sq -- $B --no-graph --steps $psteps --warmup 5 > $raw/sq.log 2>&1 || { tail -5 $raw/sq.log; exit 1; }
{
  echo "# rocprofv3 --kernel-trace --stats -- $B --steps $steps --warmup 10     ($tag, one MI355X)"
  echo "# bench line printed by the profiled run:"
  grep '"metric"' $raw/stats.log
  echo
  python3 tools/prof_summary.py $raw/stats 12
  echo
  echo "# the fused step's launches apart (rocprofv3 --kernel-trace CSV of the same run, tools/kt_triples.py):"
  python3 tools/kt_triples.py $raw/stats
} > $out/${tag}_${name}_kernel_stats.txt
cap=$(grep -o '"chunk_cap": [0-9]*' $raw/stats.log | head -1 | grep -o '[0-9]*$')
# (the per-step traffic JSON delimits steps by the apply launch: only for the plain single-GPU step forms)
case "$extra" in *--row-sharded*|*--exchange*|*--force-dense*|*--optimizer*) ;; *)
idx=dealt; case "$extra" in *--static-index*) idx=static;; esac
python3 tools/pmc_traffic.py $raw/fetch $raw/write $out/${tag}_${name}_traffic.json workload=$wl batch=$batch chunk_cap=$cap index=$idx \
  "command=rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes) -- $B --no-graph --steps $psteps --warmup 5"
;; esac
{
  echo "# rocprofv3 --pmc (separate passes; no trace domains), per-kernel averages per dispatch ($tag, $wl B=$batch)"
  for d in fetch write sq; do
    python3 tools/pmc_summary.py $raw/$d "glove::"
  done
} > $out/${tag}_${name}_pmc.txt
ls -la $out
