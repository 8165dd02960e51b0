#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench,
# condensed into the small text/JSON summaries that get committed under profiles/.
# Usage: tools/collect_profiles.sh <tag>     (writes gpurun_out/profiles_<tag>/)
set -o pipefail
tag=${1:-r01}
out=gpurun_out/profiles_$tag
raw=gpurun_out/raw_$tag
mkdir -p $out $raw
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/stats -- $B > $raw/stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $raw/fetch -- $B --no-graph --steps 40 --warmup 5 > $raw/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $raw/write -- $B --no-graph --steps 40 --warmup 5 > $raw/write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $raw/sq -- $B --no-graph --steps 40 --warmup 5 > $raw/sq.log 2>&1 || exit 1
name=text8_d64_b131072
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline     ($tag, one MI355X)"
  echo "# bench line printed by the profiled run:"
  grep '"metric"' $raw/stats.log
  echo
  python tools/prof_summary.py $raw/stats 12
} > $out/${tag}_${name}_kernel_stats.txt
cap=$(grep -o '"chunk_cap": [0-9]*' $raw/stats.log | head -1 | grep -o '[0-9]*$')
python tools/pmc_traffic.py $raw/fetch $raw/write $out/${tag}_${name}_traffic.json workload=text8_d64 batch=131072 chunk_cap=$cap \
  "command=rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes) -- python bench.py --no-cpu-baseline --no-graph --steps 40 --warmup 5"
{
  echo "# rocprofv3 --pmc (separate passes; no trace domains), per-kernel averages per dispatch ($tag, text8_d64 B=131072)"
  for d in fetch write sq; do
    python tools/pmc_summary.py $raw/$d "glove::sidepass"; python tools/pmc_summary.py $raw/$d "glove::apply"
  done
} > $out/${tag}_${name}_pmc.txt
ls -la $out
