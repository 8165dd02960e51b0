#!/bin/bash
# Runs on the GPU box: every committed profile of round 5 (profiles/r05_*), regenerated from the bench commands with the
# library in the tree.  Output: gpurun_out/profiles_r05/ (copy into profiles/).  One log per configuration.
# Usage: bash tools/collect_r05.sh [names...]   e.g. "c4d c4s c5d c5s" (default: all)
t=r05
want=" ${*:-c4d c4s c5d c5s c3d c3s t8 t1k c1} "
c() { name=$1; shift; case "$want" in *" $name "*) ;; *) return;; esac
      bash tools/collect_profiles.sh $t "$@" > gpurun_out/prof_$name.log 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/prof_$name.log; }; echo "$name done"; date; }
c c4d c4_v400k_d300_b1m_index_rebuilt zipf_v400k_d300 1048576 40
c c4s c4_v400k_d300_b1m_static_index zipf_v400k_d300 1048576 40 --static-index
c c5d c5_v2m_d128_b1m_index_rebuilt zipf_v2m_d128 1048576 40
c c5s c5_v2m_d128_b1m_static_index zipf_v2m_d128 1048576 40 --static-index
c c3d c3_v50k_d300_b131072_index_rebuilt text8_v50k_d300 131072 100
c c3s c3_v50k_d300_b131072_static_index text8_v50k_d300 131072 100 --static-index
c t8 text8_d64_b131072_index_rebuilt text8_d64 131072 200
c t1k text8_d64_b1024_index_rebuilt text8_d64 1024 2000
c c1 c1_adam_d64_b1024_index_rebuilt text8_d64 1024 2000 --optimizer Adam --learning-rate 0.001
ls -la gpurun_out/profiles_$t
rm -rf gpurun_out/raw_${t}_*
