#!/bin/bash
t=r05
c() { name=$1; shift; bash tools/collect_profiles.sh $t "$@" > gpurun_out/prof_$name.log 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/prof_$name.log; }; echo "$name done"; date; }
c c4d c4_v400k_d300_b1m_index_rebuilt zipf_v400k_d300 1048576 40
c c4s c4_v400k_d300_b1m_static_index zipf_v400k_d300 1048576 40 --static-index
c c5d c5_v2m_d128_b1m_index_rebuilt zipf_v2m_d128 1048576 40
c c5s c5_v2m_d128_b1m_static_index zipf_v2m_d128 1048576 40 --static-index
rm -rf gpurun_out/raw_${t}_*
