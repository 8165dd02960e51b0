#!/bin/bash
t=r05
c() { name=$1; shift; bash tools/collect_profiles.sh $t "$@" > gpurun_out/prof_$name.log 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/prof_$name.log; }; echo "$name done"; date; }
c c3d c3_v50k_d300_b131072_index_rebuilt text8_v50k_d300 131072 100
c c3s c3_v50k_d300_b131072_static_index text8_v50k_d300 131072 100 --static-index
rm -rf gpurun_out/raw_${t}_*
