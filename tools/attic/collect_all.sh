#!/bin/bash
# Runs on the GPU box: the committed profiles of this round (one directory gpurun_out/profiles_<tag>/).
tag=${1:-r03}
bash tools/collect_profiles.sh $tag text8_d64_b131072 text8_d64 131072 200 > gpurun_out/prof_t8.log 2>&1 || tail -5 gpurun_out/prof_t8.log
bash tools/collect_profiles.sh $tag c3_v50k_d300_b131072 text8_v50k_d300 131072 100 > gpurun_out/prof_c3.log 2>&1 || tail -5 gpurun_out/prof_c3.log
bash tools/collect_profiles.sh $tag c4_v400k_d300_b1m zipf_v400k_d300 1048576 40 > gpurun_out/prof_c4.log 2>&1 || tail -5 gpurun_out/prof_c4.log
bash tools/collect_profiles.sh $tag c5_v2m_d128_b1m zipf_v2m_d128 1048576 40 > gpurun_out/prof_c5.log 2>&1 || tail -5 gpurun_out/prof_c5.log
bash tools/collect_profiles.sh $tag c4_v400k_d300_b1m_two_launch zipf_v400k_d300 1048576 40 --step-form 1 > gpurun_out/prof_c4f1.log 2>&1 || tail -5 gpurun_out/prof_c4f1.log
ls -la gpurun_out/profiles_$tag
rm -rf gpurun_out/raw_${tag}_*
