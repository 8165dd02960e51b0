set -e
L=${LIBS:-tools/ab_libs/libglove_base.so,glove-tensorflow_amd/lib/libglove_hip.so}
python tools/ab_kernels.py --workload zipf_v400k_d300 --batch-size 1048576 --caps 32 --libs $L --twin --only step --rounds 5 --check
python tools/ab_kernels.py --workload zipf_v2m_d128 --batch-size 1048576 --caps 32 --libs $L --twin --only step --rounds 5 --check
python tools/ab_kernels.py --workload zipf_v400k_d300 --batch-size 131072 --caps 32 --libs $L --twin --only step --rounds 5 --check
