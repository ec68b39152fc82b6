#!/bin/bash
# per-lane carry registers in the insert loop of k_num_bighash (experiment builds nc1/nc2/nc3/nc2r1): parity, then bench
set -o pipefail
mkdir -p gpurun_out
SPGEMM_LIB=$PWD/sparse_matrix_with_flops_amd/libspgemm_hip_nc2.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 || exit 1
bash tools/gpu_ab_multi.sh synth_1m_16 nc1 nc2 nc3 nc2r1
bash tools/gpu_ab_multi.sh synth_1m_32 nc2
