#!/bin/bash
# folded probe tails (SMF_FOLD_AFTER: product = 2; variants 1, 3, never): parity first, then the bench workloads
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py tests/test_gpu_fuzz.py tests/test_gpu_rmcl.py -x -q -m gpu 2>&1 | tail -4 || exit 1
bash tools/gpu_ab_multi.sh synth_1m_16 fold1000000 fold1 fold3
bash tools/gpu_ab_multi.sh web_google_surrogate fold1000000 fold1 fold3
