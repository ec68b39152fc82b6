#!/bin/bash
# host-array entry point after a change of the copy lanes: its GPU test + timing on the headline matrix
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_abi.py -x -q -m gpu -k "host_array or compressive" 2>&1 | tail -3
timeout -k 10 400 python - <<'PY'
import sys, json
sys.path.insert(0, ".")
import torch; torch.zeros(1, device="cuda")
from sparse_matrix_with_flops_amd import synth, hipspgemm as hs
rp, ci, v = synth.powerlaw_csr(1 << 20, 43, 2)
A = hs.CSR.from_arrays(rp, ci, v, 1 << 20, 1 << 20)
runs = hs.host_api_timed(A, A, reps=5)
for r in runs: print({k: (round(x, 1) if isinstance(x, float) else x) for k, x in r.items()})
PY
