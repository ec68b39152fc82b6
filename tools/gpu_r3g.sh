#!/bin/bash
# round 3, closing checks: parity tests with durations, the R-MCL bench line with the host-array entry point beside it
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=5 2>&1 | tail -10 || exit 1
timeout -k 10 300 python bench.py --workload rmcl_500k --steps 10 --warmup 2 > gpurun_out/rmcl_hostapi.json 2> gpurun_out/rmcl_hostapi.err || exit 1
python - <<'PY'
import json
d = json.load(open("gpurun_out/rmcl_hostapi.json"))
print(d["ms_per_step"], d["value"], d["host_api"], d["parity"][:10])
PY
