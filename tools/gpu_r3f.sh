#!/bin/bash
# round 3: the R-MCL device loop (Mt unpacked between iterations): tests, then A/B of the loop against packing every iteration
set -o pipefail
mkdir -p gpurun_out/r3f
timeout -k 10 600 python -m pytest tests/test_gpu_rmcl.py tests/test_gpu_sharded_abi.py -x -q -m gpu > gpurun_out/r3f/pytest.log 2>&1 || { tail -30 gpurun_out/r3f/pytest.log; exit 1; }
tail -3 gpurun_out/r3f/pytest.log
for i in 1 2; do
  timeout -k 10 300 python bench.py --workload rmcl_500k --steps 20 --warmup 3 > gpurun_out/r3f/rmcl_unpacked_$i.json 2> gpurun_out/r3f/rmcl_unpacked_$i.err || exit 1
  SPGEMM_RMCL_PACK=1 timeout -k 10 300 python bench.py --workload rmcl_500k --steps 20 --warmup 3 > gpurun_out/r3f/rmcl_packed_$i.json 2> gpurun_out/r3f/rmcl_packed_$i.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3f/rmcl_*.json')):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['value'], d.get('loop_one_call_per_iteration_ms'), d.get('parity'))
PY
