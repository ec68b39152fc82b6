#!/bin/bash
# parity tests under the tuning knobs users can set
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t_default.log 2>&1; echo "default exit=$?"; tail -2 gpurun_out/t_default.log
for V in "SPGEMM_U=4" "SPGEMM_U=8" "SPGEMM_CONCURRENT=1" "SPGEMM_BHCAP=12800"; do
  env $V timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t_var.log 2>&1; echo "$V exit=$?"; tail -1 gpurun_out/t_var.log
done
