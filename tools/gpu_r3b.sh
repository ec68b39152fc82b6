#!/bin/bash
# round-3: full GPU suite on the CAS-add build, gather-rate microbenchmark, A/B of the CAS add (product vs nocas variant)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "tests exit=$?"; tail -8 gpurun_out/gpu_tests.log
timeout -k 10 200 tools/micro/gather_rate.x > gpurun_out/gather_rate.txt 2>&1; echo "micro exit=$?"
cat gpurun_out/gather_rate.txt
for wl in web_google_surrogate synth_1m_16 synth_256k_16; do
  for v in product nocas; do
    if [ $v = product ]; then unset SPGEMM_LIB; else export SPGEMM_LIB=$PWD/sparse_matrix_with_flops_amd/libspgemm_hip_$v.so; fi
    timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-host-api > gpurun_out/ab_${wl}_$v.json 2> gpurun_out/ab_${wl}_$v.err; echo "$wl $v exit=$?"
  done
done
unset SPGEMM_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d=json.load(open(f)); print(f, d["ms_per_step"], "ms", d["value"], "GFLOP/s", {k:v for k,v in d["roofline"]["all_kernels_avg_ms"].items() if k.startswith("k_num")})
    except Exception as e: print(f, "failed", e)
PY
