#!/bin/bash
# round-3 first look: new tests, L2 atomics microbenchmark, the web-Google surrogate and the headline workload
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sharded_abi.py tests/test_gpu_rmcl.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/gpu_tests_new.log 2>&1; echo "new tests exit=$?"; tail -15 gpurun_out/gpu_tests_new.log
timeout -k 10 120 tools/micro/l2_atomics.x > gpurun_out/l2_atomics.txt 2>&1; echo "micro exit=$?"
cat gpurun_out/l2_atomics.txt
timeout -k 10 400 python bench.py --workload web_google_surrogate --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_web.json 2> gpurun_out/bench_web.err; echo "web exit=$?"
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify > gpurun_out/bench_1m.json 2> gpurun_out/bench_1m.err; echo "1m exit=$?"
python - <<'PY'
import json
for f in ("gpurun_out/bench_web.json","gpurun_out/bench_1m.json"):
    try:
        d=json.load(open(f)); print(f, d["ms_per_step"], "ms", d["value"], "GFLOP/s", d.get("parity"), d["roofline"]["all_kernels_avg_ms"], d["roofline"]["phases_avg_ms"])
    except Exception as e: print(f, "failed", e)
PY
