#!/bin/bash
# round-3 closing check: GPU suite on the final tree, the default bench line (stdout must be exactly one JSON line), R-MCL line
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests_final.log 2>&1; echo "tests exit=$?"; tail -4 gpurun_out/gpu_tests_final.log
timeout -k 10 600 python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err; echo "default exit=$? lines=$(wc -l < gpurun_out/final_bench_default.json)"
timeout -k 10 600 python bench.py --workload rmcl_500k --steps 5 > gpurun_out/final_bench_rmcl.json 2> gpurun_out/final_bench_rmcl.err; echo "rmcl exit=$? lines=$(wc -l < gpurun_out/final_bench_rmcl.json)"
BENCH_FORCE_GROUP=1 timeout -k 10 300 python bench.py --workload synth_256k_16 --steps 5 --warmup 1 --no-cpu-baseline --no-host-api > gpurun_out/final_bench_group.json 2> gpurun_out/final_bench_group.err; echo "group exit=$? lines=$(wc -l < gpurun_out/final_bench_group.json)"
python - <<'PY'
import json
for f in ("final_bench_default","final_bench_rmcl","final_bench_group"):
    d=json.load(open(f"gpurun_out/{f}.json")); r=d.get("roofline") or {}
    print(f, d["ms_per_step"], d["value"], d.get("parity","")[:12], "frac", r.get("frac"), "traffic", r.get("traffic"), r.get("traffic_over_alg"), (r.get("traffic_source") or "")[:40], "host", (d.get("host_api") or {}).get("ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d.get("transport"))
PY
python __graft_entry__.py smoke
