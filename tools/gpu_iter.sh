#!/bin/bash
# one GPU iteration: parity tests, serial per-kernel timings for 256K, then both benches
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "tests exit=$?"; tail -3 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py --workload synth_256k_16 --steps 10 --warmup 2 --no-cpu-baseline --no-verify > gpurun_out/bench_256k_serial.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
python - <<'PY'
import json
for f in ("gpurun_out/bench_256k_serial.json",):
    d=json.load(open(f)); print("SERIAL 256k", d["ms_per_step"], d["roofline"]["all_kernels_avg_ms"], d["roofline"]["phases_avg_ms"])
PY
SPGEMM_CONCURRENT=1 timeout -k 10 300 python bench.py --workload synth_256k_16 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_256k.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_1m.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
python - <<'PY'
import json
for f in ("gpurun_out/bench_256k.json","gpurun_out/bench_1m.json"):
    d=json.load(open(f)); print(f, d["ms_per_step"], "ms", d["value"], "GFLOP/s", d.get("parity"), d["roofline"]["all_kernels_avg_ms"], d["roofline"]["phases_avg_ms"])
PY
