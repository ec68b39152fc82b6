#!/bin/bash
# benches only: 256K and 1M workloads, per-kernel timings
mkdir -p gpurun_out
for W in synth_256k_16 synth_1m_16; do
timeout -k 10 300 python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline $BENCH_EXTRA > gpurun_out/bench_$W.json 2> gpurun_out/err.txt || { tail -5 gpurun_out/err.txt; exit 1; }
done
python - <<'PY'
import json
for w in ("synth_256k_16","synth_1m_16"):
    d=json.load(open(f"gpurun_out/bench_{w}.json")); print(w, d["ms_per_step"], "ms", d["value"], "GFLOP/s", d.get("parity"), d["roofline"]["all_kernels_avg_ms"], d["roofline"]["phases_avg_ms"])
PY
