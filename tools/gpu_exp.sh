#!/bin/bash
timeout -k 10 600 python bench.py --workload synth_1m_32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_1m_32.json 2> gpurun_out/err32.txt; tail -2 gpurun_out/err32.txt
python - <<PY
import json
d=json.load(open("gpurun_out/bench_1m_32.json")); k=d["roofline"]["all_kernels_avg_ms"]; print("1m_32", d["ms_per_step"], d["value"], d.get("parity","")[:2], {x:k[x] for x in k if "big" in x})
PY
