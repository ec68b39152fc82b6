#!/bin/bash
# experiment: per-kernel serial timings under several env settings (256k workload)
for U in 2 4 8; do
  SPGEMM_U=$U timeout -k 10 300 python bench.py --workload synth_256k_16 --steps 10 --warmup 2 --no-cpu-baseline --no-verify > gpurun_out/exp_u$U.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_u$U.json")); print("U=$U", d["ms_per_step"], {k:v for k,v in d["roofline"]["all_kernels_avg_ms"].items() if "hash" in k}, d["roofline"]["phases_avg_ms"])
PY
done
