#!/bin/bash
# knob sweep on the 1M workload: each VAR=VALUE set is one bench run, per-kernel times printed
mkdir -p gpurun_out
for V in "$@"; do
  env $V timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-verify > gpurun_out/exp_v.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_v.json")); k=d["roofline"]["all_kernels_avg_ms"]; print("$V", d["ms_per_step"], {x:round(k[x],3) for x in k if k[x]>0.05})
PY
done
