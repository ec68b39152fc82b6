#!/bin/bash
# scratch knob sweep on the 1M workload
for V in "SPGEMM_U=2"; do
  env $V timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/exp_v.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_v.json")); k=d["roofline"]["all_kernels_avg_ms"]; print("$V", d["ms_per_step"], d["parity"][:2], {x:k[x] for x in k if "num_hash" in x})
PY
done
