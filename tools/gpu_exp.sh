#!/bin/bash
for V in "SPGEMM_BHCAP=10240" "SPGEMM_BHCAP=11264" "SPGEMM_BHCAP=12800"; do
  env $V timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify > gpurun_out/exp_v.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_v.json")); k=d["roofline"]["all_kernels_avg_ms"]; print("$V", d["ms_per_step"], {x:k[x] for x in k if "bighash" in x})
PY
done
