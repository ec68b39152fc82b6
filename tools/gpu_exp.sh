#!/bin/bash
# experiment: bighash distinct-column cap per pass (1M workload)
for CAP in 10240; do
  SPGEMM_BHCAP=$CAP timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify > gpurun_out/exp_cap$CAP.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_cap$CAP.json")); print("CAP=$CAP", d["ms_per_step"], {k:v for k,v in d["roofline"]["all_kernels_avg_ms"].items() if "big" in k or "hash<" in k})
PY
done
SPGEMM_CONCURRENT=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify > gpurun_out/exp_conc.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
python - <<PY
import json
d=json.load(open("gpurun_out/exp_conc.json")); print("CONCURRENT 1m", d["ms_per_step"], d["roofline"]["phases_avg_ms"])
PY
