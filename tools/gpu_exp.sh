#!/bin/bash
# experiment: occupancy sensitivity of the wave-per-row symbolic kernel (blocks per CU), 1M workload
for B in 8 16 24 32; do
  SPGEMM_H1SYM=$B timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify > gpurun_out/exp_h1_$B.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_h1_$B.json")); print("H1SYM=$B", d["ms_per_step"], d["roofline"]["all_kernels_avg_ms"]["k_sym_hash<1,1024>"])
PY
done
