#!/bin/bash
for V in "SPGEMM_U=2"; do
  env $V timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/exp_v.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
  env $V timeout -k 10 300 python bench.py --workload synth_256k_16 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/exp_w.json 2> gpurun_out/err.txt || tail -5 gpurun_out/err.txt
  python - <<PY
import json
for f in ("gpurun_out/exp_v.json","gpurun_out/exp_w.json"):
    d=json.load(open(f)); k=d["roofline"]["all_kernels_avg_ms"]; print("$V", d["ms_per_step"], d["parity"][:2], {x:k[x] for x in k if "hash" in x or "big" in x})
PY
done
