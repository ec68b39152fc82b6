#!/bin/bash
# A/B of several experiment builds against the product library on one workload: tools/gpu_ab_multi.sh <workload> <variant>...
WL="$1"; shift
mkdir -p gpurun_out
for v in product "$@" product; do
  if [ $v = product ]; then unset SPGEMM_LIB; else export SPGEMM_LIB=$PWD/sparse_matrix_with_flops_amd/libspgemm_hip_$v.so; fi
  timeout -k 10 400 python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-host-api > gpurun_out/abm_${WL}_$v.json 2> gpurun_out/abm_${WL}_$v.err; echo "$WL $v exit=$?"
  python - <<PY
import json
d=json.load(open("gpurun_out/abm_${WL}_$v.json")); r=d["roofline"]
print("$v", d["ms_per_step"], d.get("parity","")[:8], r["phases_avg_ms"]["ms_symbolic"], r["phases_avg_ms"]["ms_numeric"], {k:v for k,v in r["all_kernels_avg_ms"].items() if v>0.1})
PY
done
