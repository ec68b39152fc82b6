#!/bin/bash
# A/B of an experiment build against the product library: tools/gpu_ab.sh <variant name> [workloads...]
VS="$1"; shift
WLS=${@:-"synth_1m_16 web_google_surrogate synth_256k_16"}
mkdir -p gpurun_out
for wl in $WLS; do
  for v in product $VS; do
    if [ $v = product ]; then unset SPGEMM_LIB; else export SPGEMM_LIB=$PWD/sparse_matrix_with_flops_amd/libspgemm_hip_$v.so; fi
    timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-host-api > gpurun_out/ab_${wl}_$v.json 2> gpurun_out/ab_${wl}_$v.err; echo "$wl $v exit=$?"
  done
done
unset SPGEMM_LIB
python - <<'PY'
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d=json.load(open(f)); print(f.split("ab_")[1], d["ms_per_step"], "ms", d.get("parity","")[:8], d["roofline"]["phases_avg_ms"]["ms_symbolic"], d["roofline"]["phases_avg_ms"]["ms_numeric"], {k:v for k,v in d["roofline"]["all_kernels_avg_ms"].items() if v>0.08})
    except Exception as e: print(f, "failed", e)
PY
