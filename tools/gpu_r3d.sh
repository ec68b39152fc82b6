#!/bin/bash
# round-3 experiment: packed {column, value} copy of B (one 8-byte gather per product in the numeric kernels); host API after
# the madvise change
mkdir -p gpurun_out
for wl in synth_1m_16 web_google_surrogate synth_256k_16; do
  for v in product packed packedsym; do
    if [ $v = product ]; then unset SPGEMM_LIB; else export SPGEMM_LIB=$PWD/sparse_matrix_with_flops_amd/libspgemm_hip_$v.so; fi
    extra="--no-host-api"; if [ $wl = synth_1m_16 ] && [ $v = product ]; then extra=""; fi
    timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline $extra > gpurun_out/pk_${wl}_$v.json 2> gpurun_out/pk_${wl}_$v.err; echo "$wl $v exit=$?"
  done
done
unset SPGEMM_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/pk_*.json")):
    try:
        d=json.load(open(f)); print(f, d["ms_per_step"], "ms", d["value"], "GFLOP/s", d.get("parity","")[:8], d["roofline"]["phases_avg_ms"], {k:v for k,v in d["roofline"]["all_kernels_avg_ms"].items() if k.startswith("k_num") or k.startswith("k_row")}, d.get("host_api"))
    except Exception as e: print(f, "failed", e)
PY
