#!/bin/bash
# round 3: refresh of the R-MCL evidence alone (host-side change of the loop; device code as in gpu_r3c.sh's run)
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --workload rmcl_500k --steps 10 --warmup 2 > gpurun_out/r03_bench_rmcl_500k.json 2> gpurun_out/r03_bench_rmcl_500k.err; echo "rmcl_500k exit=$?"
OUT=$PWD/gpurun_out/prof_r03_rmcl_500k; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload rmcl_500k --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-host-api > $OUT/trace.log 2>&1; echo "rmcl trace exit=$?" )
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_rmcl_500k.json"))
print(d["ms_per_step"], d["value"], d["loop_one_call_per_iteration_ms"], d["host_api"], d["parity"][:10], d["roofline"]["frac"], d["pipeline_frac_of_hbm_peak"])
PY
