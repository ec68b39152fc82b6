#!/bin/bash
# round-3 evidence run (one commit): GPU suite, bench lines of every workload, rocprofv3 kernel stats + FETCH/WRITE passes,
# SQ and TCC counters.  Everything lands under gpurun_out/; profiles/summarize.py is re-run locally after the merge.
mkdir -p gpurun_out
git rev-parse HEAD > gpurun_out/r03_commit.txt 2>/dev/null || true
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "tests exit=$?"; tail -5 gpurun_out/gpu_tests.log
for wl in synth_1m_16 web_google_surrogate synth_256k_16 synth_1m_32 rmcl_500k; do
  timeout -k 10 600 python bench.py --workload $wl --steps 10 --warmup 2 > gpurun_out/r03_bench_$wl.json 2> gpurun_out/r03_bench_$wl.err; echo "$wl exit=$?"
done
BENCH_FORCE_GROUP=1 timeout -k 10 300 python bench.py --workload synth_256k_16 --steps 5 --warmup 1 --no-cpu-baseline --no-host-api > gpurun_out/r03_bench_group_rehearsal.json 2> gpurun_out/r03_bench_group_rehearsal.err; echo "group rehearsal exit=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --workload synth_256k_16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_n2_gloo.json 2> gpurun_out/r03_bench_n2_gloo.err; echo "n2 gloo exit=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline") or {}
        print(f, d["ms_per_step"], "ms", d["value"], d["unit"], d.get("parity","")[:20], "frac", r.get("frac"), "pipe", d.get("pipeline_frac_of_hbm_peak"), "host_api", (d.get("host_api") or {}).get("ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d.get("transport"))
    except Exception as e: print(f, "failed", e)
PY
bash profiles/collect.sh r03 synth_1m_16
bash profiles/collect.sh r03 web_google_surrogate
bash profiles/collect.sh r03 synth_256k_16
OUT=$PWD/gpurun_out/prof_r03_rmcl_500k; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload rmcl_500k --steps 2 --warmup 1 --no-verify --no-cpu-baseline > $OUT/trace.log 2>&1; echo "rmcl trace exit=$?" )
bash tools/pmc_sq.sh synth_1m_16
bash tools/pmc_tcc.sh synth_1m_16
cp gpurun_out/tcc_summary.txt gpurun_out/r03_tcc_counters_1m.txt
python3 tools/pmc_sq.py > gpurun_out/r03_sq_counters_1m.txt 2>&1; tail -30 gpurun_out/r03_sq_counters_1m.txt
