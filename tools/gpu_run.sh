#!/bin/bash
# One parameterised GPU runner (replaces the per-experiment scripts of round 3).
#   tools/gpu_run.sh tests [pytest args...]            the GPU suite (default: -m gpu -x -q)
#   tools/gpu_run.sh ab "<env A>" "<env B>" [workloads...]   A/B of two environments (e.g. "SPGEMM_CHAIN=0" "SPGEMM_CHAIN=1"),
#                                                      REPS processes per arm (default 5), medians printed
#   tools/gpu_run.sh bench <workload> [bench args...]   one bench line into gpurun_out/bench_<workload>.json
set -o pipefail
mkdir -p gpurun_out
mode="$1"; shift
case "$mode" in
  tests)
    timeout -k 10 1100 python -m pytest ${@:-tests -m gpu -x -q} 2>&1 | tail -25 ;;
  bench)
    wl="$1"; shift
    timeout -k 10 500 python bench.py --workload "$wl" "$@" > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err || { tail -5 gpurun_out/bench_$wl.err; exit 1; }
    python tools/bench_summary.py gpurun_out/bench_$wl.json ;;
  ab)
    A="$1"; B="$2"; shift 2
    WLS=${@:-"synth_1m_16"}
    REPS=${REPS:-5}
    for wl in $WLS; do
      for rep in $(seq 1 $REPS); do
        for arm in A B; do
          if [ $arm = A ]; then envs="$A"; else envs="$B"; fi
          env $envs timeout -k 10 300 python bench.py --workload $wl --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-host-api \
            > gpurun_out/ab_${wl}_${arm}_$rep.json 2> gpurun_out/ab_${wl}_${arm}_$rep.err || { echo "$wl $arm $rep failed"; tail -3 gpurun_out/ab_${wl}_${arm}_$rep.err; exit 1; }
        done
      done
    done
    python tools/bench_summary.py --ab "$A" "$B" gpurun_out/ab_*.json ;;
  *) echo "unknown mode $mode"; exit 2 ;;
esac
