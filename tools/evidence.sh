#!/bin/bash
# The evidence run of a round, on the GPU box (through gpurun), from ONE commit:  tools/evidence.sh <tag>
#   bench lines of every workload (the default line with other_workloads as the driver runs it), the group rehearsal of
#   the N>1 lines on one rank, the opt-in paths (SPGEMM_PATH=1|2), kernel traces + FETCH/WRITE passes (profiles/collect.sh).
# Everything lands under gpurun_out/; tools/refresh_profiles.sh <tag> condenses it into profiles/ (run locally afterwards).
TAG=${1:-r04}
mkdir -p gpurun_out
git rev-parse HEAD > gpurun_out/${TAG}_commit.txt 2>/dev/null || sha256sum sparse_matrix_with_flops_amd/libspgemm_hip.so.buildinfo > gpurun_out/${TAG}_commit.txt
B() { name=$1; shift; timeout -k 10 600 python bench.py "$@" > gpurun_out/${TAG}_bench_$name.json 2> gpurun_out/${TAG}_bench_$name.err || { echo "bench $name failed"; tail -5 gpurun_out/${TAG}_bench_$name.err; exit 1; }; python tools/bench_summary.py gpurun_out/${TAG}_bench_$name.json | head -2; }
B default
for wl in synth_256k_16 web_google_surrogate synth_1m_32 rmcl_500k; do B $wl --workload $wl; done
BENCH_FORCE_GROUP=1 B group_rehearsal_one_rank --no-cpu-baseline --no-host-api --no-other-workloads
BENCH_FORCE_GROUP=1 B rmcl_group_rehearsal_one_rank --workload rmcl_500k --steps 5 --warmup 2 --no-cpu-baseline --no-host-api
SPGEMM_PATH=1 B path1_wbatch_two_pass --no-cpu-baseline --no-host-api --no-other-workloads
SPGEMM_PATH=2 B path2_one_pass --no-cpu-baseline --no-host-api --no-other-workloads
SPGEMM_PATH=1 B web_path1 --workload web_google_surrogate --no-cpu-baseline --no-host-api
SPGEMM_PATH=2 B web_path2 --workload web_google_surrogate --no-cpu-baseline --no-host-api
for wl in synth_1m_16 web_google_surrogate synth_256k_16; do profiles/collect.sh $TAG $wl || exit 1; done
# kernel trace of the R-MCL loop
OUT=$PWD/gpurun_out/prof_${TAG}_rmcl_500k; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OLDPWD/bench.py --workload rmcl_500k --steps 3 --warmup 1 --no-verify --no-cpu-baseline --no-host-api > $OUT/trace.log 2>&1; echo "rmcl trace exit=$?" )
# SQ counters of the headline workload (three --pmc passes, kernel-trace only) -> gpurun_out/<tag>_sq_counters_1m.txt
bash tools/pmc_sq.sh synth_1m_16 && python tools/pmc_sq.py gpurun_out/pmc_sq > gpurun_out/${TAG}_sq_counters_1m.txt
