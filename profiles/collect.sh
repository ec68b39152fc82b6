#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  profiles/collect.sh <tag> <workload>
# Produces under gpurun_out/prof_<tag>/: kernel-trace stats, and two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not
# fit one pass on gfx950).  PMC passes use --kernel-trace only (never sys/hip/hsa trace together with --pmc).
TAG=${1:-r04}; WL=${2:-synth_1m_16}
OUT=$PWD/gpurun_out/prof_${TAG}_${WL}; rm -rf $OUT; mkdir -p $OUT
ARGS="$PWD/bench.py --workload $WL --steps 5 --warmup 2 --no-verify --no-cpu-baseline --no-host-api --no-other-workloads"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1; echo "trace exit=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1; echo "fetch exit=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1; echo "write exit=$?"
cd - > /dev/null
python3 profiles/summarize.py $OUT $TAG $WL   # on the box (bench.py reads the traffic file); re-run it locally after the merge,
# only gpurun_out/ travels back
