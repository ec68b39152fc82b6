#!/bin/bash
# memory-side counters for the 1M workload (separate --pmc passes, kernel-trace only)
WL=${1:-synth_1m_16}
OUT=$PWD/gpurun_out/pmc_mem; rm -rf $OUT; mkdir -p $OUT
ARGS="$PWD/bench.py --workload $WL --steps 2 --warmup 1 --no-verify --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1; echo "pass $i exit=$?"
done
