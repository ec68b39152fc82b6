#!/bin/bash
# SQ-side counters for one workload (separate --pmc passes, kernel-trace only); summarised by tools/pmc_sq.py
WL=${1:-synth_1m_16}
OUT=$PWD/gpurun_out/pmc_sq; rm -rf $OUT; mkdir -p $OUT
ARGS="$PWD/bench.py --workload $WL --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-host-api --no-other-workloads"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1; echo "pass $i exit=$?"
done
