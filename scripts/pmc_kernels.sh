#!/bin/bash
# Stall / cache / fabric counters of the sparse-step kernels running ALONE (scripts/exp_events.py none: batches and touch lists prepared
# up front, nothing beside the training stream).  Separate rocprofv3 --pmc passes.  Usage (gpurun): bash scripts/pmc_kernels.sh <tag>
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  STEPS=40 timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python3 $ROOT/scripts/exp_events.py none > $OUT/p$i.txt 2> $OUT/p$i.err
  tail -2 $OUT/p$i.err | cut -c1-200
done
cd $ROOT
python profiles/pmc_generic.py $OUT/summary.json $(find $OUT -name '*counter_collection.csv')
find $OUT -name '*.csv' -size +2M -delete
