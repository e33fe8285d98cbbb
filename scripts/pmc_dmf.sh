#!/bin/bash
# Issue / stall counters of the DMF step kernels (100 steps at the ml-1m shape, B = 4096).  Separate rocprofv3 --pmc passes.
# Usage (gpurun): bash scripts/pmc_dmf.sh <tag>
set -u
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_pmc_dmf
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_WAVE32_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python3 $ROOT/scripts/prof_models.py dmf 4096 > $OUT/p$i.txt 2> $OUT/p$i.err
  tail -2 $OUT/p$i.err | cut -c1-200
done
cd $ROOT
python - > $OUT/summary.txt <<P
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        kn = [x for x in ('k_dmf_k0_update', 'k_dmf_gather', 'k_dmf_dense_tile') if x in r['Kernel_Name']]
        if not kn: continue
        a = acc[(kn[0], r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
for k in sorted(acc): print(f'{k[0]:18s} {k[1]:28s} {acc[k][0] / acc[k][1]:14.1f}  ({acc[k][1]} dispatches)')
P
cat $OUT/summary.txt
find $OUT -name '*.csv' -size +2M -delete
