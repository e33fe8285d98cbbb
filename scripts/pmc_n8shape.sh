#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the column path at the per-rank shape of an 8-GPU job.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_n8; mkdir -p $OUT
export TMPDIR=/tmp DRX_BENCH_RCCL1=1
ARGS="--force-columns --k 16 --batch 524288 --no-hr --no-cpu-baseline --steps 6 --warmup 2"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/w.err
cd $ROOT
python profiles/pmc_summary.py $(find $OUT/f -name '*counter_collection.csv' | head -1) $(find $OUT/w -name '*counter_collection.csv' | head -1) $OUT/traffic.json
find $OUT -size +2M -delete
