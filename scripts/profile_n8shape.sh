#!/bin/bash
# rocprofv3 kernel stats of ONE rank's work of an 8-rank column-sharded job (K/N = 16 columns, global batch 524 288), lists built in turns
set -u
OUT=$PWD/gpurun_out/n8shape; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD
cd /tmp
DRX_BENCH_RCCL1=1 DRX_BENCH_EMULATE_RANKS=8 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --prepare turns --force-columns --k 16 --batch 524288 --no-hr --no-cpu-baseline --steps 60 --warmup 10 --windows 2 > $OUT/bench.json 2> $OUT/kt.err
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
find $OUT -name '*kernel_trace.csv' -delete
head -16 $OUT/kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
