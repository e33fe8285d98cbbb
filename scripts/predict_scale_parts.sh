#!/bin/bash
# Like predict_scale.sh, with the touch list NOT sorted whole on every rank (DRX_BENCH_EMULATE_RANKS=N, --prepare turns | parts):
# this GPU does what one rank of N would do and takes what the others would send from a cache, a device copy standing in for the
# broadcast / all-gather.  Batches cycle (8 distinct ones).  Usage: bash scripts/predict_scale_parts.sh [turns|parts]
set -u
MODE=${1:-turns}
echo "["
first=1
for N in 2 4 8; do
  [ $first = 1 ] || echo ","
  first=0
  DRX_BENCH_RCCL1=1 DRX_BENCH_EMULATE_RANKS=$N python bench.py --prepare $MODE --force-columns --k $((128 / N)) --batch $((65536 * N)) --no-hr --no-cpu-baseline --steps 60 --warmup 10 | head -1
done
echo "]"
