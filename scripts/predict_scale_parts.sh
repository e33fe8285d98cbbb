#!/bin/bash
# Like predict_scale.sh, with the touch list built in parts (DRX_BENCH_EMULATE_PARTS: this GPU sorts part 0 of N afresh every
# step, the other parts come from a cache and a device copy stands in for the all-gather).  Batches cycle (8 distinct ones).
set -u
echo "["
first=1
for N in 2 4 8; do
  [ $first = 1 ] || echo ","
  first=0
  DRX_BENCH_RCCL1=1 DRX_BENCH_EMULATE_PARTS=$N python bench.py --force-columns --k $((128 / N)) --batch $((65536 * N)) --no-hr --no-cpu-baseline --steps 60 --warmup 10 | head -1
done
echo "]"
