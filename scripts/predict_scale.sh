#!/bin/bash
# One GPU runs the per-rank shape of an N-GPU column-sharded job (K/N columns of every table, the N-fold batch) through the
# column code path itself (split forward + all-reduce over a 1-rank RCCL communicator): what an N-GPU run costs per step, short of
# the latency of a real N-rank all-reduce of B floats.  Usage (through gpurun): bash scripts/predict_scale.sh > gpurun_out/scale.json
set -u
echo "["
python bench.py --no-hr --no-cpu-baseline --steps 100 --warmup 20
for N in 2 4 8; do
  echo ","
  DRX_BENCH_RCCL1=1 python bench.py --force-columns --k $((128 / N)) --batch $((65536 * N)) --no-hr --no-cpu-baseline --steps 60 --warmup 10 | head -1
done
echo "]"
