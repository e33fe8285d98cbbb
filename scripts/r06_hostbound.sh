#!/bin/bash
# the HOST cost of a row-sharded step: a batch so small (1024 triples) that the device is never the bound — ms per step = host time per step
set -u
TAG=${1:-r06cc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
COMMON="--steps 300 --warmup 30 --windows 3 --no-cpu-baseline --no-hr --no-configs --batch 1024 --users 400000"
for C in 2; do
  for M in thread nothread; do
    F=""; [ $M = nothread ] && F="--no-comm-thread"
    DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --no-self-bypass --chunks $C $F > $OUT/small_c${C}_$M.json 2> $OUT/small_c${C}_$M.err
  done
done
for i in 1 2; do for C in 1 2; do for M in thread nothread; do
  F=""; [ $M = nothread ] && F="--no-comm-thread"
  DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py --steps 100 --warmup 10 --windows 3 --no-cpu-baseline --no-hr --no-configs --force-sharded --no-self-bypass --chunks $C $F > $OUT/full_c${C}_${M}_$i.json 2> $OUT/full_c${C}_${M}_$i.err
done; done; done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), round(d['ms_per_step'], 4), 'ms/step', d.get('host_issue_ms_per_step'), d['config'].get('exchanges_issued_by'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-800:])
PY
