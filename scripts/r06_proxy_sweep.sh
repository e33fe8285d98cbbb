#!/bin/bash
# r06 item 1(a): the row layout on the one-GPU all-remote proxy (1-rank RCCL communicator, every row through it) at per-rank batches of
# 65 536 / 131 072 / 262 144 and 1 / 2 / 4 micro-batches, beside the single-GPU step at the same batches.  Usage: gpurun -- bash scripts/r06_proxy_sweep.sh <tag>
set -u
TAG=${1:-r06a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
COMMON="--steps 100 --warmup 10 --windows 3 --no-cpu-baseline --no-hr --no-configs"
for B in 65536 131072 262144; do
  python bench.py $COMMON --batch $B > $OUT/single_B$B.json 2> $OUT/single_B$B.err
  for M in 1 2 4; do
    DRX_BENCH_RCCL1=1 python bench.py $COMMON --batch $B --force-sharded --no-self-bypass --micro $M > $OUT/remote_B${B}_m$M.json 2> $OUT/remote_B${B}_m$M.err
  done
  DRX_BENCH_RCCL1=1 python bench.py $COMMON --batch $B --force-sharded > $OUT/bypass_B$B.json 2> $OUT/bypass_B$B.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), 'ms', d.get('host_issue_ms_per_step'), {k[:28]: round(v, 4) for k, v in (d.get('phases_ms') or {}).items()})
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
