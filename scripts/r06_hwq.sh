#!/bin/bash
# GPU_MAX_HW_QUEUES (ROCclr: hardware queues per process, default 4): more queues = no two of a job's streams on one queue?
set -u
TAG=${1:-r06ce}
OUT=gpurun_out/$TAG
mkdir -p $OUT
COMMON="--steps 200 --warmup 20 --windows 3 --no-cpu-baseline --no-hr --no-configs"
for Q in default 8 16; do
  E=""; [ $Q != default ] && export GPU_MAX_HW_QUEUES=$Q
  for i in 1 2; do python bench.py $COMMON > $OUT/head_q${Q}_$i.json 2> $OUT/head_q${Q}_$i.err; done
  DRX_BENCH_RCCL1=1 python bench.py --steps 100 --warmup 10 --windows 3 --no-cpu-baseline --no-hr --no-configs --force-sharded --no-self-bypass --chunks 2 > $OUT/rows_q${Q}.json 2> $OUT/rows_q${Q}.err
  echo "== GPU_MAX_HW_QUEUES=$Q: seven used dummy streams, then DMF device fit B=256 (probe off)"
  DRX_STREAM_PROBE=0 python scripts/r06_stream_parity.py 6 -1 use 2>&1 | tail -1 | cut -c1-120
  DRX_STREAM_PROBE=0 python scripts/r06_stream_parity.py 10 -1 use 2>&1 | tail -1 | cut -c1-120
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), round(d['value'] / 1e6, 2), 'M/s', round(d['ms_per_step'], 4), 'ms')
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-400:])
PY
