#!/bin/bash
set -u
TAG=${1:-r06cg}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for i in 1 2 3 4; do for C in 1 2; do
  DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py --steps 100 --warmup 10 --windows 3 --no-cpu-baseline --no-hr --no-configs --force-sharded --no-self-bypass --chunks $C > $OUT/remote_c${C}_$i.json 2> $OUT/remote_c${C}_$i.err
done; done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), round(d['value'] / 1e6, 2), 'M/s', round(d['ms_per_step'], 4), 'ms', d['window_ms'], [round(v, 3) for v in d['phases_ms'].values()])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-400:])
PY
