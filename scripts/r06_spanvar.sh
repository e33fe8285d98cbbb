#!/bin/bash
# k_span_planned: more workgroups for the short spans (a group walks its spans one after the other, 3 - 4 dependent round trips each)
set -u
TAG=${1:-r06aw}
OUT=gpurun_out/$TAG
mkdir -p $OUT
COMMON="--steps 200 --warmup 20 --windows 3 --no-cpu-baseline --no-hr --no-configs"
for i in 1 2; do
  python bench.py $COMMON > $OUT/default_$i.json 2> $OUT/default_$i.err
  for v in s512 s256 l64 l1024; do
    DRX_HOST_SANITIZER_LIB=drecpy_amd/csrc/build/libdrx_span_$v.so python bench.py $COMMON > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err
  done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), round(d['value'] / 1e6, 2), 'M/s', round(d['ms_per_step'], 4), 'ms', [round(v, 4) for v in (d.get('phases_ms') or {}).values()])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-600:])
PY
