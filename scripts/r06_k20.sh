#!/bin/bash
set -u
TAG=${1:-r06bc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_cdae.py tests/test_gpu_kshard.py -q -m gpu -x > $OUT/pytest.log 2>&1; grep -n "passed\|failed" $OUT/pytest.log | tail -2
COMMON="--no-cpu-baseline --no-hr --no-configs"
for i in 1 2 3; do
  for K in 20 200; do
    python bench.py $COMMON --steps $K --warmup 5 > $OUT/k${K}_$i.json 2> $OUT/k${K}_$i.err
  done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), round(d['value'] / 1e6, 2), 'M/s', round(d['ms_per_step'], 4), 'ms', 'windows', d['window_ms'], 'timed launches', d['roofline'].get('timed_launches'), 'frac', round(d['roofline']['frac'], 4))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-600:])
PY
