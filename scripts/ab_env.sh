#!/bin/bash
# Same-box A/B of an environment switch: bash scripts/ab_env.sh NAME value1 value2 ...   (three alternating bench runs per value)
set -u
NAME=$1; shift
OUT=gpurun_out/ab_env_$NAME; mkdir -p $OUT
for rep in 1 2 3; do for v in "$@"; do
  env $NAME=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_${v}_$rep.json 2>> $OUT/bench.err
done; done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
