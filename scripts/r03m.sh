#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03m}
mkdir -p $OUT
for g in 64 128 192 256 352; do
  DRX_SORT_GRID=$g python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_grid${g}.json 2>> $OUT/bench.err
done
DRX_PREP_AHEAD=4 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_ahead4.json 2>> $OUT/bench.err
DRX_PREP_AHEAD=2 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_ahead2.json 2>> $OUT/bench.err
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
    except Exception as e:
        print(f, 'ERR', e)
PY
