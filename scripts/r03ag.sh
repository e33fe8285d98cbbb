#!/bin/bash
OUT=gpurun_out/r03ag; mkdir -p $OUT
for rep in 1 2; do for x in 0 1 2 4; do
  DRX_SAMPLE_AHEAD_EXTRA=$x python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_x${x}_$rep.json 2>> $OUT/bench.err
  DRX_SAMPLE_AHEAD_EXTRA=$x python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_short_x${x}_$rep.json 2>> $OUT/bench.err
done; done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
tail -2 $OUT/bench.err
