#!/bin/bash
# is the preparation (side stream) the pipeline's bound?  training kernels alone (lists cached) against the normal run
OUT=gpurun_out/r03w; mkdir -p $OUT
for rep in 1 2; do
  DRX_BENCH_CACHE_PREP=1 python bench.py --presampled --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_cached_$rep.json 2>> $OUT/bench.err
  python bench.py --presampled --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_presampled_$rep.json 2>> $OUT/bench.err
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_fresh_$rep.json 2>> $OUT/bench.err
done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
tail -3 $OUT/bench.err
