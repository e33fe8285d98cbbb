#!/bin/bash
OUT=gpurun_out/r03ai; mkdir -p $OUT
DRX_DRAW_STREAM=1 timeout -k 5 300 python -m pytest tests/test_gpu_fit.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1
for rep in 1 2 3; do for n in 0 1; do
  DRX_DRAW_STREAM=$n python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_d${n}_$rep.json 2>> $OUT/bench.err
  DRX_DRAW_STREAM=$n python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_short_d${n}_$rep.json 2>> $OUT/bench.err
done; done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
tail -2 $OUT/bench.err
