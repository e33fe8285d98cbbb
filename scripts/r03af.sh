#!/bin/bash
OUT=gpurun_out/r03af; mkdir -p $OUT
DRX_FWD_STRATA=15 timeout -k 5 300 python -m pytest tests/test_gpu_cdae.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1
for rep in 1 2 3; do for sarg in 0 4 15 64; do
  DRX_FWD_STRATA=$sarg python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_s${sarg}_$rep.json 2>> $OUT/bench.err
done; done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
