#!/bin/bash
OUT=gpurun_out/r03ad; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_gpu_misc.py tests/test_gpu_cdae.py tests/test_gpu_fullsize.py tests/test_gpu_fit.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_new_$rep.json 2>> $OUT/bench.err
  DRX_HOST_SANITIZER_LIB=$PWD/drecpy_amd/csrc/build/libdrx_prev.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_prev_$rep.json 2>> $OUT/bench.err
done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
