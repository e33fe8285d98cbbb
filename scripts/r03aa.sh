#!/bin/bash
OUT=gpurun_out/r03aa; mkdir -p $OUT
for v in t256i8 t512i4 t256i16; do
  DRX_HOST_SANITIZER_LIB=$PWD/drecpy_amd/csrc/build/libdrx_$v.so timeout -k 5 120 python -m pytest tests/test_gpu_misc.py -q -m gpu -k sort -p no:cacheprovider 2>&1 | tail -1
done
for rep in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_default_$rep.json 2>> $OUT/bench.err
  for v in t256i8 t512i4 t256i16; do
    DRX_HOST_SANITIZER_LIB=$PWD/drecpy_amd/csrc/build/libdrx_$v.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_${v}_$rep.json 2>> $OUT/bench.err
  done
done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
