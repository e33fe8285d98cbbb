#!/bin/bash
# Same-box A/B of the in-tree libdrx.so against a variant build (boxes differ by +-3 %, so only alternating runs on ONE box compare):
#   bash scripts/build_variant.sh prev ""            (e.g. from a stash of the previous sources)
#   gpurun -- 'bash scripts/ab_variant.sh prev'      -> bench lines new / variant, three alternating runs each
set -u
V=${1:?variant name (drecpy_amd/csrc/build/libdrx_<name>.so)}
OUT=gpurun_out/ab_$V; mkdir -p $OUT
for rep in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_new_$rep.json 2>> $OUT/bench.err
  DRX_HOST_SANITIZER_LIB=$PWD/drecpy_amd/csrc/build/libdrx_$V.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_${V}_$rep.json 2>> $OUT/bench.err
done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
