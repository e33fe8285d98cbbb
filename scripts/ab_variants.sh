#!/bin/bash
# Same-box A/B of several variant builds against the in-tree library: bash scripts/ab_variants.sh name1 name2 ...   (3 alternating runs each;
# a variant that fails the sparse-step tests is reported and skipped)
set -u
OUT=gpurun_out/ab_variants; mkdir -p $OUT; rm -f $OUT/*.json
ok=""
for v in "$@"; do
  if DRX_HOST_SANITIZER_LIB=$PWD/drecpy_amd/csrc/build/libdrx_$v.so timeout -k 5 300 python -m pytest tests/test_gpu_cdae.py -x -q -m gpu -p no:cacheprovider > $OUT/test_$v.log 2>&1; then ok="$ok $v"; else echo "variant $v FAILS tests: $(tail -1 $OUT/test_$v.log)"; fi
done
for rep in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_default_$rep.json 2>> $OUT/bench.err
  for v in $ok; do
    DRX_HOST_SANITIZER_LIB=$PWD/drecpy_amd/csrc/build/libdrx_$v.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_${v}_$rep.json 2>> $OUT/bench.err
  done
done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    L = [l for l in open(f) if l.startswith('{')]
    if not L: print(f.split('/')[-1], 'no line'); continue
    d = json.loads(L[-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
