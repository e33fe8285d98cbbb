#!/bin/bash
# One GPU-box call of round 3: the whole GPU suite, then A/B bench lines of library variants / environment switches.
#   bash scripts/r03_ab.sh <tag> "<variant>:<env>" ...      variant = name under drecpy_amd/csrc/build/libdrx_<name>.so or "default"
set -u
TAG=${1:-r03x}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  python -m pytest tests -x -q -m gpu ${PYTEST_ARGS:-} > $OUT/tests.log 2>&1
  echo "tests rc=$?" | tee -a $OUT/tests.log
  tail -4 $OUT/tests.log
fi
for rep in 1 2; do
  for spec in "$@"; do
    var=${spec%%:*}; envs=${spec#*:}
    lib=""; [ "$var" != "default" ] && lib="DRX_HOST_SANITIZER_LIB=$PWD/drecpy_amd/csrc/build/libdrx_$var.so"
    env $lib $envs python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr ${BENCH_ARGS:-} > "$OUT/bench_${var}_${envs//[^A-Za-z0-9=]/_}_$rep.json" 2>> $OUT/bench.err
  done
done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -5 $OUT/bench.err
