#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03p}
mkdir -p $OUT
python -m pytest tests/test_gpu_cdae.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_fit.py tests/test_gpu_kshard.py tests/test_gpu_baseline_shapes.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
for rep in 1 2 3; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_default_$rep.json 2>> $OUT/bench.err; done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()], 'frac', round(d['roofline']['frac'], 3), round(d['roofline']['whole_step_frac'], 3))
    except Exception as e:
        print(f, 'ERR', e)
PY
bash scripts/r03j.sh ${1:-r03p}_kt | grep "drx::"
