#!/bin/bash
OUT=gpurun_out/r03ab; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_gpu_misc.py tests/test_gpu_cdae.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_$rep.json 2>> $OUT/bench.err
done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
export TMPDIR=/tmp; ROOT=$PWD; cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/kt -o kt -- python3 $ROOT/bench.py --steps 200 --warmup 10 --windows 2 --no-cpu-baseline --no-hr --no-configs > /dev/null 2> $ROOT/$OUT/kt.err
python3 - $ROOT/$OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'drx' in r['Name']: print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), round(float(r['AverageNs']) / 1e3, 1))
PY
