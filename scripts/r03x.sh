#!/bin/bash
# the sort alone: per-kernel times (kernel trace) at the touch list's size
OUT=$PWD/gpurun_out/r03x; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD
python scripts/bench_sort.py 2>&1 | tail -6
cd /tmp
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/bench_sort.py > /dev/null 2> $OUT/kt.err
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us avg', round(float(r['MinNs'])/1e3,1), round(float(r['MaxNs'])/1e3,1))
PY
