#!/bin/bash
OUT=$PWD/gpurun_out/r03am; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD
cd /tmp
DRX_BENCH_RCCL1=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --force-sharded --steps 100 --warmup 10 --windows 2 --no-cpu-baseline --no-hr --no-configs > $OUT/bench.json 2> $OUT/kt.err
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/kt/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:28]:
        print(r['Name'][:86].ljust(86), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(8), 'us', ('%.1f' % (float(r['TotalDurationNs']) / 1e6)).rjust(8), 'ms')
PY
