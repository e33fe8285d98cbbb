#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03j}
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/kt -o kt -- python3 $ROOT/bench.py --steps 100 --warmup 10 --windows 2 --no-cpu-baseline --no-hr --no-configs > $ROOT/$OUT/kt_bench.json 2> $ROOT/$OUT/kt.err
cd $ROOT
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -size +4M -delete
python - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), str(round(float(r['AverageNs']) / 1e3, 1)).rjust(8), 'us avg', str(round(float(r['TotalDurationNs']) / 1e6, 2)).rjust(8), 'ms total')
PY
