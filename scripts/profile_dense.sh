#!/bin/bash
# per-kernel time of the reference-mode step (rocprofv3 --kernel-trace --stats on scripts/run_dense_steps.py) at both MovieLens shapes
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
for sh in ml-1m ml-100k; do
  rm -rf /tmp/d_$sh
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/d_$sh -o kt -- python3 $ROOT/scripts/run_dense_steps.py $sh > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/d_$sh/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    if int(r["Calls"])>=290:
        per=float(r["TotalDurationNs"])/300/1e3; tot+=per
        print("$sh", r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%.1f us/step"%per)
print("$sh total %.1f us/step"%tot)
PY
  [ -n "${1:-}" ] && cp $(find /tmp/d_$sh -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/${1}_dense_step_${sh}_kernel_stats.csv
done
