#!/bin/bash
# kernel table of DMF.fit(device_sampler=True) at the ml-1m shape, B = 4096 (scripts/r06_dmf_host_profile.py without cProfile's cost would be nicer: the
# profile script's fit of 2050 steps is what rocprofv3 sees)
set -u
TAG=${1:-r06bo}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o devfit -- python3 $GRAFT_REPO_ROOT/scripts/r06_dmf_host_profile.py 4096 device > $OUT/devfit.txt 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/dmf_device_fit_B4096_kernel_stats.csv 2>/dev/null
find $OUT/prof -name '*kernel_trace.csv' -delete
python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/dmf_device_fit_B4096_kernel_stats.csv')))
tot=0
for r in rows:
    c=int(r['Calls'])
    if c>=2000:
        per=float(r['TotalDurationNs'])/2050/1e3
        tot+=per
        print(f"{per:8.1f} us/step  x{c/2050:.1f}  {r['Name'][:100]}")
print('total per step', round(tot,1))
PY
