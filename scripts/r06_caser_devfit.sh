#!/bin/bash
# kernel table of Caser.fit(device_sampler=True) at the ml-1m shape, B = 4096
set -u
TAG=${1:-r06ae}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o devfit -- python3 $GRAFT_REPO_ROOT/scripts/prof_caser_device_fit.py > $OUT/devfit.txt 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/caser_device_fit_kernel_stats.csv 2>/dev/null
find $OUT/prof -name '*kernel_trace.csv' -delete
head -24 $OUT/caser_device_fit_kernel_stats.csv | cut -c1-150; tail -3 $OUT/devfit.txt
