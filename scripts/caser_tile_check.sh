#!/bin/bash
# Caser tests + kernel stats of 100 Caser steps at ml-1m shape (B = 4096 and 512).  Usage (gpurun): bash scripts/caser_tile_check.sh <tag>
set -u
TAG=${1:-r05v}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_caser
mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_caser.py tests/test_gpu_baseline_shapes.py -k "caser or Caser" -x -q -m gpu > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log
export TMPDIR=/tmp
cd /tmp
for B in 4096 512; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/caser_$B -o kt -- python3 $ROOT/scripts/prof_models.py caser $B > $OUT/caser_$B.txt 2> $OUT/caser_$B.err
  cp $(find $OUT/caser_$B -name '*kernel_stats.csv' | head -1) $OUT/caser_B${B}_kernel_stats.csv
  find $OUT/caser_$B -name '*kernel_trace.csv' -delete
  cat $OUT/caser_$B.txt
  head -8 $OUT/caser_B${B}_kernel_stats.csv | cut -c1-150
done
