#!/bin/bash
# rocprofv3 kernel-trace stats of DMF and Caser steps at ml-1m shape.  Usage (gpurun): bash scripts/profile_models.sh <tag>
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_models
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in "dmf 256" "dmf 4096" "caser 512" "caser 4096"; do
  set -- $cfg
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1_$2 -o kt -- python3 $ROOT/scripts/prof_models.py $1 $2 > $OUT/$1_$2.txt 2> $OUT/$1_$2.err
  cp $(find $OUT/$1_$2 -name '*kernel_stats.csv' | head -1) $OUT/${1}_B$2_kernel_stats.csv
  find $OUT/$1_$2 -name '*kernel_trace.csv' -delete
  cat $OUT/$1_$2.txt
  head -12 $OUT/${1}_B$2_kernel_stats.csv | cut -c1-160
done
