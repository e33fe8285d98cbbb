#!/bin/bash
# Caser tests + kernel stats of 100 Caser steps at ml-1m shape (B = 4096 and 512) + phase stamps when the stamps variant exists.
# Usage (gpurun): bash scripts/caser_tile_check.sh <tag>
set -u
TAG=${1:-r05v}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_caser
mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_caser.py tests/test_gpu_baseline_shapes.py -k "caser or Caser" -x -q -m gpu > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
if [ -f drecpy_amd/csrc/build/libdrx_stamps.so ]; then
  DRX_HOST_SANITIZER_LIB=$ROOT/drecpy_amd/csrc/build/libdrx_stamps.so timeout -k 5 300 python scripts/stamps_caser.py > $OUT/stamps.json 2> $OUT/stamps.err
  python - <<P
import json
d = json.load(open('$OUT/stamps.json'))
print({k: (v['mean'] if isinstance(v, dict) and 'mean' in v else v) for k, v in d.items()})
P
fi
export TMPDIR=/tmp
cd /tmp
for B in 4096 512; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/caser_$B -o kt -- python3 $ROOT/scripts/prof_models.py caser $B > $OUT/caser_$B.txt 2> $OUT/caser_$B.err
  cp $(find $OUT/caser_$B -name '*kernel_stats.csv' | head -1) $OUT/caser_B${B}_kernel_stats.csv
  find $OUT/caser_$B -name '*kernel_trace.csv' -delete
  cat $OUT/caser_$B.txt
  head -4 $OUT/caser_B${B}_kernel_stats.csv | cut -c1-150
done
