#!/bin/bash
# r06 item 5: the DMF dense layers on the matrix cores (k_dmf_dense_tile) — parity suite, then kernel stats of the step at B = 4096 / 256
set -u
TAG=${1:-r06i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_dmf.py tests/test_gpu_baseline_shapes.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for B in 4096 256; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dmf$B -o p -- python3 $GRAFT_REPO_ROOT/scripts/prof_models.py dmf $B > $OUT/dmf$B.txt 2>&1
  cp $(find $OUT/dmf$B -name '*kernel_stats.csv' | head -1) $OUT/dmf_B${B}_kernel_stats.csv
  find $OUT/dmf$B -name '*kernel_trace.csv' -delete
  grep "drx::" $OUT/dmf_B${B}_kernel_stats.csv | awk -F'",' '{print $1, $2, $4}' | cut -c1-60,200- | head -12
  grep "ms/step" $OUT/dmf$B.txt
done
cd $GRAFT_REPO_ROOT
python scripts/sort_bench.py > $OUT/sort_bench.txt 2>&1
for v in ipt4 ipt16 t256 t1024 t256ipt16; do DRX_HOST_SANITIZER_LIB=drecpy_amd/csrc/build/libdrx_sort_$v.so python scripts/sort_bench.py >> $OUT/sort_bench.txt 2>&1; done
grep "sort of" $OUT/sort_bench.txt
python scripts/copy_bench.py > $OUT/copy_bench.txt 2>&1; cat $OUT/copy_bench.txt
