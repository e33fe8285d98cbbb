#!/bin/bash
set -u
TAG=${1:-r06k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
python scripts/copy_bench.py > $OUT/copy_bench.txt 2>&1; cat $OUT/copy_bench.txt
timeout -k 5 2400 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -15 $OUT/gpu_tests.log
