#!/bin/bash
# r05: parity of the streamed reduction + A/B of variants at the headline workload:  bash scripts/r05_a.sh <tag> <variants...>
TAG=$1; shift
mkdir -p gpurun_out/$TAG
timeout -k 5 600 python -m pytest tests/test_gpu_cdae.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_baseline_shapes.py -x -q -m gpu -p no:cacheprovider > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/$TAG/tests.log
bash scripts/ab_headline.sh "$@" > gpurun_out/$TAG/ab.log 2>&1
cat gpurun_out/$TAG/ab.log
