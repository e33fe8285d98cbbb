#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03g}
mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -15 $OUT/tests.log
