#!/bin/bash
set -u
TAG=${1:-r06az}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_dmf.py tests/test_gpu_baseline_shapes.py -q -m gpu -x > $OUT/pytest.log 2>&1; grep -n "passed\|failed\|Error" $OUT/pytest.log | tail -5
python scripts/r06_dmf_dev256.py DMF,ModifiedDMF 256 2>&1 | grep steady | cut -c1-70
python scripts/r06_dmf_dev256.py DMF 4096 2>&1 | grep steady | cut -c1-70
python scripts/r06_dmf_host_profile.py 256 device 2>&1 | tail -50 > $OUT/dmf_host_profile_B256.txt; head -12 $OUT/dmf_host_profile_B256.txt | cut -c1-150
