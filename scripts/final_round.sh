#!/bin/bash
# The artifacts of a round in ONE GPU-box call: profile round (bench line, kernel stats, PMC passes), model kernel stats, full GPU suite.
# Usage (gpurun): bash scripts/final_round.sh <tag>     then HERE: copy gpurun_out/<tag>_final/* to profiles/<tag>_* and
# gpurun_out/<tag>_final/pmc_traffic.json to profiles/pmc_traffic.json (tests/test_bench_launcher.py checks its kernel-source hash)
TAG=${1:-r03}
mkdir -p gpurun_out
bash scripts/profile_round.sh ${TAG}_final > gpurun_out/${TAG}_final_profile.log 2>&1
bash scripts/profile_models.sh ${TAG} > gpurun_out/${TAG}_models.log 2>&1
timeout -k 5 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_final/gpu_tests.log 2>&1; echo "full suite rc=$?"; tail -2 gpurun_out/${TAG}_final/gpu_tests.log
tail -3 gpurun_out/${TAG}_final_profile.log
