#!/bin/bash
# final round artifacts: profile round (bench, kernel stats, PMC), model kernel stats, full GPU suite
bash scripts/profile_round.sh r03_final > gpurun_out/r03_final_profile.log 2>&1
bash scripts/profile_models.sh r03 > gpurun_out/r03_models.log 2>&1
timeout -k 5 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r03_final/gpu_tests.log 2>&1; echo "full suite rc=$?"; tail -2 gpurun_out/r03_final/gpu_tests.log
tail -3 gpurun_out/r03_final_profile.log
