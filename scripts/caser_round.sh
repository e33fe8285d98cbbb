#!/bin/bash
# The Caser kernel's evidence in one GPU-box call: tests + phase stamps + kernel stats (caser_tile_check.sh), issue / stall counters
# (pmc_caser.sh), the fp32-MFMA micro-benchmark (mb/mb_mfma32.hip), the host profile of a step.  Usage (gpurun): bash scripts/caser_round.sh <tag>
set -u
TAG=${1:-r05}
ROOT=$(pwd)
bash scripts/build_variant.sh stamps "-DDRX_STAMPS" > /dev/null 2>&1
bash scripts/caser_tile_check.sh $TAG
bash scripts/pmc_caser.sh $TAG | tail -40
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_mfma32 scripts/mb/mb_mfma32.hip 2>/dev/null && /tmp/mb_mfma32 > gpurun_out/${TAG}_caser/mb_mfma32.log
cat gpurun_out/${TAG}_caser/mb_mfma32.log
python scripts/caser_host_profile.py 2>&1 | tail -40 > gpurun_out/${TAG}_caser/host_profile.txt
head -12 gpurun_out/${TAG}_caser/host_profile.txt
