#!/bin/bash
set -u
TAG=${1:-r06ap}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
{
echo "# DMF.fit() on the reference-exact host sampler, ml-1m shape: steady ms/step (median of 6 fenced windows, min, max), three models per setting"
for M in default inline spin; do for B in 256 1024 4096; do
  echo "## DRX_HOST_PREFETCH=$M B=$B"
  DRX_HOST_PREFETCH=$( [ $M = default ] && echo spin || echo $M ) DRX_FORCE_WORKER=$( [ $M = spin ] && echo 1 || echo 0 ) python scripts/r06_dmf_host256.py $B 2>&1 | grep switch | head -3 | cut -d' ' -f5-
done; done
} > $OUT/handover.log 2>&1
cat $OUT/handover.log | cut -c1-200
