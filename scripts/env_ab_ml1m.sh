#!/bin/bash
# run-ahead settings at the ml-1m-shaped sampled step and at the headline, one box:  bash scripts/env_ab_ml1m.sh
run() { python bench.py --workload $W --steps $S --warmup 10 --no-cpu-baseline --no-hr --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$W $1', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k[:12]: round(x,3) for k,x in d['phases_ms'].items()})"; }
for W in ml-1m synth-10m; do
  S=100; [ $W = synth-10m ] && S=200
  run base
  DRX_SIDE_PRIORITY=0 run prio0
  run base
  DRX_SIDE_PRIORITY=0 run prio0
done
