#!/bin/bash
# A/B of bench.py flag sets at the headline workload and the ml-1m shape:  bash scripts/ab_flags.sh "<flags A>" "<flags B>" ...
for wl in synth-10m ml-1m; do for f in "$@"; do
  python bench.py --workload $wl --steps 300 --warmup 20 --no-cpu-baseline --no-hr --no-configs $f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$wl [$f]', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k[:12]: round(x,3) for k,x in d['phases_ms'].items()})"
done; done
