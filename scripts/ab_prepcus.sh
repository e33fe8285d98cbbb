#!/bin/bash
# A/B of the run-ahead streams' CU slice at the headline workload (and the ml-1m shape):  bash scripts/ab_prepcus.sh <lib variant or -> <cus per xcd ...>
v=$1; shift
if [ "$v" = "-" ]; then unset DRX_HOST_SANITIZER_LIB; else export DRX_HOST_SANITIZER_LIB=drecpy_amd/csrc/build/libdrx_$v.so; fi
for wl in synth-10m ml-1m; do for c in "$@"; do
  python bench.py --workload $wl --steps 300 --warmup 20 --no-cpu-baseline --no-hr --no-configs --prep-cus $c 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v $wl prep-cus=$c', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k[:12]: round(x,3) for k,x in d['phases_ms'].items()})"
done; done
