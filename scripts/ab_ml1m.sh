#!/bin/bash
# A/B of library variants x the shared form at the ml-1m-shaped sampled step:  bash scripts/ab_ml1m.sh <variant names ...>   ("-" = the shipped library)
for v in "$@"; do for sh in "" "--no-share-users"; do
  if [ "$v" = "-" ]; then unset DRX_HOST_SANITIZER_LIB; else export DRX_HOST_SANITIZER_LIB=drecpy_amd/csrc/build/libdrx_$v.so; fi
  python bench.py --workload ml-1m --steps 100 --warmup 10 --no-cpu-baseline --no-hr --no-configs $sh 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v $sh', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k[:12]: round(x,3) for k,x in d['phases_ms'].items()})"
done; done
