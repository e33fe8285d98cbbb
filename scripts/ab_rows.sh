#!/bin/bash
# A/B of library variants on the row layout at world 1 through a 1-rank RCCL communicator:  bash scripts/ab_rows.sh <variant names ...>   ("-" = the shipped library)
for v in "$@"; do
  if [ "$v" = "-" ]; then unset DRX_HOST_SANITIZER_LIB; else export DRX_HOST_SANITIZER_LIB=drecpy_amd/csrc/build/libdrx_$v.so; fi
  DRX_BENCH_RCCL1=1 python bench.py --force-sharded --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k[:14]: round(x,3) for k,x in d['phases_ms'].items()})"
done
