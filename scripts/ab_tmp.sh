export DRX_HOST_SANITIZER_LIB=drecpy_amd/csrc/build/libdrx_it128.so
run() { python bench.py --workload ml-1m --steps 100 --warmup 10 --no-cpu-baseline --no-hr --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k[:12]: round(x,3) for k,x in d['phases_ms'].items()})"; }
run it128_base
DRX_SIDE_STREAMS=3 run streams3
DRX_SIDE_STREAMS=3 DRX_PREP_AHEAD=4 run streams3_ahead4
DRX_SIDE_STREAMS=4 DRX_PREP_AHEAD=5 run streams4_ahead5
DRX_SIDE_PRIORITY=0 run prio0
DRX_SIDE_STREAMS=3 DRX_SIDE_PRIORITY=0 DRX_PREP_AHEAD=4 run streams3_prio0_ahead4
