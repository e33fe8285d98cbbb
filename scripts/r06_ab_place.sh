#!/bin/bash
# r06 item 4: the streamed reduction with the hot rows' blocks placed by XCD (default build) against list order (variant `noplace`).
set -u
TAG=${1:-r06f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
COMMON="--steps 200 --warmup 20 --windows 3 --no-cpu-baseline --no-hr --no-configs"
for i in 1 2; do
  python bench.py $COMMON > $OUT/placed_$i.json 2> $OUT/placed_$i.err
  DRX_HOST_SANITIZER_LIB=drecpy_amd/csrc/build/libdrx_noplace.so python bench.py $COMMON > $OUT/noplace_$i.json 2> $OUT/noplace_$i.err
  [ -f drecpy_amd/csrc/build/libdrx_lookuponly.so ] && DRX_HOST_SANITIZER_LIB=drecpy_amd/csrc/build/libdrx_lookuponly.so python bench.py $COMMON > $OUT/lookuponly_$i.json 2> $OUT/lookuponly_$i.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), round(d['value'] / 1e6, 2), 'M/s', round(d['ms_per_step'], 4), 'ms', [round(v, 4) for v in (d.get('phases_ms') or {}).values()], 'frac', round(d['roofline']['frac'], 4))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-800:])
PY
