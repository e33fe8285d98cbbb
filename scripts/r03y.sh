#!/bin/bash
OUT=gpurun_out/r03y; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_caser.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
DRX_HOST_SANITIZER_LIB=$PWD/drecpy_amd/csrc/build/libdrx_stamps.so timeout -k 5 300 python scripts/stamps_caser.py > $OUT/stamps.json 2> $OUT/stamps.err
python - $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + '/stamps.json'))
print({k: (v['mean'] if isinstance(v, dict) and 'mean' in v else v) for k, v in d.items()})
PY
python - <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
from bench_configs import frame_of, caser_block
from drecpy_amd.Dataset import InteractionDataset
ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
print(caser_block(ds, torch.device('cuda:0')))
PY
