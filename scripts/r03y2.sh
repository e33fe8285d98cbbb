#!/bin/bash
timeout -k 5 600 python -m pytest tests/test_gpu_caser.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from bench_configs import frame_of, caser_block
from drecpy_amd.Dataset import InteractionDataset
ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
print(caser_block(ds, torch.device('cuda:0')))
PY
