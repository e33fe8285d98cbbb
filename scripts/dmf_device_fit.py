"""cfg 3 of the bench line alone: DMF / ModifiedDMF step and fit() rates at the ml-1m shape, host stream vs device sampler.
    python scripts/dmf_device_fit.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_configs as bc                                    # noqa: E402
from drecpy_amd.Dataset import InteractionDataset             # noqa: E402

ds = InteractionDataset.read_df(bc.frame_of('ml-1m'), verbose=False)
out = bc.dmf_block(ds, torch.device('cuda', 0))
print(json.dumps({k: v for k, v in out.items() if k.startswith(('DMF', 'Modified'))}, indent=1))
