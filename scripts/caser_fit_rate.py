"""Caser.fit() rates at the ml-1m shape, B = 4096 (bench_configs.caser_block: the engine step, fit() on the reference-exact ListSampler
stream and with the device sampler, fenced windows).  python scripts/caser_fit_rate.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                     # noqa: E402
import bench_configs as bc                                       # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402

ds = InteractionDataset.read_df(bc.frame_of('ml-1m'), verbose=False)
print(json.dumps(bc.caser_block(ds, torch.device('cuda:0'))))
