"""DMF.fit(device_sampler=True) at B = 256: steady ms per step for DMF and ModifiedDMF, in both orders (bench_configs._fit_steady)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                     # noqa: E402
import bench_configs as bc                                       # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import DMF                           # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))
from extending_recommender_dmf import ModifiedDMF               # noqa: E402

ds = InteractionDataset.read_df(bc.frame_of('ml-1m'), verbose=False)
order = sys.argv[1].split(',') if len(sys.argv) > 1 else ['DMF', 'ModifiedDMF', 'DMF']
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
classes = {'DMF': DMF, 'ModifiedDMF': ModifiedDMF}
for name in order:
    md = classes[name](user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=False, device='cuda:0')
    md.fit(ds, epochs=3, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5, device_sampler=True)
    e2d, steady, spread = bc._fit_steady(md, lambda n: md.fit(ds, epochs=n, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5, device_sampler=True), 400)
    print(name, B, 'steady ms/step', round(steady * 1e3, 4), spread, flush=True)
