"""cProfile of DMF.fit(device_sampler=True) at B = 256 (ml-1m shape): where the host's 0.11 ms per step go."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                     # noqa: E402
import bench_configs as bc                                       # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import DMF                           # noqa: E402

ds = InteractionDataset.read_df(bc.frame_of('ml-1m'), verbose=False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev_sampler = (sys.argv[2] if len(sys.argv) > 2 else 'device') == 'device'
md = DMF(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=False, device='cuda:0')
md.fit(ds, epochs=50, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5, device_sampler=dev_sampler)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
md.fit(ds, epochs=2000, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5, device_sampler=dev_sampler)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
