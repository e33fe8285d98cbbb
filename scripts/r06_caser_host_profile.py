"""cProfile of Caser.fit(device_sampler=True) at B = 4096 (ml-1m shape)."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                     # noqa: E402
import bench_configs as bc                                       # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import Caser                         # noqa: E402

ds = InteractionDataset.read_df(bc.frame_of('ml-1m'), verbose=False)
m = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, seed=10, verbose=False)
m.fit(ds, epochs=50, batch_size=4096, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3, device_sampler=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
m.fit(ds, epochs=2000, batch_size=4096, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3, device_sampler=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
