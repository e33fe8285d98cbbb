"""cProfile of Caser.fit(device_sampler=True), 300 steps at B = 4096 (where the host time of the fit loop goes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_configs import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import Caser
B = 4096
ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
m = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, seed=10, verbose=False)
m.fit(ds, epochs=3, batch_size=B, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3, device_sampler=True)
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
cProfile.run('m.fit(ds, epochs=300, batch_size=B, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3, device_sampler=True); torch.cuda.synchronize()', '/tmp/cf.prof')
print('ms per step under cProfile', (time.perf_counter() - t0) / 300 * 1e3)
pstats.Stats('/tmp/cf.prof').sort_stats('cumtime').print_stats(30)
