"""Caser.fit(device_sampler=True) at the ml-1m shape, B = 4096 (for rocprofv3 --kernel-trace --stats): 3 warm-up epochs, then 10 timed."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch                                                     # noqa: E402
from measure_models import frame_of                              # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import Caser                         # noqa: E402

ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
m = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, seed=10, verbose=False)
m.fit(ds, epochs=3, batch_size=4096, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3, device_sampler=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
m.fit(ds, epochs=200, batch_size=4096, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3, device_sampler=True)
torch.cuda.synchronize()
print('ms/step incl. setup', (time.perf_counter() - t0) / 200 * 1e3)
