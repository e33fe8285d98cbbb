"""Long fits: does anything grow?  (device memory, host RSS) — DMF and Caser with the device samplers, CDAE sampled mode; 20 000 steps each."""
import os
import resource
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                     # noqa: E402
import bench_configs as bc                                       # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import CDAE, DMF, Caser              # noqa: E402

ds = InteractionDataset.read_df(bc.frame_of('ml-1m'), verbose=False)


def report(tag, t0, n):
    torch.cuda.synchronize()
    print(f'{tag}: {n} steps in {time.time() - t0:.2f} s; device {torch.cuda.memory_allocated() / 2**20:.1f} MiB allocated, '
          f'{torch.cuda.memory_reserved() / 2**20:.1f} reserved; host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MiB', flush=True)


for rounds in range(2):
    m = DMF(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=False)
    t0 = time.time(); m.fit(ds, epochs=20000, batch_size=256, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5, device_sampler=True); report('DMF device B=256', t0, 20000)
    t0 = time.time(); m.fit(ds, epochs=5000, batch_size=4096, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5); report('DMF host B=4096', t0, 5000)
    c = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, sort_column='timestamp', seed=10, verbose=False)
    t0 = time.time(); c.fit(ds, epochs=20000, batch_size=4096, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3, device_sampler=True); report('Caser device B=4096', t0, 20000)
    a = CDAE(hidden_factors=128, mode='sampled', device_sampler=True, seed=10, verbose=False)
    t0 = time.time(); a.fit(ds, epochs=20000, batch_size=65536, learning_rate=0.05, reg_rate=1e-3, neg_ratio=5); report('CDAE sampled B=65536', t0, 20000)
    del m, c, a
