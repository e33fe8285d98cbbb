"""Reference-mode fit(): wall time against the number of steps (is the per-step cost flat from the first step?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from measure_models import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import CDAE
ds = InteractionDataset.read_df(frame_of('ml-100k'), verbose=False)
m = CDAE(hidden_factors=50, corruption_level=0.2, seed=10, verbose=False)
m.fit(ds, epochs=10, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
for n in (1, 10, 100, 300, 1000, 2000, 5000, 5000, 10000):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        m.fit(ds, epochs=n, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f'fit({n}) {best*1e3:.1f} ms')
