"""Long reference-mode fits: is the per-step time of the in-library loop flat in the number of steps?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from measure_models import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import CDAE
shape, K = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ('ml-100k', 50)
ds = InteractionDataset.read_df(frame_of(shape), verbose=False)
m = CDAE(hidden_factors=K, corruption_level=0.2, seed=10, verbose=False)
for n in (2000, 5000, 20000, 50000, 100000, 5000):
    t0 = time.perf_counter()
    m.fit(ds, epochs=n, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(shape, n, 'steps', round(dt, 3), 's', round(dt / n * 1e6, 1), 'us/step', 'finite', bool(np.isfinite(m._engine.W.cpu().numpy()).all()))
