"""cProfile of CDAE.fit() in the sampled mode on a 1 M-user x 1 M-item, 20 M-interaction set: what the set-up before the first step costs."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drecpy_amd import synth
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import CDAE
users = 1_000_000
_, n_items, mean_deg, min_deg, alpha = synth.SHAPES['synth-10m']
indptr, indices = synth.synth_history(users, n_items, mean_deg, min_deg, alpha, seed=0, device='cuda', user_hi=users)
indptr, indices = indptr.cpu().numpy(), indices.cpu().numpy()
user = np.repeat(np.arange(users, dtype=np.int64), np.diff(indptr))
order = np.random.RandomState(0).permutation(len(user))
t0 = time.time()
ds = InteractionDataset.from_arrays(user[order], indices.astype(np.int64)[order], np.ones(len(user)))
print('from_arrays', round(time.time() - t0, 2), 's')
model = CDAE(hidden_factors=128, corruption_level=0.2, mode="sampled", device_sampler=True, seed=10, verbose=False)
model.fit(ds, learning_rate=0.05, reg_rate=0.001, epochs=2, batch_size=65536, neg_ratio=5)      # (code objects, allocator)
pr = cProfile.Profile(); pr.enable()
t0 = time.time()
model.fit(ds, learning_rate=0.05, reg_rate=0.001, epochs=10, batch_size=65536, neg_ratio=5)
torch.cuda.synchronize()
print('fit(10 epochs)', round(time.time() - t0, 2), 's')
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
