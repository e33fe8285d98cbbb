"""CDAE in the sampled sparse-Adagrad mode through the ordinary fit() call: triples are drawn on the GPU two batches ahead,
the touch list of a batch is sorted one batch ahead, only touched rows are updated (engine.SampledPipeline).
    python examples/cdae_sampled_scale.py [--users 1000000]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from drecpy_amd import synth
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import CDAE

ap = argparse.ArgumentParser()
ap.add_argument('--users', type=int, default=1_000_000)
ap.add_argument('--epochs', type=int, default=400)
ap.add_argument('--batch', type=int, default=65536)
args = ap.parse_args()

_, n_items, mean_deg, min_deg, alpha = synth.SHAPES['synth-10m']
indptr, indices = synth.synth_history(args.users, n_items, mean_deg, min_deg, alpha, seed=0, device='cuda', user_hi=args.users)
indptr, indices = indptr.cpu().numpy(), indices.cpu().numpy()
user = np.repeat(np.arange(args.users, dtype=np.int64), np.diff(indptr))
order = np.random.RandomState(0).permutation(len(user))
ds = InteractionDataset.from_arrays(user[order], indices.astype(np.int64)[order], np.ones(len(user)))
print(f'{len(user)} interactions, {args.users} users, {n_items} items')

model = CDAE(hidden_factors=128, corruption_level=0.2, mode="sampled", device_sampler=True, seed=10, verbose=False)  # verbose fetches the loss every step
t0 = time.time()
model.fit(ds, learning_rate=0.05, reg_rate=0.001, epochs=args.epochs, batch_size=args.batch, neg_ratio=5)
torch.cuda.synchronize()
dt = time.time() - t0
print(f'fit (id map, CSR, sampler, tables + {args.epochs} steps of {args.batch}): {dt:.2f} s')
print('recommendations for user 0:', model.recommend(0, n=5))
