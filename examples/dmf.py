"""DMF (two dense towers over rating rows / columns, cosine score) — the model DRecPy builds in
examples/extending_recommender_dmf.py; rank() scores a user against all candidates with the bf16 MFMA kernel.
    python examples/dmf.py [--movielens /data/ml-1m]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from drecpy_amd import synth
from drecpy_amd.Dataset import load_movielens
from drecpy_amd.Evaluation import leave_k_out, ranking_evaluation
from drecpy_amd.Recommender import DMF

ap = argparse.ArgumentParser()
ap.add_argument('--movielens', help='folder holding ratings.dat (ml-1m)')
ap.add_argument('--epochs', type=int, default=200)
args = ap.parse_args()

ds = load_movielens('ml-1m', args.movielens) if args.movielens else synth.dataset('ml-100k', extra_per_user=12)
train, test = leave_k_out(ds, k=10, min_user_interactions=10, seed=10)
model = DMF(user_factors=[64, 32], item_factors=[64, 32], seed=10)
t0 = time.time()
model.fit(train, epochs=args.epochs, batch_size=256, learning_rate=0.001, reg_rate=0.0001, neg_ratio=5)
print(f'fit: {time.time() - t0:.2f} s')
print(ranking_evaluation(model, test, k=[1, 5, 10], novelty=True, n_test_users=100, n_pos_interactions=1,
                         n_neg_interactions=100, generate_negative_pairs=True, seed=10, verbose=False))
