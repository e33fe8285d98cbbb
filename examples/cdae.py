"""CDAE, reference mode: the step DRecPy takes (all output units, dense Keras Adam), on one MI355X.
    python examples/cdae.py [--movielens /data/ml-100k]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from drecpy_amd import synth
from drecpy_amd.Dataset import load_movielens
from drecpy_amd.Evaluation import leave_k_out, ranking_evaluation
from drecpy_amd.Recommender import CDAE

ap = argparse.ArgumentParser()
ap.add_argument('--movielens', help='folder holding u.data (ml-100k)')
ap.add_argument('--epochs', type=int, default=500)
args = ap.parse_args()

ds = load_movielens('ml-100k', args.movielens) if args.movielens else synth.dataset('ml-100k', extra_per_user=12)
train, test = leave_k_out(ds, k=10, min_user_interactions=10, seed=10)

t0 = time.time()
model = CDAE(hidden_factors=50, corruption_level=0.2, loss='bce', seed=10)
model.fit(train, learning_rate=0.001, reg_rate=0.001, epochs=args.epochs, batch_size=64, neg_ratio=5)
print(f'fit: {time.time() - t0:.2f} s for {args.epochs} steps of 64')

print(ranking_evaluation(model, test, k=[1, 5, 10], novelty=True, n_test_users=100, n_pos_interactions=1,
                         n_neg_interactions=100, generate_negative_pairs=True, seed=10, verbose=False))
print('top-5 for user 1:', model.recommend(1, n=5))
