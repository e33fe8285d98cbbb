"""Caser (sequence model: horizontal/vertical convolutions over the last L items), trained from ListSampler windows and
scored with recommendation_evaluation, as DRecPy's examples/caser.py does.
    python examples/caser.py [--movielens /data/ml-1m]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from drecpy_amd import synth
from drecpy_amd.Dataset import load_movielens
from drecpy_amd.Evaluation import leave_k_out, recommendation_evaluation
from drecpy_amd.Recommender import Caser

ap = argparse.ArgumentParser()
ap.add_argument('--movielens', help='folder holding ratings.dat (ml-1m)')
ap.add_argument('--epochs', type=int, default=100)
args = ap.parse_args()

ds = load_movielens('ml-1m', args.movielens) if args.movielens else synth.dataset('ml-100k', extra_per_user=12)
train, test = leave_k_out(ds, k=10, min_user_interactions=20, seed=10)
model = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, sort_column='timestamp', seed=10)
t0 = time.time()
model.fit(train, epochs=args.epochs, batch_size=512, learning_rate=0.001, reg_rate=1e-6, neg_ratio=3)
print(f'fit: {time.time() - t0:.2f} s')
print(recommendation_evaluation(model, test, n_test_users=200, k=[1, 5, 10], novelty=True, seed=10, verbose=False))
