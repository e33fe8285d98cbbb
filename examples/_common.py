"""Shared plumbing of the example scripts: repo import path, dataset choice (a local MovieLens folder or the synthetic
MovieLens-shaped stand-in), train/test split, and a one-line report."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from drecpy_amd import synth                                    # noqa: E402
from drecpy_amd.Dataset import load_movielens                   # noqa: E402
from drecpy_amd.Evaluation import leave_k_out                   # noqa: E402


def arguments(default_epochs, dataset_name):
    ap = argparse.ArgumentParser()
    ap.add_argument('--movielens', metavar='FOLDER', help=f'folder holding the {dataset_name} ratings file (default: synthetic stand-in)')
    ap.add_argument('--epochs', type=int, default=default_epochs)
    ap.add_argument('--quiet', action='store_true', help='no progress bar (the bar fetches the loss from the GPU every step)')
    return ap.parse_args()


def split(args, dataset_name, hold_out=10, min_user_interactions=10):
    full = load_movielens(dataset_name, args.movielens) if args.movielens else synth.dataset('ml-100k', extra_per_user=hold_out + 2)
    return leave_k_out(full, k=hold_out, min_user_interactions=min_user_interactions, seed=10)


class stopwatch:
    def __init__(self, what):
        self.what = what

    def __enter__(self):
        self.t0 = time.time()
        return self

    def __exit__(self, *exc):
        print(f'{self.what}: {time.time() - self.t0:.2f} s')
        return False
