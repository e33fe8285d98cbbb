"""Loaders for the MovieLens wire formats the reference's integrated datasets use
(DRecPy/Dataset/integrated_datasets.py:25-56) — from LOCAL files only (the reference downloads them; this engine is
built for offline machines).  Columns are always user, item, interaction, timestamp.

    ml-100k : u.data      tab-separated     ml-1m : ratings.dat    '::'-separated
    ml-10m  : ratings.dat '::'-separated    ml-20m / ml-latest: ratings.csv comma-separated with a header
"""
import os

import numpy as np

from .interaction_dataset import MemoryInteractionDataset

FORMATS = {
    'ml-100k': ('u.data', '\t', False),
    'ml-1m': ('ratings.dat', '::', False),
    'ml-10m': ('ratings.dat', '::', False),
    'ml-20m': ('ratings.csv', ',', True),
}


def read_ratings(path, delimiter, has_header=False):
    """Parses a ratings file into int64 user/item/timestamp and float64 interaction arrays (no pandas needed)."""
    users, items, vals, stamps = [], [], [], []
    with open(path, 'r', encoding='latin-1') as f:
        if has_header:
            next(f)
        for line in f:
            line = line.rstrip('\r\n')
            if not line:
                continue
            u, i, r, t = line.split(delimiter)[:4]
            users.append(int(u)); items.append(int(i)); vals.append(float(r)); stamps.append(int(float(t)))
    vals = np.asarray(vals, dtype=np.float64)
    if np.all(vals == np.round(vals)):
        vals = vals.astype(np.int64)                     # integer ratings stay integers, like pandas would infer
    return {'user': np.asarray(users, np.int64), 'item': np.asarray(items, np.int64), 'interaction': vals,
            'timestamp': np.asarray(stamps, np.int64)}


def load_movielens(name, folder, **kwds):
    """InteractionDataset of the full ratings file of `name` found under `folder`."""
    if name not in FORMATS:
        raise Exception(f'Unknown dataset "{name}" (supported: {sorted(FORMATS)}).')
    fname, delimiter, header = FORMATS[name]
    path = os.path.join(folder, fname)
    if not os.path.exists(path):
        raise FileNotFoundError(f'{path} not found: place the MovieLens file there (no download is attempted).')
    return MemoryInteractionDataset(df=read_ratings(path, delimiter, header), verbose=kwds.get('verbose', False))
