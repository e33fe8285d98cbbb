"""Shared helpers for tests: golden-frame loading and oracle-side dataset preparation."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_frames():
    z = np.load(os.path.join(GOLDEN, 'frames.npz'), allow_pickle=False)
    frames = {}
    for key in z.files:
        k, c = key.split('__')
        frames.setdefault(k, {})[c] = z[key]
    return frames
