"""Runs reference-mode CDAE steps at an ml-1m-shaped problem (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drecpy_amd import synth                      # noqa: E402
from drecpy_amd.engine import CdaeEngine          # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else 'ml-1m'
K = 128 if shape == 'ml-1m' else 50
U, N, md, mn, a = synth.SHAPES[shape]
ip, idx = synth.synth_history(U, N, md, mn, a, seed=0, device='cuda')
eng = CdaeEngine(U, N, K)
eng.init_glorot_device(10)
eng.set_history(ip, idx)
eng.init_optimizer('adam', 1e-3, 1e-3)
rng = np.random.default_rng(0)
bts = [eng.make_batch(rng.integers(0, U, size=64), q=0.2, mask_seed=s) for s in range(8)]
for s in range(300):
    eng.step_dense(s, bts[s % 8][0])
torch.cuda.synchronize()
print('done')
