"""The preparation of a MovieLens-shaped sampled batch ALONE (no training kernels beside it): draw + touch list, timed per call.
    python scripts/prep_alone.py [workload] [B]      (under rocprofv3 --kernel-trace --stats for the per-kernel times)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from drecpy_amd import synth
    from drecpy_amd.engine import CdaeEngine
    shape = sys.argv[1] if len(sys.argv) > 1 else 'ml-1m'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    U, N, md, mn, a = synth.SHAPES[shape]
    ip, idx = synth.synth_history(U, N, md, mn, a, seed=0, device='cuda', user_hi=U)
    eng = CdaeEngine(U, N, 128)
    eng.init_glorot_device(10)
    eng.set_history(ip, idx)
    eng.init_optimizer('adagrad', 0.05, 1e-3)
    prep = None
    for phase, n in (('warm', 5), ('timed', 50)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(n):
            uid, iid, y, ko = eng.sample_device(B, 5, 100 + s, n_items=N)
            bt, alive = eng.make_batch(uid, iid, y, keep_off=ko, q=0.2, mask_seed=7 + s)
            prep = eng.prepare_sparse(bt, prep)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f'{shape} B={B}: draw + preparation alone {dt * 1e3:.3f} ms per batch (host-synchronous draw: includes one host round trip)')


if __name__ == '__main__':
    main()
