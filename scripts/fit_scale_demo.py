"""Public-API run at scale on one MI355X: a 1M-user x 1M-item x ~20M-interaction synthetic set goes through
InteractionDataset.from_arrays -> CDAE(mode='sampled', device_sampler=True).fit(), i.e. the same calls a DRecPy user makes
(examples/cdae.py), and the fit() throughput is compared with bench.py's number for the same shape
(`python bench.py --users 1000000`).  Prints one JSON object.
    python scripts/fit_scale_demo.py > profiles/r01_fit_scale.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from drecpy_amd import synth                                            # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                       # noqa: E402
from drecpy_amd.Recommender import CDAE                                 # noqa: E402


def main():
    U = int(os.environ.get('DEMO_USERS', 1_000_000))
    _, N, md, mn, a = synth.SHAPES['synth-10m']
    ip, idx = synth.synth_history(U, N, md, mn, a, seed=0, device='cuda', user_hi=U)
    ip, idx = ip.cpu().numpy(), idx.cpu().numpy()
    rng = np.random.RandomState(0)
    user = np.repeat(np.arange(U, dtype=np.int64), np.diff(ip)) + 1000          # raw ids differ from internal ids
    item = idx.astype(np.int64) + 500
    perm = rng.permutation(len(user))
    out = {'rows': int(len(user)), 'users': U, 'items': N}
    t0 = time.perf_counter()
    ds = InteractionDataset.from_arrays(user[perm], item[perm], np.ones(len(user), dtype=np.float64))
    out['dataset_build_s'] = round(time.perf_counter() - t0, 2)
    B, epochs = 65536, int(os.environ.get('DEMO_EPOCHS', 400))
    m = CDAE(hidden_factors=128, corruption_level=0.2, loss='bce', mode='sampled', device_sampler=True, seed=10, verbose=False)
    t0 = time.perf_counter()
    if os.environ.get('DEMO_PROFILE'):
        import cProfile
        import pstats
        cProfile.runctx("m.fit(ds, epochs=2, batch_size=B, learning_rate=0.05, reg_rate=1e-3, neg_ratio=5)", globals(), locals(), '/tmp/fit.prof')
        pstats.Stats('/tmp/fit.prof', stream=sys.stderr).sort_stats('cumulative').print_stats(30)
    else:
        m.fit(ds, epochs=2, batch_size=B, learning_rate=0.05, reg_rate=1e-3, neg_ratio=5)   # id map, CSR, sampler, tables
    torch.cuda.synchronize()
    out['first_fit_incl_setup_s'] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    m.fit(ds, epochs=epochs, batch_size=B, learning_rate=0.05, reg_rate=1e-3, neg_ratio=5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out['fit_epochs'] = epochs
    out['fit_s_incl_setup'] = round(dt, 3)
    out['fit_samples_per_s_incl_setup'] = round(B * epochs / dt, 1)
    # steady state of the same fit loop: time the epochs alone via the hooks fit() itself calls
    m._pre_fit(0.05, 5, 1e-3)
    for e in range(20):
        m._do_batch(m._sample_batch(B), step=e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in range(20, 20 + epochs):
        m._do_batch(m._sample_batch(B), step=e)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out['hook_loop_samples_per_s'] = round(B * epochs / dt, 1)
    out['hook_loop_ms_per_step'] = round(dt / epochs * 1e3, 4)
    u0 = int(user[perm][0])
    out['rank_example'] = [[float(s), int(i)] for s, i in m.recommend(u0, n=3, novelty=True)]
    print(json.dumps(out))


if __name__ == '__main__':
    main()
