"""End-to-end check of the shared-form sampled step through the PUBLIC fit(): CDAE(mode='sampled', device_sampler=True) on the
ml-1m-shaped synthetic set (leave-10-out, protocol of examples/cdae.py:15-17), K = 128, 300 steps of 65 536 triples, once with
DRX_BATCH_SHARE_USERS (the default where the history's transpose exists) and once without (eng.share_users = False): HR@10 / NDCG@10
after training must agree to sampling noise — the two forms are the same model under another association of the sums.
    python scripts/hr10_ml1m.py > profiles/r04_hr10_ml1m_shared.json"""
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
from drecpy_amd.Evaluation import leave_k_out, ranking_evaluation       # noqa: E402
from drecpy_amd.Recommender import CDAE                                 # noqa: E402
from drecpy_amd.engine import CdaeEngine                                # noqa: E402


def main():
    U, N, md, mn, a = synth.SHAPES['ml-1m']
    ip, idx = synth.synth_history(U, N, md + 12, mn, a, seed=0)
    ip, idx = ip.numpy(), idx.numpy()
    rng = np.random.RandomState(0)
    user = np.repeat(np.arange(U), np.diff(ip)) + 1
    perm = rng.permutation(len(user))
    ds = InteractionDataset.read_df({'user': user[perm], 'item': (idx.astype(np.int64) + 1)[perm],
                                     'interaction': rng.randint(1, 6, size=len(user))[perm]}, verbose=False)
    tr, te = leave_k_out(ds, k=10, min_user_interactions=10, seed=10, verbose=False)
    proto = dict(k=[1, 5, 10], novelty=True, n_test_users=300, n_pos_interactions=1, n_neg_interactions=100, generate_negative_pairs=True,
                 seed=10, verbose=False)
    out = {'dataset': {'users': U, 'items': N, 'train_rows': len(tr), 'test_rows': len(te)}, 'steps': 300, 'batch': 65536}
    for name, share in (('shared_form', True), ('plain_lists', False)):
        CdaeEngine.share_users = share
        w = CDAE(hidden_factors=128, corruption_level=0.2, loss='bce', mode='sampled', device_sampler=True, seed=10, verbose=False)
        w.fit(tr, epochs=20, batch_size=65536, learning_rate=0.05, reg_rate=1e-3, neg_ratio=5)      # (loads kernels, builds the transpose once)
        m = CDAE(hidden_factors=128, corruption_level=0.2, loss='bce', mode='sampled', device_sampler=True, seed=10, verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.fit(tr, epochs=300, batch_size=65536, learning_rate=0.05, reg_rate=1e-3, neg_ratio=5)
        torch.cuda.synchronize()
        fit_s = time.perf_counter() - t0
        res = ranking_evaluation(m, te, **proto)
        out[name] = {'fit_seconds_incl_setup': round(fit_s, 3), 'shared_form_flag': bool(m._engine._batch_flags(0) & 1), 'after_training': res}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
