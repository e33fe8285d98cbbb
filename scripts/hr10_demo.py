"""End-to-end sanity run on one MI355X: train CDAE (reference mode, README configuration: K=50, q=0.2, BCE, 100 one-batch
epochs of 64, lr 1e-3, reg 1e-3, neg_ratio 5, seed 10) on the ml-100k-shaped synthetic set with a leave-10-out split and
report HR@k / NDCG@k under the protocol of examples/cdae.py:15-17 before and after training, then the same data with the
sampled sparse-Adagrad mode.  No MovieLens files exist offline, so the numbers are not comparable with README.md:140-141
(0.5536 on the real ml-100k); they show that the loop learns (untrained HR@10 ~ 10/101).
    python scripts/hr10_demo.py > profiles/r01_hr10.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from drecpy_amd import synth                                            # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                       # noqa: E402
from drecpy_amd.Evaluation import leave_k_out, ranking_evaluation       # noqa: E402
from drecpy_amd.Recommender import CDAE                                 # noqa: E402


def main():
    U, N, md, mn, a = synth.SHAPES['ml-100k']
    ip, idx = synth.synth_history(U, N, md + 12, mn, a, seed=0)          # +10 per user for the hold-out
    ip, idx = ip.numpy(), idx.numpy()
    rng = np.random.RandomState(0)
    user = np.repeat(np.arange(U), np.diff(ip)) + 1
    item = idx.astype(np.int64) + 1
    perm = rng.permutation(len(user))
    ds = InteractionDataset.read_df({'user': user[perm], 'item': item[perm], 'interaction': rng.randint(1, 6, size=len(user))[perm]},
                                    verbose=False)
    ds_train, ds_test = leave_k_out(ds, k=10, min_user_interactions=10, seed=10, verbose=False)
    proto = dict(k=[1, 5, 10], novelty=True, n_test_users=100, n_pos_interactions=1, n_neg_interactions=100,
                 generate_negative_pairs=True, seed=10, verbose=False)
    out = {'dataset': {'users': U, 'items': N, 'train_rows': len(ds_train), 'test_rows': len(ds_test)}, 'protocol': 'examples/cdae.py:15-17'}
    for name, kw, fitkw in (('reference_mode_100_steps_of_64', dict(mode='reference'), dict(epochs=100, batch_size=64, learning_rate=1e-3)),
                            ('reference_mode_2000_steps_of_64', dict(mode='reference'), dict(epochs=2000, batch_size=64, learning_rate=1e-3)),
                            ('sampled_mode_400_steps_of_4096', dict(mode='sampled'), dict(epochs=400, batch_size=4096, learning_rate=0.05))):
        m = CDAE(hidden_factors=50, corruption_level=0.2, loss='bce', seed=10, verbose=False, **kw)
        m.fit(ds_train, epochs=1, reg_rate=1e-3, neg_ratio=5, **{k: v for k, v in fitkw.items() if k != 'epochs'})
        before = ranking_evaluation(m, ds_test, **proto)
        t0 = time.perf_counter()
        m2 = CDAE(hidden_factors=50, corruption_level=0.2, loss='bce', seed=10, verbose=False, **kw)
        m2.fit(ds_train, reg_rate=1e-3, neg_ratio=5, **fitkw)
        fit_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        after = ranking_evaluation(m2, ds_test, **proto)
        ev_s = time.perf_counter() - t0
        out[name] = {'after_1_step': before, 'after_training': after, 'fit_seconds': round(fit_s, 2),
                     'samples_per_s_incl_host': round(fitkw['epochs'] * fitkw['batch_size'] / fit_s, 1),
                     'ranking_evaluation_seconds_100_users': round(ev_s, 2)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
