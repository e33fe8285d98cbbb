"""Pins the FLOATING-POINT hot path to the real reference — the day TensorFlow exists in the dev container.

TEST INFRASTRUCTURE (see oracle/__init__.py).  TensorFlow (`requirements.txt:6`, tensorflow>=2.0<3) is absent from this image and
cannot be installed (no network): every arithmetic parity claim of this repo is against oracle/*_oracle.py, "parity unpinned".
This script closes that gap without further work: run in a container that has /root/reference AND an importable tensorflow, it
executes the REAL DRecPy.Recommender.CDAE / DMF / Caser fit() for k in {1, 10} one-batch epochs from INJECTED initial weights
(TF's initialisers are irreproducible; everything else — PointSampler / ListSampler streams, the `random.Random(seed)` corruption
stream of cdae.py:63 — is deterministic given `seed`) and records inputs + resulting weights + predictions as

    tests/golden/tf_cdae.npz, tf_dmf.npz, tf_caser.npz

which tests/test_tf_golden.py (CPU: the oracle; GPU: the HIP path) compares against at 1e-5 relative — skipped while the files do not
exist.  Caser: dropout_rate = 0 (TF's dropout stream cannot be injected).  Nothing here is imported by the product.

    python oracle/gen_golden_tf.py            # prints what it wrote, or why it wrote nothing
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
REFERENCE_ROOT = '/root/reference'
STEPS = (1, 10)


def frame(seed=5, n_users=48, n_items=70, n_rows=1400, with_ts=False):
    r = np.random.RandomState(seed)
    u = r.randint(0, n_users, size=n_rows)
    pop = 1.0 / np.arange(1, n_items + 1)
    i = r.choice(n_items, size=n_rows, p=pop / pop.sum())
    _, first = np.unique(u.astype(np.int64) * n_items + i, return_index=True)
    first.sort()
    u, i = u[first], i[first]
    f = {'user': (u + 1).astype(np.int64), 'item': (i + 1).astype(np.int64), 'interaction': r.randint(1, 6, size=len(u)).astype(np.int64)}
    if with_ts:
        f['timestamp'] = r.randint(0, 10 ** 6, size=len(u)).astype(np.int64)
    return f


def glorot(rng, shape):
    fi, fo = (shape[0], shape[0]) if len(shape) == 1 else (shape[-2] * int(np.prod(shape[:-2])), shape[-1] * int(np.prod(shape[:-2])))
    lim = np.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def main():
    # the script's own pieces first (they need no TensorFlow): a broken helper must not hide behind the missing module
    f0 = frame()
    assert len(f0['user']) == len(f0['item']) == len(f0['interaction']) > 0 and glorot(np.random.default_rng(0), (3, 4)).shape == (3, 4)
    try:
        import tensorflow as tf
    except ModuleNotFoundError as e:
        if e.name != 'tensorflow':                          # (a TensorFlow that is installed but broken is an ERROR, not "absent")
            raise
        print(f'tensorflow is not importable here ({e!r}): nothing generated, FP parity stays "unpinned"')
        return 0
    if not os.path.isdir(REFERENCE_ROOT):
        print('the reference tree is not present: nothing generated')
        return 0
    warnings.filterwarnings('ignore')
    sys.dont_write_bytecode = True
    os.environ.setdefault('MPLBACKEND', 'Agg')
    if not hasattr(np, 'float'):
        np.float = float                                    # mem_dataset.py:150 uses the removed alias
    sys.path.insert(0, REFERENCE_ROOT)
    import pandas as pd
    from DRecPy.Dataset import InteractionDataset
    from DRecPy.Recommender import CDAE, DMF, Caser
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(2024)

    def dataset(f):
        return InteractionDataset.read_df(pd.DataFrame(f), verbose=False)

    # ---- CDAE (cdae.py:25-82; recommender_abc.py:186-205,328-334) --------------------------------------------------------------
    f = frame()
    n_users, n_items = len(np.unique(f['user'])), len(np.unique(f['item']))
    K, B = 12, 16
    init = {'W': glorot(rng, (n_items, K)), 'W_': glorot(rng, (K, n_items)), 'V': glorot(rng, (n_users, K)), 'b': glorot(rng, (K,)),
            'b_': glorot(rng, (n_items,))}

    class InjectedCDAE(CDAE):
        def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
            super()._pre_fit(learning_rate, neg_ratio, reg_rate, **kwds)
            for name, v in init.items():
                getattr(self, name).assign(v)

    rec = {'frame_' + k: v for k, v in f.items()}
    rec.update({'init_' + k: v for k, v in init.items()})
    rec.update(K=K, B=B, seed=10, q=0.2, lr=1e-3, reg=1e-3, neg_ratio=5, steps=np.array(STEPS))
    for loss in ('bce', 'mse'):
        for k in STEPS:
            m = InjectedCDAE(hidden_factors=K, corruption_level=0.2, loss=loss, seed=10, verbose=False)
            m.fit(dataset(f), epochs=k, batch_size=B, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
            for name in init:
                rec[f'{loss}_k{k}_{name}'] = getattr(m, name).numpy()
            rec[f'{loss}_k{k}_pred'] = np.stack([m._predict(u) for u in range(m.n_users)])
    np.savez_compressed(os.path.join(OUT, 'tf_cdae.npz'), **rec)
    print('wrote tf_cdae.npz')

    # ---- DMF (dmf.py:46-99) ---------------------------------------------------------------------------------------------------
    uf, itf = [16, 8], [16, 8]
    dinit = {}
    dims_u, dims_i = [n_items] + uf, [n_users] + itf
    for t, dims in (('u', dims_u), ('i', dims_i)):
        for li in range(len(dims) - 1):
            dinit[f'{t}_k{li}'] = glorot(rng, (dims[li], dims[li + 1]))
            dinit[f'{t}_b{li}'] = (rng.standard_normal(dims[li + 1]) * 0.01).astype(np.float32)

    class InjectedDMF(DMF):
        def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
            super()._pre_fit(learning_rate, neg_ratio, reg_rate, **kwds)
            for t, nn in (('u', self.user_nn), ('i', self.item_nn)):
                for li, layer in enumerate(nn.layers):
                    layer.set_weights([dinit[f'{t}_k{li}'], dinit[f'{t}_b{li}']])

    rec = {'frame_' + k: v for k, v in f.items()}
    rec.update({'init_' + k: v for k, v in dinit.items()})
    rec.update(user_factors=np.array(uf), item_factors=np.array(itf), B=B, seed=10, lr=1e-3, reg=1e-4, neg_ratio=5, steps=np.array(STEPS))
    pairs = [(u, i) for u in range(0, n_users, 5) for i in range(0, n_items, 7)]
    rec['probe_pairs'] = np.array(pairs)
    for k in STEPS:
        m = InjectedDMF(user_factors=uf, item_factors=itf, seed=10, verbose=False)
        m.fit(dataset(f), epochs=k, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5)
        for t, nn in (('u', m.user_nn), ('i', m.item_nn)):
            for li, layer in enumerate(nn.layers):
                kk, bb = layer.get_weights()
                rec[f'k{k}_{t}_k{li}'], rec[f'k{k}_{t}_b{li}'] = kk, bb
        rec[f'k{k}_pred'] = np.array([float(m._predict(u, i)) for u, i in pairs], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'tf_dmf.npz'), **rec)
    print('wrote tf_dmf.npz')

    # ---- Caser (caser.py:45-120), dropout off ---------------------------------------------------------------------------------------
    f = frame(seed=6, n_users=40, n_items=60, n_rows=1600, with_ts=True)
    n_users, n_items = len(np.unique(f['user'])), len(np.unique(f['item']))
    L, T, d, n_v, n_h = 4, 2, 8, 2, 4
    cinit = {'user_emb': rng.uniform(-0.05, 0.05, (n_users, d)).astype(np.float32), 'item_emb': rng.uniform(-0.05, 0.05, (n_items, d)).astype(np.float32),
             'conv_v_k': glorot(rng, (L, d, n_v)), 'conv_v_b': np.zeros(n_v, np.float32),
             'dense_0_k': glorot(rng, (d * n_v + n_h * L, d)), 'dense_0_b': np.zeros(d, np.float32),
             'dense_1_W': rng.uniform(-0.05, 0.05, (n_items, 2 * d)).astype(np.float32),
             'dense_1_b': rng.uniform(-0.05, 0.05, (n_items, 1)).astype(np.float32)}
    for i in range(L):
        cinit[f'conv_h{i}_k'], cinit[f'conv_h{i}_b'] = glorot(rng, (i + 1, d, n_h)), np.zeros(n_h, np.float32)

    class InjectedCaser(Caser):
        def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
            super()._pre_fit(learning_rate, neg_ratio, reg_rate, **kwds)
            # Keras layers build on first call: one dry prediction creates the variables, then they are overwritten
            self._predict_batch_aux([0], [[0] * self.L], [[0] * (self.T * (1 + neg_ratio))], training=False)
            self.user_embeddings.set_weights([cinit['user_emb']]); self.item_embeddings.set_weights([cinit['item_emb']])
            self.conv_v.set_weights([cinit['conv_v_k'], cinit['conv_v_b']])
            for i, c in enumerate(self.convs_h):
                c.set_weights([cinit[f'conv_h{i}_k'], cinit[f'conv_h{i}_b']])
            self.dense_0.set_weights([cinit['dense_0_k'], cinit['dense_0_b']])
            self.dense_1_W.set_weights([cinit['dense_1_W']]); self.dense_1_b.set_weights([cinit['dense_1_b']])

    rec = {'frame_' + k: v for k, v in f.items()}
    rec.update({'init_' + k: v for k, v in cinit.items()})
    rec.update(L=L, T=T, d=d, n_v=n_v, n_h=n_h, B=B, seed=10, lr=5e-3, reg=1e-6, neg_ratio=3, steps=np.array(STEPS))
    for k in STEPS:
        m = InjectedCaser(L=L, T=T, d=d, n_v=n_v, n_h=n_h, dropout_rate=0.0, seed=10, verbose=False)
        m.fit(dataset(f), epochs=k, batch_size=B, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3)
        rec[f'k{k}_user_emb'], rec[f'k{k}_item_emb'] = m.user_embeddings.get_weights()[0], m.item_embeddings.get_weights()[0]
        rec[f'k{k}_conv_v_k'], rec[f'k{k}_conv_v_b'] = m.conv_v.get_weights()
        for i, c in enumerate(m.convs_h):
            rec[f'k{k}_conv_h{i}_k'], rec[f'k{k}_conv_h{i}_b'] = c.get_weights()
        rec[f'k{k}_dense_0_k'], rec[f'k{k}_dense_0_b'] = m.dense_0.get_weights()
        rec[f'k{k}_dense_1_W'], rec[f'k{k}_dense_1_b'] = m.dense_1_W.get_weights()[0], m.dense_1_b.get_weights()[0]
    np.savez_compressed(os.path.join(OUT, 'tf_caser.npz'), **rec)
    print('wrote tf_caser.npz')
    print('tensorflow', tf.__version__)
    return 0


if __name__ == '__main__':
    sys.exit(main())
