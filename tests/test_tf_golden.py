"""Arithmetic parity against the REAL reference (TensorFlow) — dormant until tests/golden/tf_*.npz exist.

oracle/gen_golden_tf.py writes those files in a dev container that has /root/reference and an importable tensorflow (this image has
no TensorFlow: the FP oracle is "parity unpinned").  When they exist:
  * CPU (`-m "not gpu"`): oracle/cdae_oracle.py, fed the reference-exact PointSampler / corruption streams, must reproduce the weights
    and predictions TensorFlow produced after 1 and 10 one-batch epochs — that PINS the oracle;
  * GPU (`-m gpu`): drecpy_amd.Recommender.CDAE.fit() from the same injected weights must match them to 1e-5 relative.
DMF / Caser fixtures are checked the same way through the public fit() on the GPU."""
import os
import random

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-5          # north_star: predictions within 1e-5 relative of the reference TF path


def test_generator_script_runs_and_reports_only_a_missing_tensorflow():
    """oracle/gen_golden_tf.py stays runnable — it either writes the fixtures (TensorFlow present) or says that
    TensorFlow is absent and exits 0; any OTHER failure (a broken helper, a broken TensorFlow install, a changed reference) is loud."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'oracle', 'gen_golden_tf.py')], capture_output=True, text=True, timeout=600,
                         cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    said = out.stdout
    assert ('tensorflow is not importable here' in said and "No module named 'tensorflow'" in said) or 'wrote tf_caser.npz' in said or \
        'the reference tree is not present' in said, said[-2000:]


def _load(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f'{name} not generated (needs TensorFlow in the dev container: python oracle/gen_golden_tf.py)')
    return np.load(path)


def _frame(z):
    return {k[len('frame_'):]: z[k] for k in z.files if k.startswith('frame_')}


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / np.maximum(np.abs(b), 1e-12)))


@pytest.mark.parametrize('loss', ['bce', 'mse'])
def test_cdae_oracle_reproduces_tensorflow(loss):
    from oracle import cdae_oracle as co
    from oracle import data_oracle as do
    z = _load('tf_cdae.npz')
    f = _frame(z)
    uid, _ = do.first_appearance_codes(f['user'].tolist())
    iid, _ = do.first_appearance_codes(f['item'].tolist())
    U, N = int(uid.max()) + 1, int(iid.max()) + 1
    indptr, cols, vals = do.interaction_csr(uid, iid, f['interaction'], U, N)
    tmat = np.zeros((U, N), dtype=bool)
    for u in range(U):
        s, e = indptr[u], indptr[u + 1]
        tmat[u, cols[s:e][vals[s:e] >= 1e-3]] = True
    B, seed, q = int(z['B']), int(z['seed']), float(z['q'])
    for k in z['steps'].tolist():
        sampler = do.PointSamplerOracle(uid, iid, f['interaction'], int(z['neg_ratio']), 1e-3, seed)
        rng = random.Random(seed)
        p = {n: z['init_' + n].astype(np.float32).copy() for n in ('W', 'W_', 'V', 'b', 'b_')}
        st = co.adam_state(p)
        for step in range(k):
            users = np.array([t[0] for t in sampler.sample(B)])
            keep = do.corruption_keep_mask(rng, B, N, q)
            x = ((tmat[users] & keep).astype(np.float32) / np.float32(1.0 - np.float32(q))).astype(np.float32)
            co.dense_step(p, st, step, users, x, tmat[users], float(z['lr']), float(z['reg']), loss, 'reference')
        for n in p:
            assert _rel(p[n], z[f'{loss}_k{k}_{n}']) < 5 * TOL, (loss, k, n)
        pred = np.stack([co.predict_row(p, u, tmat[u]) for u in range(U)])
        assert _rel(pred, z[f'{loss}_k{k}_pred']) < TOL, (loss, k)


@pytest.mark.gpu
@pytest.mark.parametrize('loss', ['bce', 'mse'])
def test_cdae_hip_fit_reproduces_tensorflow(loss):
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    z = _load('tf_cdae.npz')
    f = _frame(z)
    w = {n: z['init_' + n] for n in ('W', 'W_', 'V', 'b', 'b_')}
    for k in z['steps'].tolist():
        m = CDAE(hidden_factors=int(z['K']), corruption_level=float(z['q']), loss=loss, seed=int(z['seed']), verbose=False)
        m.fit(InteractionDataset.read_df(f, verbose=False), epochs=k, batch_size=int(z['B']), learning_rate=float(z['lr']),
              reg_rate=float(z['reg']), neg_ratio=int(z['neg_ratio']), initial_weights=w)
        pred = np.stack([m._predict(u) for u in range(m.n_users)])
        assert _rel(pred, z[f'{loss}_k{k}_pred']) < TOL, (loss, k)


@pytest.mark.gpu
def test_dmf_hip_fit_reproduces_tensorflow():
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import DMF
    z = _load('tf_dmf.npz')
    f = _frame(z)
    w = {k[len('init_'):]: z[k] for k in z.files if k.startswith('init_')}
    for k in z['steps'].tolist():
        m = DMF(user_factors=z['user_factors'].tolist(), item_factors=z['item_factors'].tolist(), seed=int(z['seed']), verbose=False)
        m.fit(InteractionDataset.read_df(f, verbose=False), epochs=k, batch_size=int(z['B']), learning_rate=float(z['lr']),
              reg_rate=float(z['reg']), neg_ratio=int(z['neg_ratio']), initial_weights=w)
        got = np.array([float(m._predict(int(u), int(i))) for u, i in z['probe_pairs']])
        assert _rel(got, z[f'k{k}_pred']) < TOL, k


@pytest.mark.gpu
def test_caser_hip_fit_reproduces_tensorflow():
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import Caser
    z = _load('tf_caser.npz')
    f = _frame(z)
    w = {k[len('init_'):]: z[k] for k in z.files if k.startswith('init_')}
    for k in z['steps'].tolist():
        m = Caser(L=int(z['L']), T=int(z['T']), d=int(z['d']), n_v=int(z['n_v']), n_h=int(z['n_h']), dropout_rate=0.0, seed=int(z['seed']),
                  verbose=False)
        m.fit(InteractionDataset.read_df(f, verbose=False), epochs=k, batch_size=int(z['B']), learning_rate=float(z['lr']),
              reg_rate=float(z['reg']), neg_ratio=int(z['neg_ratio']), initial_weights=w)
        p = m._engine.get_params() if hasattr(m._engine, 'get_params') else {}
        for name in ('user_emb', 'item_emb'):
            if name in p:
                assert _rel(p[name], z[f'k{k}_{name}']) < 5 * TOL, (k, name)
