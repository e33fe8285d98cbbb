"""Every BASELINE.json configuration through the HIP path AT ITS REAL SHAPE:

  config 1  reference-mode CDAE step, ml-100k shape 943 x 1682, K = 50, B = 64  (examples/cdae.py)          vs fp64 oracle, 1/10/50 steps
  config 2  reference-mode CDAE step, ml-1m shape 6040 x 3706, K = 128, B = 64                              vs fp64 oracle, 1/10/50 steps
            the same tables in the sampled-output / sparse-Adagrad engine mode, B = 4096, inline and prepared     vs fp64 oracle, 3 steps
  config 3  DMF [64,32] / [64,32] at 6040 x 3706, B = 256 and 4096 (+ the ModifiedDMF scalar, test_gpu_dmf)    vs dmf_oracle, 3 steps
            bf16-MFMA all-pairs scorer, 2048 users x 3706 items                                           vs fp32 and vs bf16-rounded fp32
  config 5  Caser d=50 L=5 T=3 n_v=4 n_h=16 at 6040 x 3706, B = 4096 (examples/caser.py:13-14)            vs caser_oracle, 2 steps
  config 4  the 10M-user x 1M-item set at 10M users: tests/test_gpu_fullsize.py (properties; no oracle follows there)

Synthetic interaction sets of the MovieLens shapes (drecpy_amd.synth; no MovieLens files exist offline).  Tolerances are
those of the small-shape parity tests: predictions 1e-5 relative vs the fp64 oracle (after 50 dense-Adam steps:
max(1e-5, 2 x the fp32 oracle's own drift), < 5e-5), parameters to an absolute 5e-5 / 3e-5."""
import numpy as np
import pytest

from oracle import caser_oracle as ca
from oracle import cdae_oracle as co
from oracle import data_oracle as do
from oracle import dmf_oracle as dm
from helpers import batch_rows, x_tilde

pytestmark = pytest.mark.gpu
REL = 1e-5


def _relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def _history(shape):
    from drecpy_amd import synth
    U, N, md, mn, a = synth.SHAPES[shape]
    ip, idx = synth.synth_history(U, N, md, mn, a, seed=0)
    return U, N, ip.numpy(), idx.numpy()


@pytest.mark.parametrize('shape,K', [('ml-100k', 50), ('ml-1m', 128)])
def test_reference_mode_cdae_at_baseline_shape(shape, K):
    """configs 1 and 2: the reference step (all N output units, (B,B,N) loss, L2/B on full tables, 5 Keras-Adam applies per step)
    at the full table sizes, B = 64 (examples/cdae.py:12), 50 steps, against the fp64 oracle."""
    from drecpy_amd.engine import CdaeEngine
    U, N, indptr, indices = _history(shape)
    assert (U, N) == ((943, 1682) if shape == 'ml-100k' else (6040, 3706))
    rng = np.random.default_rng(11)
    p = co.init_params(rng, U, N, K, np.float64)
    eng = CdaeEngine(U, N, K)
    eng.set_params(**p)
    eng.set_history(indptr, indices)
    eng.init_optimizer('adam', 1e-3, 1e-3)
    st = co.adam_state(p)
    p32 = {k: v.astype(np.float32) for k, v in p.items()}
    st32 = co.adam_state(p32)
    B, q = 64, 0.2
    qf = float(np.float32(q))
    probe = rng.integers(0, U, size=32)
    tp, _, _ = batch_rows(indptr, indices, probe, N)
    for step in range(50):
        uids = rng.integers(0, U, size=B)
        t, keep_off, _ = batch_rows(indptr, indices, uids, N)
        keep = (rng.random(keep_off[-1]) >= q).astype(np.uint8)
        _, _, kept = batch_rows(indptr, indices, uids, N, keep)
        bt, alive = eng.make_batch(uids, keep_off=keep_off, keep=keep, q=q)
        lo = co.dense_step(p, st, step, uids, x_tilde(t, kept, qf, np.float64), t, 1e-3, 1e-3, 'bce', 'reference')
        co.dense_step(p32, st32, step, uids, x_tilde(t, kept, qf, np.float32), t, 1e-3, 1e-3, 'bce', 'reference')
        lg = eng.step_dense(step, bt, 'bce', 'reference', want_loss=True).cpu().numpy()
        assert abs(lg.sum() - lo) / abs(lo) < 1e-4, (step, lg, lo)
        if step + 1 in (1, 10, 50):
            _, pred = eng.forward(probe)
            _, po = co.forward(p, probe, tp.astype(np.float64))
            _, po32 = co.forward(p32, probe, tp.astype(np.float32))
            err, drift32 = _relerr(pred.cpu().numpy(), po), _relerr(po32, po)
            assert err < max(REL, 2.0 * drift32), (step, err, drift32)
            assert err < 5e-5
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=5e-5, err_msg=k)


@pytest.mark.parametrize('prepared', [False, True])
def test_sampled_mode_cdae_at_ml1m_shape(prepared):
    """BASELINE config 1's engine mode at its shape: CDAE K = 128 on 6040 x 3706, one sampled output unit per triple, sparse Adagrad —
    4096 triples per step (165-item histories: 0.54 M touches, every item row shared by ~150 samples), touch list inline and prepared
    ahead (sole-toucher rows updated by the forward kernel), against the fp64 oracle."""
    from drecpy_amd.engine import CdaeEngine
    from helpers import hash_u32, q_threshold
    U, N, indptr, indices = _history('ml-1m')
    K, B, q, lr = 128, 4096, 0.2, 0.05
    rng = np.random.default_rng(17)
    p = co.init_params(rng, U, N, K, np.float64)
    eng = CdaeEngine(U, N, K)
    eng.set_params(**p)
    eng.set_history(indptr, indices)
    eng.init_optimizer('adagrad', lr, 1e-3)
    st = co.sparse_state(p, 'adagrad')
    qf, thr = float(np.float32(q)), q_threshold(q)
    probe = rng.integers(0, U, size=32)
    tp, _, _ = batch_rows(indptr, indices, probe, N)
    for step in range(3):
        uids = rng.integers(0, U, size=B)
        iids = rng.integers(0, N, size=B)
        y = (rng.random(B) < 0.3).astype(np.float32)
        _, keep_off, _ = batch_rows(indptr, indices, uids, N)
        seed = 4242 + step * 7919
        deg = np.diff(keep_off)
        keep = (hash_u32(seed, np.repeat(np.arange(B), deg), np.arange(keep_off[-1]) - np.repeat(keep_off[:-1], deg)) >= thr).astype(np.uint8)
        _, _, kept = batch_rows(indptr, indices, uids, N, keep)
        bt, alive = eng.make_batch(uids, iids, y, q=q, mask_seed=seed)
        lo, _ = co.sparse_step(p, st, step, uids, iids, y, kept, qf, lr, 1e-3, 'bce', 'adagrad')
        lg = eng.step_sparse(step, bt, 'bce', want_loss=True, prepared=eng.prepare_sparse(bt) if prepared else None).cpu().numpy()
        assert abs(lg[0] - lo) / abs(lo) < 1e-4, (step, lg, lo)
    _, pred = eng.forward(probe)
    _, po = co.forward(p, probe, tp.astype(np.float64))
    assert _relerr(pred.cpu().numpy(), po) < REL
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=2e-5, err_msg=k)


def test_shared_form_at_the_bench_batch_on_the_ml1m_shape_matches_the_oracle():
    """The configuration bench.py times as `cfg2`: CDAE K = 128 on 6040 x 3706, B = 65 536 triples per step in USER order, lists
    prepared through the history's transpose in the shared form (DRX_BATCH_SHARE_USERS: ~11 triples of a user share their gather —
    the fp32 MFMA masked product — and their gradient row), sparse Adagrad.  Two steps against the fp64 oracle (row gradients as
    sparse matrix products: cdae_oracle.sparse_step accumulate='matrix' — 8.6 M row additions per step), then the 1e-5 gate on the
    predictions of a probe of users and 2e-5 on every parameter."""
    from drecpy_amd.engine import CdaeEngine
    from helpers import hash_u32, q_threshold
    U, N, indptr, indices = _history('ml-1m')
    K, B, q, lr = 128, 65536, 0.2, 0.05
    rng = np.random.default_rng(23)
    p = co.init_params(rng, U, N, K, np.float64)
    eng = CdaeEngine(U, N, K)
    eng.set_params(**p)
    eng.set_history(indptr, indices)
    assert eng.share_users and eng._hist_t is not None
    eng.init_optimizer('adagrad', lr, 1e-3)
    st = co.sparse_state(p, 'adagrad')
    qf, thr = float(np.float32(q)), q_threshold(q)
    for step in range(2):
        uids = np.sort(rng.integers(0, U, size=B))
        iids = rng.integers(0, N, size=B)
        y = (rng.random(B) < 0.3).astype(np.float32)
        deg = (indptr[uids + 1] - indptr[uids]).astype(np.int64)
        keep_off = np.zeros(B + 1, dtype=np.int64)
        keep_off[1:] = np.cumsum(deg)
        row = np.repeat(np.arange(B), deg)
        j = np.arange(keep_off[-1]) - keep_off[:-1][row]
        seed = 9100 + step * 7919
        keep = hash_u32(seed, row, j) >= thr
        items = indices[indptr[uids][row] + j]
        kept_flat, cuts = items[keep], np.cumsum(np.bincount(row[keep], minlength=B))[:-1]
        kept = np.split(kept_flat, cuts)
        bt, alive = eng.make_batch(uids, iids, y, q=q, mask_seed=seed)
        assert bt.n_touch_slots + 2 * B > 8 * (2 * N + U) and (bt.flags & 1)               # long segments, DRX_BATCH_SHARE_USERS
        lo, _ = co.sparse_step(p, st, step, uids, iids, y, kept, qf, lr, 1e-3, 'bce', 'adagrad', accumulate='matrix')
        lg = eng.step_sparse(step, bt, 'bce', want_loss=True, prepared=eng.prepare_sparse(bt)).cpu().numpy()
        assert abs(lg[0] - lo) / abs(lo) < 1e-4, (step, lg, lo)
    probe = rng.integers(0, U, size=64)
    tp, _, _ = batch_rows(indptr, indices, probe, N)
    _, pred = eng.forward(probe)
    _, po = co.forward(p, probe, tp.astype(np.float64))
    assert _relerr(pred.cpu().numpy(), po) < REL
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=2e-5, err_msg=k)


def _ml1m_ratings():
    """Interaction matrix of the ml-1m shape with ratings 1..5: CSR, CSC and the dense fp64 matrix the oracle reads."""
    U, N, indptr, indices = _history('ml-1m')
    rng = np.random.default_rng(5)
    u = np.repeat(np.arange(U), np.diff(indptr))
    i = indices.astype(np.int64)
    v = rng.integers(1, 6, size=len(u)).astype(np.float64)
    csr = do.interaction_csr(u, i, v, U, N)
    csc = do.interaction_csr(i, u, v, N, U)
    dense = np.zeros((U, N))
    dense[u, i] = v
    return U, N, csr, csc, dense


@pytest.mark.parametrize('B', [256, 4096])
def test_dmf_at_ml1m_shape(B):
    """config 3's towers ([64,32] / [64,32], extending_recommender_dmf.py builds on the DMF defaults, dmf.py:25-33) at
    U = 6040, N = 3706: three steps against the fp64 oracle."""
    from drecpy_amd.engine_dmf import DmfEngine
    U, N, csr, csc, dense = _ml1m_ratings()
    rng = np.random.default_rng(B)
    p = dm.init_params(rng, U, N, (64, 32), (64, 32), np.float64)
    eng = DmfEngine(U, N, (64, 32), (64, 32), True)
    eng.set_interactions(csr, csc)
    eng.set_params(p)
    eng.lr, eng.reg = 1e-3, 1e-3
    st = dm.adam_state(p)
    for step in range(3):
        uids = rng.integers(0, U, size=B)
        iids = rng.integers(0, N, size=B)
        y = (dense[uids, iids] / 5.0) if step else rng.random(B)          # standardised ratings (dmf.py:69) / arbitrary targets
        lo = dm.step(p, st, step, dense[uids], dense[:, iids].T.copy(), y, 1e-3, 1e-3, 2, 2, True)
        lg = eng.step(step, uids, iids, y, want_loss=True)
        assert abs(lg - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)
    uids = rng.integers(0, U, size=512)
    iids = rng.integers(0, N, size=512)
    pred = eng.predict(uids, iids).cpu().numpy()
    want, _ = dm.forward(p, dense[uids], dense[:, iids].T.copy(), 2, 2, True)
    assert np.max(np.abs(pred - want) / np.maximum(np.abs(want), 1e-6)) < 1e-4


def test_mfma_scorer_2048_users_by_all_ml1m_items():
    """The bf16-MFMA all-pairs scorer at config 3's size: 2048 users x 3706 items.  Against the fp32 cosine to bf16 accuracy,
    and against the SAME product with operands rounded to bf16 first (what the matrix cores are given) to fp32 accuracy."""
    import torch
    from drecpy_amd.engine_dmf import DmfEngine
    U, N, csr, csc, dense = _ml1m_ratings()
    p = dm.init_params(np.random.default_rng(4), U, N, (64, 32), (64, 32), np.float64)
    eng = DmfEngine(U, N, (64, 32), (64, 32), True)
    eng.set_interactions(csr, csc)
    eng.set_params(p)
    uids = np.random.default_rng(6).choice(U, size=2048, replace=False)
    sc = eng.score_matrix_bf16(uids)
    assert tuple(sc.shape) == (2048, N)
    _, ru, _ = eng.predict(uids, np.zeros(2048, np.int64), want_reps=True)
    _, _, ri = eng.predict(np.zeros(N, np.int64), np.arange(N), want_reps=True)
    f32 = torch.clamp(ru[:, :32].double() @ ri[:, :32].double().t(), min=1e-6)
    assert float((sc.double() - f32).abs().max()) < 1.5e-2                       # 8 mantissa bits per operand
    rb = torch.clamp(ru[:, :32].bfloat16().double() @ ri[:, :32].bfloat16().double().t(), min=1e-6)
    assert float((sc.double() - rb).abs().max()) < 2e-6                          # fp32 accumulation of exact bf16 products
    # rows against the oracle's own forward (fp64 towers): the scorer ranks what DMF._predict would
    for r in (0, 777, 2047):
        u = int(uids[r])
        want, _ = dm.forward(p, np.repeat(dense[u:u + 1], N, axis=0), dense.T.copy(), 2, 2, True)
        assert np.max(np.abs(sc[r].cpu().numpy() - want)) < 1.5e-2


def test_caser_at_ml1m_shape_b4096():
    """config 5: Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout 0.5), batch_size = 4096 (examples/caser.py:13-14) at U = 6040,
    N = 3706; two steps with injected dropout masks against the fp64 oracle."""
    from drecpy_amd.engine_caser import CaserEngine
    U, N, L, T, d, n_v, n_h, neg, B = 6040, 3706, 5, 3, 50, 4, 16, 3, 4096
    rng = np.random.default_rng(17)
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    eng = CaserEngine(U, N, L, T, neg, d, n_v, n_h)
    eng.set_params(p)
    eng.lr, eng.reg = 1e-3, 1e-4
    st = ca.adam_state(p)
    nx = n_v + L * n_h
    pop = 1.0 / np.arange(1, N + 1)
    pop /= pop.sum()
    for step in range(2):
        uids = rng.integers(0, U, size=B)
        before = rng.choice(N, size=(B, L), p=pop)                   # popular items repeat inside the batch (hot embedding rows)
        after = rng.choice(N, size=(B, T + T * neg), p=pop)
        keep = rng.random((B, nx)) >= 0.5
        lo = ca.step(p, st, step, uids, before, after, T, 1e-3, 1e-4, keep, 0.5)
        lg = eng.step(step, uids, before, after, keep, 0.5, want_loss=True)
        assert abs(lg - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)
    uids = rng.integers(0, U, size=3)
    before = rng.integers(0, N, size=(3, L))
    sc = eng.scores_all(uids, before).cpu().numpy()
    for r in range(3):
        want = ca.rank_scores(p, uids[r], before[r])
        assert np.max(np.abs(sc[r] - want)) < 2e-5 * max(1.0, np.max(np.abs(want)))
