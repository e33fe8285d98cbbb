"""Edge cases of the CDAE kernels against the oracle: empty histories, B = 1, every sample the same user, a user whose
history is longer than several lane groups, all inputs dropped (q close to 1), K not a multiple of 4, one-item catalogue
rows at the key-space borders (first / last user and item), ragged batches (B not a multiple of the group count)."""
import numpy as np
import pytest

from oracle import cdae_oracle as co
from helpers import batch_rows, hash_u32, q_threshold, x_tilde

pytestmark = pytest.mark.gpu


def _hist(U, N, rows):
    indptr = np.zeros(U + 1, np.int64)
    idx = []
    for u in range(U):
        r = np.sort(np.asarray(rows.get(u, []), dtype=np.int64))
        idx.append(r)
        indptr[u + 1] = indptr[u] + len(r)
    return indptr, (np.concatenate(idx) if idx else np.zeros(0)).astype(np.int32)


def _run(U, N, K, indptr, indices, batches, q, mode, opt='adagrad', explicit=True, steps=None):
    from drecpy_amd.engine import CdaeEngine
    rng = np.random.default_rng(K + U)
    p = co.init_params(rng, U, N, K, np.float64)
    eng = CdaeEngine(U, N, K)
    eng.set_params(**p)
    eng.set_history(indptr, indices)
    lr = 0.05 if (mode == 'sparse' and opt == 'adagrad') else 1e-3
    eng.init_optimizer('adam' if mode == 'dense' else opt, lr, 1e-3)
    st = co.adam_state(p) if mode == 'dense' else co.sparse_state(p, opt)
    qf = float(np.float32(q))
    for step, (uids, iids, y) in enumerate(batches):
        uids = np.asarray(uids)
        B = len(uids)
        t, keep_off, _ = batch_rows(indptr, indices, uids, N)
        seed = 99 + step
        if explicit:
            keep = (rng.random(keep_off[-1]) >= q).astype(np.uint8)
        else:
            keep = np.concatenate([hash_u32(seed, np.full(keep_off[b + 1] - keep_off[b], b), np.arange(keep_off[b + 1] - keep_off[b])) >= q_threshold(q)
                                   for b in range(B)] + [np.zeros(0, bool)]).astype(np.uint8)
        _, _, kept = batch_rows(indptr, indices, uids, N, keep)
        kw = dict(keep_off=keep_off, keep=keep) if explicit else dict(mask_seed=seed)
        if mode == 'dense':
            bt, alive = eng.make_batch(uids, q=q, **kw)
            lo = co.dense_step(p, st, step, uids, x_tilde(t, kept, qf, np.float64), t, lr, 1e-3)
            lg = float(eng.step_dense(step, bt, want_loss=True).sum().item())
        else:
            bt, alive = eng.make_batch(uids, np.asarray(iids), np.asarray(y, np.float32), q=q, **kw)
            lo, _ = co.sparse_step(p, st, step, uids, np.asarray(iids), np.asarray(y, np.float64), kept, qf, lr, 1e-3, 'bce', opt)
            # odd steps go through the prepared path (touch list built ahead, sole-toucher rows updated by the forward kernel)
            prep = eng.prepare_sparse(bt) if step % 2 else None
            lg = float(eng.step_sparse(step, bt, want_loss=True, prepared=prep)[0].item())
        assert abs(lg - lo) <= 1e-4 * abs(lo) + 1e-7, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)
    tp, _, _ = batch_rows(indptr, indices, np.arange(U), N)
    _, pred = eng.forward(np.arange(U))
    _, po = co.forward(p, np.arange(U), tp.astype(np.float64))
    assert float(np.max(np.abs(pred.cpu().numpy() - po) / np.abs(po))) < 1e-5


@pytest.mark.parametrize('mode', ['dense', 'sparse'])
def test_empty_histories_and_single_sample(mode):
    U, N, K = 7, 13, 10
    indptr, indices = _hist(U, N, {1: [0, 12], 4: [5]})            # users 0,2,3,5,6 have no positives at all
    batches = [([0], [12], [0.0]), ([1], [0], [1.0]), ([6, 0, 2], [0, 12, 5], [0, 1, 0]), ([4, 4, 1, 3, 5], [5, 5, 12, 0, 1], [1, 1, 1, 0, 0])]
    _run(U, N, K, indptr, indices, batches, 0.2, mode)


@pytest.mark.parametrize('mode,K', [('dense', 7), ('sparse', 7), ('sparse', 33), ('dense', 130)])
def test_same_user_everywhere_long_history_odd_k(mode, K):
    rng = np.random.default_rng(1)
    U, N = 5, 400
    rows = {0: list(range(0, N, 2)), 1: [0], 2: list(range(N)), 4: [N - 1]}      # 200- and 400-item histories: many lane groups
    indptr, indices = _hist(U, N, rows)
    B = 37                                                                          # ragged: not a multiple of the groups per block
    batches = [([2] * B, rng.integers(0, N, size=B), (rng.random(B) < 0.5)),
               ([0] * B, [0] * B, [1.0] * B),                                      # every triple identical: one hot row per table
               (rng.integers(0, U, size=B), rng.integers(0, N, size=B), (rng.random(B) < 0.5))]
    _run(U, N, K, indptr, indices, batches, 0.3, mode, explicit=(mode == 'dense'))


@pytest.mark.parametrize('q', [0.0, 0.97])
def test_corruption_extremes_and_border_rows(q):
    rng = np.random.default_rng(2)
    U, N, K = 64, 97, 16
    rows = {u: rng.choice(N, size=rng.integers(0, 9), replace=False).tolist() for u in range(U)}
    rows[0] = [0, N - 1]
    rows[U - 1] = [N - 1]
    indptr, indices = _hist(U, N, rows)
    B = 100
    u = rng.integers(0, U, size=B); u[:4] = [0, U - 1, 0, U - 1]
    i = rng.integers(0, N, size=B); i[:4] = [0, N - 1, N - 1, 0]
    batches = [(u, i, (rng.random(B) < 0.5)) for _ in range(3)]
    _run(U, N, K, indptr, indices, batches, q, 'sparse', opt='adam', explicit=False)
    _run(U, N, K, indptr, indices, [(u, i, None)] * 2, q, 'dense')


def test_bad_arguments_are_rejected():
    import ctypes as C
    import torch
    from drecpy_amd import _lib
    from drecpy_amd.engine import CdaeEngine
    eng = CdaeEngine(4, 6, 8)
    eng.set_history(np.zeros(5, np.int64), np.zeros(0, np.int32))
    eng.init_optimizer('adagrad', 0.1, 0.0)
    bt, alive = eng.make_batch(np.array([0, 1]), np.array([0, 1]), np.array([0., 1.], np.float32), q=0.2)
    L = _lib.lib()
    o = eng._optim([0.1] * 5)
    tiny = torch.empty(64, dtype=torch.uint8, device='cuda')
    rc = L.drx_cdae_step_sparse(C.byref(eng._params), C.byref(o), C.byref(eng._hist), C.byref(bt), 0, _lib.ptr(tiny), 64, None, None)
    assert rc == -2                                  # DRX_ESCRATCH
    bt.q = 1.5
    sc = eng._ensure_scratch(2, 0)
    rc = L.drx_cdae_step_sparse(C.byref(eng._params), C.byref(o), C.byref(eng._hist), C.byref(bt), 0, _lib.ptr(sc), sc.numel(), None, None)
    assert rc == -1                                  # DRX_EINVAL
    with pytest.raises(_lib.DrxError):
        _lib.check(rc, 'drx_cdae_step_sparse')


@pytest.mark.parametrize('K', [1, 3, 13, 16, 17, 20, 33, 64, 65, 129, 200, 256, 260, 512, 600, 1000])
@pytest.mark.parametrize('mode,opt', [('sparse', 'adagrad'), ('sparse', 'adam'), ('dense', 'adam')])
def test_every_row_geometry(K, mode, opt):
    """One K per lane-group geometry and both of its borders — (4,1) ld<=16, (8,1) <=32, (16,1) <=64, (32,1) <=128, (64,1) <=256,
    (64,2) <=512, (64,4) <=1024 — incl. K with padding columns and the > 48 KB LDS of the long-span tier at K = 1000;
    hot rows that cross many chunks, sole-toucher rows (odd steps go through the prepared path), ragged batch."""
    rng = np.random.default_rng(K)
    U, N = 23, 61
    rows = {u: rng.choice(N, size=rng.integers(0, 14), replace=False).tolist() for u in range(U)}
    rows[3] = list(range(N))                                                 # one user with every item
    indptr, indices = _hist(U, N, rows)
    B = 150                                                                  # 150 triples over 61 items: every W row is hot
    batches = []
    for _ in range(3):
        u = rng.integers(0, U, size=B); u[:40] = 3
        batches.append((u, rng.integers(0, N, size=B), (rng.random(B) < 0.4)))
    _run(U, N, K, indptr, indices, batches, 0.25, mode, opt=opt, explicit=(mode == 'dense'))
