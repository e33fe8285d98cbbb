"""Independent checks of oracle/cdae_oracle.py (its TF arithmetic is 'parity unpinned', so the
restatement is cross-checked against torch-CPU autograd built literally from cdae.py:73-82)."""
import numpy as np
import torch

from oracle import cdae_oracle as co


def _setup(dtype, B=6, U=9, N=17, K=5, seed=0):
    rng = np.random.default_rng(seed)
    p = co.init_params(rng, U, N, K, dtype)
    uids = rng.integers(0, U, size=B)
    t = (rng.random((B, N)) < 0.3)
    keep = rng.random((B, N)) >= 0.2
    xt = (t & keep).astype(dtype) / dtype(0.8)
    return p, uids, t, xt


def _torch_loss(p, uids, t, xt, reg, loss):
    """Literal (B,B,N) construction of cdae.py:50-57,73-82 with torch autograd."""
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    preds = []
    for b in range(len(uids)):
        x = torch.tensor(xt[b:b + 1], dtype=torch.float64)
        h = torch.sigmoid(x @ tp['W'] + tp['V'][uids[b]] + tp['b'])
        preds.append(torch.sigmoid(h @ tp['W_'] + tp['b_']))          # (1,N)
    y_pred = torch.stack(preds)                                       # (B,1,N)
    y_true = torch.tensor(t.astype(np.float64))                       # (B,N)
    eps = 1e-7
    if loss == 'bce':
        pc = torch.clamp(y_pred, eps, 1 - eps)
        e = -(y_true * torch.log(pc + eps) + (1 - y_true) * torch.log(1 - pc + eps))   # (B,B,N)
    else:
        e = (y_pred - y_true) ** 2
    L = e.mean(dim=-1).mean()
    L = L + sum((tp[k] ** 2).sum() / 2 for k in ('W', 'W_', 'V')) * reg / len(uids)
    L.backward()
    return L.item(), {k: v.grad.numpy() for k, v in tp.items()}


def test_dense_grads_match_autograd_of_literal_broadcast():
    for loss in ('bce', 'mse'):
        p, uids, t, xt = _setup(np.float64)
        L, g, _ = co.dense_grads(p, uids, xt, t, 1e-3, loss)
        Lt, gt = _torch_loss(p, uids, t, xt, 1e-3, loss)
        assert abs(L - Lt) < 1e-12
        for k in g:
            np.testing.assert_allclose(g[k], gt[k], rtol=1e-9, atol=1e-14)
        # literal (B,B,N) value == batch-mean-target value
        _, pred = co.forward(p, uids, xt)
        lit = co.batch_loss(pred, t, loss, 'reference')
        reg = 1e-3 / len(uids) * 0.5 * sum((p[k] ** 2).sum() for k in ('W', 'W_', 'V'))
        assert abs(lit + reg - L) < 1e-12


def test_fp32_vs_fp64_predictions_after_steps():
    p64, uids, t, xt = _setup(np.float64, B=8, U=20, N=40, K=8)
    p32 = {k: v.astype(np.float32) for k, v in p64.items()}
    s64, s32 = co.adam_state(p64), co.adam_state(p32)
    for step in range(10):
        co.dense_step(p64, s64, step, uids, xt, t, 1e-3, 1e-3)
        co.dense_step(p32, s32, step, uids, xt.astype(np.float32), t, 1e-3, 1e-3)
    _, a = co.forward(p64, uids, xt)
    _, b = co.forward(p32, uids, xt.astype(np.float32))
    assert np.max(np.abs(a - b) / np.abs(a)) < 1e-4


def test_adam_counter_is_per_variable():
    # t = 5*step + j + 1 (recommender_abc.py:328-334): the first update of variable j has
    # |delta| = lr_t(j+1) * |g|/(|g| + eps*sqrt(1-b2)...) ~ lr * sqrt(1-b2^t)/(1-b1^t) * (1-b1)/sqrt(1-b2)
    p, uids, t, xt = _setup(np.float64)
    before = {k: v.copy() for k, v in p.items()}
    st = co.adam_state(p)
    _, g, _ = co.dense_grads(p, uids, xt, t, 1e-3)
    co.dense_step(p, st, 0, uids, xt, t, 1e-3, 1e-3)
    for j, k in enumerate(co.VAR_ORDER):
        tt = j + 1
        a = co.adam_alpha(1e-3, tt)
        want = before[k] - a * (co.ADAM_OMB1 * g[k]) / (np.sqrt(co.ADAM_OMB2 * g[k] ** 2) + co.ADAM_EPS)
        np.testing.assert_allclose(p[k], want, rtol=1e-12)
    # TF forms (1 - beta) in fp32: 1.0f - 0.999f is NOT 0.001
    assert co.ADAM_OMB2 == 0.0009999871253967285 and co.ADAM_OMB1 == 0.10000002384185791


def test_sparse_step_matches_autograd():
    rng = np.random.default_rng(3)
    U, N, K, B = 12, 20, 6, 10
    p = co.init_params(rng, U, N, K, np.float64)
    uids = rng.integers(0, U, size=B)
    iids = rng.integers(0, N, size=B)
    y = (rng.random(B) < 0.4).astype(np.float64)
    kept = [sorted(rng.choice(N, size=rng.integers(0, 6), replace=False).tolist()) for _ in range(B)]
    q = 0.2
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    ls = []
    for b in range(B):
        x = torch.zeros(1, N, dtype=torch.float64)
        x[0, kept[b]] = 1 / (1 - q)
        h = torch.sigmoid(x @ tp['W'] + tp['V'][uids[b]] + tp['b'])
        pr = torch.sigmoid(h @ tp['W_'][:, iids[b]] + tp['b_'][iids[b]])
        pc = torch.clamp(pr, 1e-7, 1 - 1e-7)
        ls.append(-(y[b] * torch.log(pc + 1e-7) + (1 - y[b]) * torch.log(1 - pc + 1e-7)))
    L = torch.stack(ls).mean()
    L.backward()
    g = {k: v.grad.numpy() for k, v in tp.items()}
    # touched-row L2 (once per row per step)
    rb = 1e-3 / B
    tw = sorted(set(n for kk in kept for n in kk))
    g['W'][tw] += rb * p['W'][tw]
    tu = sorted(set(uids.tolist()))
    g['V'][tu] += rb * p['V'][tu]
    ti = sorted(set(iids.tolist()))
    g['W_'][:, ti] += rb * p['W_'][:, ti]
    want = {}
    for k in p:
        acc = np.full_like(p[k], co.ADAGRAD_INIT) + g[k] ** 2
        want[k] = p[k] - float(np.float32(0.05)) * g[k] / (np.sqrt(acc) + co.ADAGRAD_EPS)
    st = co.sparse_state(p, 'adagrad')
    lval, _ = co.sparse_step(p, st, 0, uids, iids, y, kept, q, 0.05, 1e-3, 'bce', 'adagrad')
    assert abs(lval - L.item()) < 1e-12
    for k in p:
        np.testing.assert_allclose(p[k], want[k], rtol=1e-9, atol=1e-13)


def test_sparse_step_matrix_accumulation_equals_the_loop():
    """accumulate='matrix' (the row gradients as sparse matrix products, for batches the Python loop cannot follow) is the same step
    as the definition ('loop'): three steps, Adagrad and lazy Adam, users and items that repeat, an empty history."""
    import copy
    for opt in ('adagrad', 'adam'):
        rng = np.random.default_rng(8)
        U, N, K, B = 14, 23, 7, 40
        p0 = co.init_params(rng, U, N, K, np.float64)
        pa, pb = copy.deepcopy(p0), copy.deepcopy(p0)
        sa, sb = co.sparse_state(pa, opt), co.sparse_state(pb, opt)
        for step in range(3):
            uids = rng.integers(0, U, size=B)
            iids = rng.integers(0, N, size=B)
            y = (rng.random(B) < 0.4).astype(np.float64)
            kept = [sorted(rng.choice(N, size=rng.integers(0, 7), replace=False).tolist()) for _ in range(B)]
            kept[3] = []
            la, _ = co.sparse_step(pa, sa, step, uids, iids, y, kept, 0.2, 0.05, 1e-3, 'bce', opt)
            lb, _ = co.sparse_step(pb, sb, step, uids, iids, y, kept, 0.2, 0.05, 1e-3, 'bce', opt, accumulate='matrix')
            assert abs(la - lb) < 1e-13
        for k in pa:
            np.testing.assert_allclose(pb[k], pa[k], rtol=1e-12, atol=1e-14, err_msg=f'{opt} {k}')
