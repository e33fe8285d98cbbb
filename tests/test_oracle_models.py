"""Cross-checks of the Caser / DMF NumPy restatements against torch-CPU autograd built literally from caser.py / dmf.py
(their TF arithmetic is 'parity unpinned')."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import caser_oracle as ca


def _caser_torch(p, uids, before, after, T, reg, keep, rate, act_h=torch.relu, act_mlp=torch.relu):
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    B, L = before.shape
    E = tp['item_emb'][torch.as_tensor(before)]                                  # [B,L,d]
    x = E.transpose(1, 2)                                                         # conv1d wants [B,C,W]
    out_v = F.conv1d(x, tp['conv_v_k'].permute(2, 1, 0), tp['conv_v_b']).reshape(B, -1)
    outs = []
    for i in range(L):
        c = act_h(F.conv1d(x, tp[f'conv_h{i}_k'].permute(2, 1, 0), tp[f'conv_h{i}_b']))   # [B,n_h,L-i]
        outs.append(c.max(dim=2).values)
    cat0 = torch.cat([out_v] + outs, dim=1)
    if keep is not None:
        cat0 = torch.where(torch.as_tensor(keep), cat0 / (1 - rate), torch.zeros_like(cat0))
    z = act_mlp(cat0 @ tp['dense0_k'] + tp['dense0_b'])
    cat = torch.cat([z, tp['user_emb'][torch.as_tensor(uids)]], dim=1).unsqueeze(1)
    w = tp['W1'][torch.as_tensor(after)]
    b = tp['b1'][torch.as_tensor(after)]
    pred = torch.sigmoid((cat * w).sum(2) + b.squeeze(2))
    y = torch.zeros_like(pred); y[:, :T] = 1
    eps = 1e-7
    pc = pred.clamp(eps, 1 - eps)
    loss = (-(y * torch.log(pc + eps) + (1 - y) * torch.log(1 - pc + eps))).mean(dim=-1).mean()
    for name, v in tp.items():
        if ca.REGULARISED(name):
            loss = loss + reg * (v ** 2).sum()
    loss.backward()
    return loss.item(), {k: v.grad.numpy() for k, v in tp.items()}


def test_caser_grads_match_autograd():
    rng = np.random.default_rng(0)
    U, N, L, d, n_v, n_h, T, neg, B = 9, 31, 5, 6, 3, 4, 2, 2, 7
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    for k in p:                                     # non-zero biases so their gradients are exercised
        if k.endswith('_b'):
            p[k] = rng.normal(0, 0.1, size=p[k].shape)
    uids = rng.integers(0, U, size=B)
    before = rng.integers(0, N, size=(B, L))
    after = rng.integers(0, N, size=(B, T + T * neg))
    keep = rng.random((B, n_v + L * n_h)) >= 0.5
    for kp, rate in ((None, 0.0), (keep, 0.5)):
        lo, g, _ = ca.loss_and_grads(p, uids, before, after, T, 1e-3, kp, rate)
        lt, gt = _caser_torch(p, uids, before, after, T, 1e-3, kp, rate)
        assert abs(lo - lt) < 1e-12
        for k in g:
            np.testing.assert_allclose(g[k], gt[k], rtol=1e-9, atol=1e-13, err_msg=k)
    # act_h / act_mlp other than relu (caser.py:29-30): tanh convolutions, sigmoid dense_0; linear
    for ah, am, th, tm in (('tanh', 'sigmoid', torch.tanh, torch.sigmoid), ('linear', 'tanh', lambda v: v, torch.tanh)):
        lo, g, _ = ca.loss_and_grads(p, uids, before, after, T, 1e-3, keep, 0.5, ah, am)
        lt, gt = _caser_torch(p, uids, before, after, T, 1e-3, keep, 0.5, th, tm)
        assert abs(lo - lt) < 1e-12
        for k in g:
            np.testing.assert_allclose(g[k], gt[k], rtol=1e-9, atol=1e-13, err_msg=(ah, am, k))


def test_dmf_grads_match_autograd():
    from oracle import dmf_oracle as dm
    rng = np.random.default_rng(1)
    U, N, B = 13, 17, 9
    uf, itf = (8, 5), (7, 6, 5)
    p = dm.init_params(rng, U, N, uf, itf, np.float64)
    for k in p:
        if k.endswith('_b'):
            p[k] = rng.normal(0, 0.1, size=p[k].shape)
    xu = rng.integers(0, 6, size=(B, N)).astype(np.float64) * (rng.random((B, N)) < 0.4)
    xi = rng.integers(0, 6, size=(B, U)).astype(np.float64) * (rng.random((B, U)) < 0.4)
    xu[0] = 0                                         # an all-zero input row exercises the 1e-12 clamp
    y = rng.random(B)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}

    def l2n(x):
        return x * torch.rsqrt(torch.clamp((x * x).sum(dim=1, keepdim=True), min=1e-12))

    def tower(t, x, n):
        for l in range(n):
            x = torch.relu(x @ tp[f'{t}{l}_k'] + tp[f'{t}{l}_b'])
        return x
    ru = tower('u', l2n(torch.tensor(xu)), len(uf))
    ri = tower('i', l2n(torch.tensor(xi)), len(itf))
    pred = torch.clamp((l2n(ru) * l2n(ri)).sum(dim=1), min=1e-6)
    eps = 1e-7
    pc = pred.clamp(eps, 1 - eps)
    yt = torch.tensor(y)
    loss = (-(yt * torch.log(pc + eps) + (1 - yt) * torch.log(1 - pc + eps))).mean()
    loss = loss + sum(1e-3 * (v ** 2).sum() for k, v in tp.items() if k.endswith('_k'))
    loss.backward()
    lo, g, _ = dm.loss_and_grads(p, xu, xi, y, 1e-3, len(uf), len(itf))
    assert abs(lo - loss.item()) < 1e-12
    for k in g:
        np.testing.assert_allclose(g[k], tp[k].grad.numpy(), rtol=1e-9, atol=1e-13, err_msg=k)


def test_modified_dmf_grads_match_autograd_of_the_literal_broadcast():
    """ModifiedDMF (examples/extending_recommender_dmf.py:9-18): predictions = [extra_w * pred for pred in predictions] is a list
    of B (1,)-tensors -> (B,1); Keras BCE against the (B,) targets broadcasts to (B,B), mean(axis=-1) then the batch mean.
    Built literally in torch; also the identity the HIP kernel uses: that loss equals the BCE against the batch-MEAN target."""
    from oracle import cdae_oracle as co
    from oracle import dmf_oracle as dm
    rng = np.random.default_rng(3)
    U, N, B = 11, 19, 8
    uf, itf = (8, 5), (6, 5)
    p = dm.init_params(rng, U, N, uf, itf, np.float64)
    p['extra_w'] = np.array([0.83])
    xu = rng.integers(0, 6, size=(B, N)).astype(np.float64) * (rng.random((B, N)) < 0.5)
    xi = rng.integers(0, 6, size=(B, U)).astype(np.float64) * (rng.random((B, U)) < 0.5)
    y = rng.random(B)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}

    def l2n(x):
        return x * torch.rsqrt(torch.clamp((x * x).sum(dim=1, keepdim=True), min=1e-12))

    def tower(t, x, n):
        for l in range(n):
            x = torch.relu(x @ tp[f'{t}{l}_k'] + tp[f'{t}{l}_b'])
        return x
    pred = torch.clamp((l2n(tower('u', l2n(torch.tensor(xu)), 2)) * l2n(tower('i', l2n(torch.tensor(xi)), 2))).sum(dim=1), min=1e-6)
    preds = torch.stack([tp['extra_w'] * pr for pr in pred])              # list of (1,) tensors -> (B,1)
    assert tuple(preds.shape) == (B, 1)
    eps = 1e-7
    pc = preds.clamp(eps, 1 - eps)
    yt = torch.tensor(y)                                                   # (B,) broadcasts along the last axis -> (B,B)
    elem = -(yt * torch.log(pc + eps) + (1 - yt) * torch.log(1 - pc + eps))
    assert tuple(elem.shape) == (B, B)
    loss = elem.mean(dim=-1).mean() + sum(1e-3 * (v ** 2).sum() for k, v in tp.items() if k.endswith('_k'))
    loss.backward()
    lo, g, po = dm.loss_and_grads(p, xu, xi, y, 1e-3, 2, 2, True, broadcast_targets=True)
    assert abs(lo - loss.item()) < 1e-12
    assert set(g) == set(p)
    for k in g:
        np.testing.assert_allclose(g[k], tp[k].grad.numpy(), rtol=1e-9, atol=1e-13, err_msg=k)
    # the kernel's form: BCE against the batch-mean target (the element is affine in t)
    dt = np.dtype(np.float64)
    reg = sum(1e-3 * (v * v).sum() for k, v in p.items() if k.endswith('_k'))
    assert abs(co.bce_elem(dt.type(y.mean()), po, dt).mean() + reg - lo) < 1e-12
    # Adam order (recommender_abc.py:194-196,328-334): the scalar first, then user_nn, then item_nn -> t = 3s+1, 3s+2, 3s+3
    st = dm.adam_state(p)
    before = {k: v.copy() for k, v in p.items()}
    dm.step(p, st, 4, xu, xi, y, 1e-3, 1e-3, 2, 2, True, broadcast_targets=True)
    for name, j in (('extra_w', 0), ('u0_k', 1), ('i1_b', 2)):
        a = co.adam_alpha(1e-3, 3 * 4 + j + 1)
        m = g[name] * co.ADAM_OMB1
        v = g[name] * g[name] * co.ADAM_OMB2
        np.testing.assert_allclose(p[name], before[name] - (m * a) / (np.sqrt(v) + co.ADAM_EPS), rtol=1e-12, atol=0, err_msg=name)
