"""NumPy restatement of the reference Caser step (TEST INFRASTRUCTURE, see oracle/__init__.py — **parity unpinned**:
TensorFlow/Keras arithmetic, no reference test or golden vector exists; cross-checked against torch-CPU autograd in
tests/test_oracle_models.py).

Follows DRecPy/Recommender/caser.py (paths under /root/reference):
  embeddings ........ caser.py:47-50,66-69  Keras Embedding, uniform(-0.05, 0.05) init, l2 on the whole matrices
  conv_v ............ caser.py:53,103       Conv1D(n_v, kernel_size=L): kernel [L,d,n_v] -> [B,1,n_v] (sums over d)
  convs_h[i] ........ caser.py:55-58,106-108 Conv1D(n_h, kernel_size=i+1) + act_h (relu) + max over time
                      (tf.nn.max_pool1d(ksize=n_h, strides=n_h, 'SAME') on <= L <= n_h steps == global max, App. A.6)
  dropout ........... caser.py:61,114       TF-RNG mask; here the keep mask is INJECTED (or None = no dropout)
  dense_0 ........... caser.py:63,114       Dense(d, relu) on the 84-vector [out_v, out_h]
  target scores ..... caser.py:115-120      sum(concat([dense_0, user_emb]) * W1[iid], -1) + b1[iid]
  loss .............. caser.py:86-95        Keras BCE of sigmoid(scores) vs [1]*T + [0]*(T*neg), mean over (B, T')
  regularisation .... Keras l2(reg) = reg * sum(w^2) on both embedding tables, all conv kernels, the dense_0 kernel and
                      dense_1_W; biases and dense_1_b unregularised (caser.py:46-69, recommender_abc.py:325-326)
  optimizer ......... one apply_gradients per registered layer (6 + L calls per step): Adam t = (6+L)*step + j + 1 with
                      j in registration order user_emb, item_emb, conv_v, convs_h[0..L-1], dense_0, dense_1_W, dense_1_b
                      (recommender_abc.py:328-334, caser.py:47-70)
"""
import numpy as np

from . import cdae_oracle as co


def init_params(rng, n_users, n_items, L=5, d=50, n_v=4, n_h=16, dtype=np.float32):
    u = lambda *s: rng.uniform(-0.05, 0.05, size=s).astype(dtype)              # Keras Embedding default
    p = {'user_emb': u(n_users, d), 'item_emb': u(n_items, d),
         'conv_v_k': co.glorot_uniform_conv(rng, (L, d, n_v), dtype), 'conv_v_b': np.zeros(n_v, dtype)}
    for i in range(L):
        p[f'conv_h{i}_k'] = co.glorot_uniform_conv(rng, (i + 1, d, n_h), dtype)
        p[f'conv_h{i}_b'] = np.zeros(n_h, dtype)
    p['dense0_k'] = co.glorot_uniform(rng, (n_v + L * n_h, d), dtype)
    p['dense0_b'] = np.zeros(d, dtype)
    p['W1'] = u(n_items, 2 * d)
    p['b1'] = u(n_items, 1)
    return p


ACT = {'relu': (lambda v: np.maximum(v, 0), lambda v: (v > 0).astype(v.dtype)),
       'tanh': (np.tanh, lambda v: 1 - np.tanh(v) ** 2),
       'sigmoid': (co.sigmoid, lambda v: co.sigmoid(v) * (1 - co.sigmoid(v))),
       'linear': (lambda v: v, lambda v: np.ones_like(v))}      # act_h / act_mlp (caser.py:29-30,57,63): value, derivative


def layer_order(L):
    """registration order -> list of (layer, [param names])"""
    order = [('user_emb', ['user_emb']), ('item_emb', ['item_emb']), ('conv_v', ['conv_v_k', 'conv_v_b'])]
    order += [(f'conv_h{i}', [f'conv_h{i}_k', f'conv_h{i}_b']) for i in range(L)]
    order += [('dense0', ['dense0_k', 'dense0_b']), ('W1', ['W1']), ('b1', ['b1'])]
    return order


REGULARISED = lambda name: name.endswith('_k') or name in ('user_emb', 'item_emb', 'W1')


def forward(p, uids, before, after, keep=None, rate=0.0, act_h='relu', act_mlp='relu'):
    """Scores [B, T'] (pre-sigmoid) + cache.  keep: dropout keep mask [B, n_v + L*n_h] (bool) or None."""
    dt = p['item_emb'].dtype
    E = p['item_emb'][before]                     # [B, L, d]
    B, L, d = E.shape
    out_v = np.einsum('btc,tcf->bf', E, p['conv_v_k']) + p['conv_v_b']
    outs, arg, pre = [], [], []
    for i in range(L):
        k = p[f'conv_h{i}_k']                     # [i+1, d, n_h]
        c = np.stack([np.einsum('bsc,scf->bf', E[:, t:t + i + 1], k) for t in range(L - i)], axis=1) + p[f'conv_h{i}_b']
        r = ACT[act_h][0](c)                      # [B, L-i, n_h]
        a = r.argmax(axis=1)                      # first maximum, like max-pool's gradient routing
        outs.append(np.take_along_axis(r, a[:, None, :], axis=1)[:, 0])
        arg.append(a)
        pre.append(c)
    x = np.concatenate([out_v] + outs, axis=1)    # [B, 84]
    if keep is not None:
        xd = np.where(keep, x / dt.type(1.0 - rate), dt.type(0))
    else:
        xd = x
    z0 = xd @ p['dense0_k'] + p['dense0_b']
    z = ACT[act_mlp][0](z0)
    cat = np.concatenate([z, p['user_emb'][uids]], axis=1)            # [B, 2d]
    w = p['W1'][after]                                                 # [B, T', 2d]
    scores = np.einsum('bk,bjk->bj', cat, w) + p['b1'][after][:, :, 0]
    return scores, dict(E=E, out_v=out_v, arg=arg, pre=pre, x=x, xd=xd, z0=z0, z=z, cat=cat, w=w)


def loss_and_grads(p, uids, before, after, T, reg_rate, keep=None, rate=0.0, act_h='relu', act_mlp='relu'):
    dt = p['item_emb'].dtype
    B, Tp = after.shape
    L = before.shape[1]
    d = p['item_emb'].shape[1]
    n_v = p['conv_v_k'].shape[2]
    n_h = p['conv_h0_k'].shape[2]
    scores, c = forward(p, uids, before, after, keep, rate, act_h, act_mlp)
    pred = co.sigmoid(scores)
    y = np.zeros((B, Tp), dt); y[:, :T] = 1
    lval = co.bce_elem(y, pred, dt).mean(axis=-1).mean()
    dpred = co.bce_grad(y, pred, dt) / dt.type(B * Tp)
    ds = dpred * pred * (1 - pred)                                     # [B, T']
    g = {k: np.zeros_like(v) for k, v in p.items()}
    np.add.at(g['b1'][:, 0], after, ds)
    np.add.at(g['W1'], after, ds[:, :, None] * c['cat'][:, None, :])
    dcat = np.einsum('bj,bjk->bk', ds, c['w'])
    np.add.at(g['user_emb'], uids, dcat[:, d:])
    dz0 = dcat[:, :d] * ACT[act_mlp][1](c['z0'])
    g['dense0_k'] = c['xd'].T @ dz0
    g['dense0_b'] = dz0.sum(axis=0)
    dxd = dz0 @ p['dense0_k'].T
    dx = np.where(keep, dxd / dt.type(1.0 - rate), dt.type(0)) if keep is not None else dxd
    E = c['E']
    dE = np.zeros_like(E)
    dv = dx[:, :n_v]
    g['conv_v_k'] = np.einsum('btc,bf->tcf', E, dv)
    g['conv_v_b'] = dv.sum(axis=0)
    dE += np.einsum('bf,tcf->btc', dv, p['conv_v_k'])
    for i in range(L):
        do = dx[:, n_v + i * n_h:n_v + (i + 1) * n_h]                  # [B, n_h]
        a = c['arg'][i]                                                # [B, n_h] argmax time step
        pre = np.take_along_axis(c['pre'][i], a[:, None, :], axis=1)[:, 0]
        dc = do * ACT[act_h][1](pre)                                   # through act_h at the arg-max position
        k = p[f'conv_h{i}_k']
        g[f'conv_h{i}_b'] = dc.sum(axis=0)
        for b in range(B):
            for f in range(n_h):
                if dc[b, f] != 0:
                    t = a[b, f]
                    g[f'conv_h{i}_k'][:, :, f] += E[b, t:t + i + 1] * dc[b, f]
                    dE[b, t:t + i + 1] += k[:, :, f] * dc[b, f]
    np.add.at(g['item_emb'], before, dE)
    reg = dt.type(0)
    for name in p:
        if REGULARISED(name):
            g[name] = g[name] + dt.type(2.0 * reg_rate) * p[name]
            reg = reg + dt.type(reg_rate) * (p[name] * p[name]).sum()
    return lval + reg, g, pred


def adam_state(p):
    return {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in p.items()}


def step(p, state, step_idx, uids, before, after, T, lr, reg_rate, keep=None, rate=0.0, act_h='relu', act_mlp='relu'):
    dt = p['item_emb'].dtype
    L = before.shape[1]
    lval, g, _ = loss_and_grads(p, uids, before, after, T, reg_rate, keep, rate, act_h, act_mlp)
    order = layer_order(L)
    for j, (_, names) in enumerate(order):
        a = dt.type(co.adam_alpha(lr, len(order) * step_idx + j + 1))
        for name in names:
            m, v = state[name]
            m[...] = m + (g[name] - m) * dt.type(co.ADAM_OMB1)
            v[...] = v + (g[name] * g[name] - v) * dt.type(co.ADAM_OMB2)
            p[name][...] = p[name] - (m * a) / (np.sqrt(v) + dt.type(co.ADAM_EPS))
    return lval


def rank_scores(p, uid, last_items):
    """Caser._rank (caser.py:128-146): scores of ALL items for one user from the last L items (no dropout)."""
    n_items = p['item_emb'].shape[0]
    s, _ = forward(p, np.array([uid]), np.asarray(last_items)[None, :], np.arange(n_items)[None, :])
    return s[0]
