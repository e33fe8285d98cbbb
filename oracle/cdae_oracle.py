"""NumPy restatement of the reference CDAE training/inference arithmetic (TEST INFRASTRUCTURE, see
oracle/__init__.py — **parity unpinned** for this file: TensorFlow is absent).

Every function cites the reference lines it follows (paths under /root/reference) and the TF-2.x /
Keras numerics of SURVEY.md App. A.  `dtype` selects float32 (what TF computes in) or float64
(the arbiter for the 1e-5 gate).

Reference ("dense") mode, one fit() iteration = one mini-batch (recommender_abc.py:186-205):
    t_bn   = 1[r_{u_b,n} >= thr]                                   cdae.py:61
    x~_bn  = keep_bn * t_bn / (1-q)                                cdae.py:63
    h_b    = sigmoid(x~_b W + V[u_b] + b)                          cdae.py:74-75
    p_b    = sigmoid(h_b W_ + b_)                                  cdae.py:76
    L_pred = Keras BCE / MSE of y_true (B,N) vs y_pred (B,1,N) -> (B,B,N) broadcast, mean
             == loss of every prediction row against the batch-mean target   cdae.py:78-79, App. A.3
    L_reg  = reg/B * 1/2 (|W|^2 + |W_|^2 + |V|^2)                  cdae.py:81-82
    Adam   : one apply_gradients per variable -> t = 5*step + j + 1, j in W,W_,V,b,b_ order
                                                                   recommender_abc.py:328-334, cdae.py:43

Sampled ("sparse") mode is the engine's documented deviation: one output unit
per (u,i,y) triple, sparse Adagrad/Adam on touched rows.  Its definition lives here so that the
HIP path can be checked against a CPU statement of exactly that mode.
"""
import numpy as np

KERAS_EPS = 1e-7           # tf.keras.backend.epsilon()
# tf.keras.optimizers.Adam defaults (App. A.5).  TF casts the hyper-parameters to the variable dtype (fp32) and
# its ApplyAdam kernel forms (1 - beta) IN fp32:  m += (g - m) * (1 - b1);  v += (g*g - v) * (1 - b2);
# var -= (m * lr_t) / (sqrt(v) + eps).  1.0f - 0.999f = 0.00099998713 (not 0.001): the constants below are those
# fp32 values, used as-is by both the float32 and the float64 restatement.
ADAM_B1, ADAM_B2, ADAM_EPS = float(np.float32(0.9)), float(np.float32(0.999)), float(np.float32(1e-7))
ADAM_OMB1 = float(np.float32(1.0) - np.float32(0.9))
ADAM_OMB2 = float(np.float32(1.0) - np.float32(0.999))
ADAGRAD_INIT, ADAGRAD_EPS = float(np.float32(0.1)), float(np.float32(1e-7))   # tf.keras.optimizers.Adagrad defaults


def glorot_uniform(rng, shape, dtype=np.float32):
    """tf.initializers.GlorotUniform (cdae.py:35-41): U(-l, l), l = sqrt(6/(fan_in+fan_out));
    1-D shape [n] -> fan_in = fan_out = n (App. A.1).  `rng` is a numpy Generator (TF's RNG is not
    reproducible outside TF, so weights are always injected)."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        fan_in, fan_out = shape[0], shape[1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def glorot_uniform_conv(rng, shape, dtype=np.float32):
    """Keras glorot_uniform for a Conv1D kernel [k, in, out]: fan_in = k*in, fan_out = k*out."""
    k, cin, cout = shape
    lim = np.sqrt(6.0 / (k * cin + k * cout))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def init_params(rng, n_users, n_items, k, dtype=np.float32):
    """cdae.py:34-41.  W_ is kept in the reference's [K,N] orientation here."""
    return {
        'W': glorot_uniform(rng, (n_items, k), dtype),
        'W_': glorot_uniform(rng, (k, n_items), dtype),
        'V': glorot_uniform(rng, (n_users, k), dtype),
        'b': glorot_uniform(rng, (k,), dtype),
        'b_': glorot_uniform(rng, (n_items,), dtype),
    }


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def forward(params, uids, x_tilde):
    """cdae.py:73-76 for a stack of rows.  x_tilde [B,N] (already corrupted/scaled, or the plain
    binary vector for inference, cdae.py:67-71)."""
    z1 = x_tilde @ params['W'] + params['V'][uids] + params['b']
    h = sigmoid(z1)
    p = sigmoid(h @ params['W_'] + params['b_'])
    return h, p


def predict_row(params, uid, t_vec):
    """CDAE._predict / _reconstruct_for_predictions (cdae.py:67-71, 84-88): uncorrupted, unscaled."""
    dt = params['W'].dtype
    _, p = forward(params, np.array([uid]), np.asarray(t_vec, dtype=dt)[None, :])
    return p[0]


def bce_elem(t, p, dt):
    """tf.keras.backend.binary_crossentropy, from_logits=False (App. A.3)."""
    eps = dt.type(KERAS_EPS)
    one = dt.type(1.0)
    pc = np.clip(p, eps, one - eps)
    return -(t * np.log(pc + eps) + (one - t) * np.log(one - pc + eps))


def bce_grad(t, p, dt):
    """d bce_elem / d p with TF's clip_by_value gradient (passes where eps <= p <= 1-eps)."""
    eps = dt.type(KERAS_EPS)
    one = dt.type(1.0)
    pc = np.clip(p, eps, one - eps)
    g = -(t / (pc + eps) - (one - t) / (one - pc + eps))
    inside = (p >= eps) & (p <= one - eps)
    return np.where(inside, g, dt.type(0.0))


def batch_loss(p, t, loss='bce', targets='reference'):
    """cdae.py:78-79.  targets='reference': the (B,B,N) broadcast, computed literally."""
    dt = p.dtype
    B, N = p.shape
    if targets == 'reference':
        tt = t.astype(dt)[:, None, :]             # y_true (B,N)   -> broadcast axis j
        pp = p[None, :, :]                         # y_pred (B,1,N) -> broadcast axis i
        # Keras: mean over last axis, then mean over remaining (B,B) grid
        if loss == 'bce':
            e = bce_elem(tt, pp, dt)
        else:
            e = (pp - tt) ** 2
        return e.mean(axis=-1).mean()
    tt = t.astype(dt)
    e = bce_elem(tt, p, dt) if loss == 'bce' else (p - tt) ** 2
    return e.mean(axis=-1).mean()


def dense_grads(params, uids, x_tilde, t, reg_rate, loss='bce', targets='reference'):
    """Loss and dense gradients of  L_pred + L_reg  wrt (W, W_, V, b, b_)."""
    dt = params['W'].dtype
    B, N = x_tilde.shape
    W, W_, V, b, b_ = (params[k] for k in ('W', 'W_', 'V', 'b', 'b_'))
    h, p = forward(params, uids, x_tilde)
    tt = t.astype(dt)
    tbar = tt.mean(axis=0, keepdims=True) if targets == 'reference' else tt
    if loss == 'bce':
        # bce is affine in t, so the (B,B,N) mean equals the loss against the batch-mean target
        lval = bce_elem(np.broadcast_to(tbar, p.shape), p, dt).mean(axis=-1).mean()
        dp = bce_grad(np.broadcast_to(tbar, p.shape), p, dt) / dt.type(B * N)
    else:
        lval = batch_loss(p, t, 'mse', targets)
        dp = dt.type(2.0) * (p - tbar) / dt.type(B * N)
    dz2 = dp * p * (1 - p)
    rb = dt.type(reg_rate) / dt.type(B)
    g = {}
    g['W_'] = h.T @ dz2 + rb * W_
    g['b_'] = dz2.sum(axis=0)
    dh = dz2 @ W_.T
    dz1 = dh * h * (1 - h)
    g['b'] = dz1.sum(axis=0)
    gV = np.zeros_like(V)
    np.add.at(gV, uids, dz1)
    g['V'] = gV + rb * V
    g['W'] = x_tilde.T @ dz1 + rb * W
    reg = rb * dt.type(0.5) * ((W * W).sum() + (W_ * W_).sum() + (V * V).sum())
    return lval + reg, g, (h, p)


def adam_state(params):
    return {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in params.items()}


VAR_ORDER = ('W', 'W_', 'V', 'b', 'b_')      # registration order, cdae.py:43


def adam_alpha(lr, t):
    """lr_t of Keras Adam for the 1-based step t (App. A.5), evaluated in fp32 like optimizer_v2/adam.py does
    (beta powers via pow on the fp32-cast hyper-parameters)."""
    f = np.float32
    b1p = np.power(f(0.9), f(t))
    b2p = np.power(f(0.999), f(t))
    return float(f(lr) * np.sqrt(f(1.0) - b2p) / (f(1.0) - b1p))


def dense_step(params, state, step, uids, x_tilde, t, lr, reg_rate, loss='bce', targets='reference'):
    """One fit() iteration in reference mode; updates params/state in place; returns the loss.
    `step` is the 0-based batch index: variable j uses Adam t = 5*step + j + 1 (App. A.5)."""
    dt = params['W'].dtype
    lval, g, _ = dense_grads(params, uids, x_tilde, t, reg_rate, loss, targets)
    for j, name in enumerate(VAR_ORDER):
        tt = 5 * step + j + 1
        a = dt.type(adam_alpha(lr, tt))
        m, v = state[name]
        m[...] = m + (g[name] - m) * dt.type(ADAM_OMB1)
        v[...] = v + (g[name] * g[name] - v) * dt.type(ADAM_OMB2)
        params[name][...] = params[name] - (m * a) / (np.sqrt(v) + dt.type(ADAM_EPS))
    return lval


# ----------------------------------------------------------------------------------------------
# sampled-output / sparse-optimizer mode (engine deviation; definition of record)
# ----------------------------------------------------------------------------------------------
def sparse_state(params, optimizer='adagrad'):
    if optimizer == 'adagrad':
        return {k: np.full_like(v, ADAGRAD_INIT) for k, v in params.items()}
    if optimizer == 'rowwise_adagrad':
        # one accumulator per table ROW (W rows, W_ columns, V rows); the bias vectors keep one per element
        return {'W': np.full(params['W'].shape[0], ADAGRAD_INIT, params['W'].dtype),
                'W_': np.full(params['W_'].shape[1], ADAGRAD_INIT, params['W_'].dtype),
                'V': np.full(params['V'].shape[0], ADAGRAD_INIT, params['V'].dtype),
                'b': np.full_like(params['b'], ADAGRAD_INIT), 'b_': np.full_like(params['b_'], ADAGRAD_INIT)}
    return {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in params.items()}


def sparse_step(params, state, step, uids, iids, y, kept, q, lr, reg_rate, loss='bce',
                optimizer='adagrad', dot_reduce=None, accumulate='loop'):
    """One step over B triples (u_b, i_b, y_b).  kept[b] = item ids of u_b's positives that
    survive corruption (cdae.py:61-63 restricted to the non-zero entries).

        h_b  = sigmoid(1/(1-q) * sum_{n in kept_b} W[n] + V[u_b] + b)
        p_b  = sigmoid(h_b . W_[:, i_b] + b_[i_b])
        L    = 1/B sum_b l(y_b, p_b)                     (l = Keras BCE with eps, or squared error)
        g_row (W_, V, W rows touched by the batch) = data gradient + reg/B * row   (once per step)
        b: dense gradient; b_: touched entries only
        Adagrad (Keras: acc0 = .1, eps = 1e-7): acc += g^2 ; p -= lr * g / (sqrt(acc) + eps)
        lazy Adam: m, v, p of touched rows only, global step t = step + 1, Keras lr_t.

    accumulate: how the row gradients are summed over the batch.  'loop' (the definition): triple by triple, a Python loop — seconds
    per thousand triples.  'matrix': the same sums as sparse matrix products (scipy: indicator^T @ per-triple gradients), for batches
    the loop cannot follow (65 536 triples over MovieLens-length histories: 8.6 M row additions); another association of the same
    fp64 sums — tests/test_oracle_cdae.py holds the two to 1e-12 of each other.
    """
    dt = params['W'].dtype
    W, W_, V, b, b_ = (params[k] for k in ('W', 'W_', 'V', 'b', 'b_'))
    B = len(uids)
    s = dt.type(1.0 / (1.0 - q))
    z1 = np.zeros((B, W.shape[1]), dtype=dt)
    if accumulate == 'matrix':
        import scipy.sparse as sp
        lens = np.fromiter((len(k_) for k_ in kept), dtype=np.int64, count=B)
        ip = np.zeros(B + 1, dtype=np.int64)
        ip[1:] = np.cumsum(lens)
        cols = np.concatenate([np.asarray(k_, dtype=np.int64) for k_ in kept]) if ip[-1] else np.zeros(0, np.int64)
        A = sp.csr_matrix((np.ones(len(cols), dtype=dt), cols, ip), shape=(B, W.shape[0]))        # [B, N]: triple b keeps item n
        z1 = np.asarray(A @ W, dtype=dt) * s
    else:
        for bi in range(B):
            if len(kept[bi]):
                z1[bi] = W[np.asarray(kept[bi])].sum(axis=0, dtype=dt) * s
    z1 = z1 + V[uids] + b
    h = sigmoid(z1)
    w2 = W_[:, iids].T                                   # [B,K]
    d = (h * w2).sum(axis=1)
    if dot_reduce is not None:       # column-sharded layout: `params` holds a column slice, the dot products are summed over ranks
        d = dot_reduce(d)
    p = sigmoid(d + b_[iids])
    yy = np.asarray(y, dtype=dt)
    if loss == 'bce':
        lval = bce_elem(yy, p, dt).mean()
        dp = bce_grad(yy, p, dt) / dt.type(B)
    else:
        lval = ((p - yy) ** 2).mean()
        dp = dt.type(2.0) * (p - yy) / dt.type(B)
    dz2 = dp * p * (1 - p)
    dh = dz2[:, None] * w2
    dz1 = dh * h * (1 - h)
    rb = dt.type(reg_rate) / dt.type(B)

    if accumulate == 'matrix':
        def by_key(keys, rows, n_keys):
            """{key: sum of rows[b] over the triples b with keys[b] == key} as one indicator^T @ rows product"""
            keys = np.asarray(keys, dtype=np.int64)
            Ik = sp.csr_matrix((np.ones(B, dtype=dt), (keys, np.arange(B))), shape=(n_keys, B))
            G = np.asarray(Ik @ rows, dtype=dt)
            return {int(k_): G[k_] for k_ in np.unique(keys)}
        gW_ = by_key(iids, dz2[:, None] * h, W_.shape[1])
        gb_ = {k_: v_[0] for k_, v_ in by_key(iids, dz2[:, None], W_.shape[1]).items()}
        gV = by_key(uids, dz1, V.shape[0])
        GW = np.asarray(A.T @ (dz1 * s), dtype=dt)
        gW = {int(n): GW[n] for n in np.unique(cols)}
    else:
        gW_ = {}
        gb_ = {}
        for bi in range(B):
            i = int(iids[bi])
            gW_[i] = gW_.get(i, 0) + dz2[bi] * h[bi]
            gb_[i] = gb_.get(i, 0) + dz2[bi]
        gV = {}
        for bi in range(B):
            u = int(uids[bi])
            gV[u] = gV.get(u, 0) + dz1[bi]
        gW = {}
        for bi in range(B):
            for n in kept[bi]:
                n = int(n)
                gW[n] = gW.get(n, 0) + dz1[bi] * s
    gb = dz1.sum(axis=0)

    def upd(name, index, g):
        """index: tuple selecting the touched slice of params[name]."""
        g = np.asarray(g, dtype=dt)
        if optimizer == 'rowwise_adagrad' and name in ('W', 'W_', 'V'):
            # engine extension (no reference counterpart): acc_row += mean_k(g^2); p -= lr * g / (sqrt(acc_row) + eps)
            r = index[1] if name == 'W_' else index[0]
            acc = state[name]
            acc[r] = acc[r] + (g * g).mean(dtype=dt)
            params[name][index] = params[name][index] - dt.type(np.float32(lr)) * g / (np.sqrt(acc[r]) + dt.type(ADAGRAD_EPS))
            return
        if optimizer in ('adagrad', 'rowwise_adagrad'):
            acc = state[name]
            acc[index] = acc[index] + g * g
            params[name][index] = params[name][index] - dt.type(np.float32(lr)) * g / (np.sqrt(acc[index]) + dt.type(ADAGRAD_EPS))
        else:
            m, v = state[name]
            a = dt.type(adam_alpha(lr, step + 1))
            m[index] = m[index] + (g - m[index]) * dt.type(ADAM_OMB1)
            v[index] = v[index] + (g * g - v[index]) * dt.type(ADAM_OMB2)
            params[name][index] = params[name][index] - (m[index] * a) / (np.sqrt(v[index]) + dt.type(ADAM_EPS))

    # gradients are all taken at the pre-update parameters
    regW_ = {i: rb * W_[:, i].copy() for i in gW_}
    regV = {u: rb * V[u].copy() for u in gV}
    regW = {n: rb * W[n].copy() for n in gW}
    for i in gW_:
        upd('W_', (slice(None), i), gW_[i] + regW_[i])
        upd('b_', (i,), gb_[i])
    for u in gV:
        upd('V', (u,), gV[u] + regV[u])
    for n in gW:
        upd('W', (n,), gW[n] + regW[n])
    upd('b', (slice(None),), gb)
    return lval, (h, p)


def drx_hash_u32(seed, a, b):
    """numpy restatement of drx_hash_u32 (include/drx.h): the counter-based corruption mask of the throughput mode.
    Entry j of batch row b survives iff drx_hash_u32(mask_seed, b, j) >= floor(float32(q) * 2^32)."""
    M = (1 << 64) - 1
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over='ignore'):
        x = np.uint64(seed & M) + a * np.uint64(0x9E3779B97F4A7C15) + b * np.uint64(0xD1B54A32D192ED03)
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(32)).astype(np.uint32)


def q_threshold(q):
    t = float(np.float32(q)) * 4294967296.0
    return 0 if t <= 0 else min(int(t), 0xFFFFFFFF)


# ----------------------------------------------------------------------------------------------
# ranking (cdae.py:90-103): heapq.nlargest over (prediction, iid) => ties broken by larger iid
# ----------------------------------------------------------------------------------------------
def rank_row(pred, candidate_iids, n, exclude=()):
    from heapq import nlargest
    ex = set(int(i) for i in exclude)
    cand = set(int(i) for i in candidate_iids) - ex
    lst = [(float(pred[i]), i) for i in range(len(pred)) if i in cand]
    return nlargest(n, lst)
