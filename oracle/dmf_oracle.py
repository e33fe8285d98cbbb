"""NumPy restatement of the reference DMF step (TEST INFRASTRUCTURE, see oracle/__init__.py — **parity unpinned**:
TensorFlow/Keras arithmetic; cross-checked against torch-CPU autograd in tests/test_oracle_models.py).

Follows DRecPy/Recommender/dmf.py (paths under /root/reference):
  inputs ............ dmf.py:75-86   user row / item column of the interaction matrix with RAW values, l2-normalised
                      (tf.nn.l2_normalize: x * rsqrt(max(sum x^2, 1e-12)))
  towers ............ dmf.py:46-58,89-90  Keras Sequential of Dense(relu); glorot_uniform kernels, zero biases,
                      kernel_regularizer l2(reg) = reg * sum(w^2) (no 1/B, biases unregularised)
  score ............. dmf.py:92-95   cosine of the two representations, tf.maximum(1e-6, .)
  target ............ dmf.py:69      (r - min)/(max - min) when use_nce (min forced to 0 when the data minimum is 1,
                      recommender_abc.py:141,463-465) else the raw value
  loss .............. dmf.py:98-99   Keras BCE, [B] vs [B], mean
  optimizer ......... one apply_gradients per registered model (recommender_abc.py:328-334): user_nn's weights use Adam
                      t = 2*step + 1, item_nn's t = 2*step + 2

ModifiedDMF (examples/extending_recommender_dmf.py:5-18, BASELINE config 3) — present when p holds 'extra_w':
  extra weight ...... :11-12  tf.Variable([1.]) registered AFTER DMF._pre_fit; a tf.Variable lands in trainable_weights
                      (recommender_abc.py:274-275), which precede the models in the apply order (:194-196): Adam t =
                      3*step + 1 for the scalar, 3*step + 2 for user_nn, 3*step + 3 for item_nn; no regulariser on it
  predictions ....... :14-17  [extra_w * pred for pred in predictions]: a LIST of B (1,)-tensors -> Keras converts it to (B,1),
                      the (B,) targets broadcast against it to (B,B) (lists skip the squeeze step):
                      L = 1/B^2 sum_i sum_j bce(y_j, w*pred_i), computed literally here (`broadcast_targets=True`)
"""
import numpy as np

from . import cdae_oracle as co

L2N_EPS = 1e-12


def init_params(rng, n_users, n_items, user_factors=(64, 32), item_factors=(64, 32), dtype=np.float32):
    p = {}
    for tower, n_in, factors in (('u', n_items, user_factors), ('i', n_users, item_factors)):
        prev = n_in
        for l, f in enumerate(factors):
            p[f'{tower}{l}_k'] = co.glorot_uniform(rng, (prev, f), dtype)
            p[f'{tower}{l}_b'] = np.zeros(f, dtype)
            prev = f
    return p


def l2_normalize(x):
    q = (x * x).sum(axis=1, keepdims=True)
    rho = 1.0 / np.sqrt(np.maximum(q, x.dtype.type(L2N_EPS)))
    return x * rho, q, rho


def l2_normalize_bwd(dn, n, q, rho):
    inside = q > L2N_EPS
    return np.where(inside, rho * (dn - n * (n * dn).sum(axis=1, keepdims=True)), rho * dn)


def tower_fwd(p, tower, x, n_layers):
    acts = [x]
    pres = []
    for l in range(n_layers):
        z = acts[-1] @ p[f'{tower}{l}_k'] + p[f'{tower}{l}_b']
        pres.append(z)
        acts.append(np.maximum(z, 0))
    return acts, pres


def forward(p, xu, xi, nu_layers, ni_layers, l2_norm_vectors=True):
    """xu [B,N] raw user rows, xi [B,U] raw item columns -> predictions [B] and the cache."""
    c = {}
    if l2_norm_vectors:
        xu_n, c['qu0'], c['rhou0'] = l2_normalize(xu)
        xi_n, c['qi0'], c['rhoi0'] = l2_normalize(xi)
    else:
        xu_n, xi_n = xu, xi
    c['au'], c['pu'] = tower_fwd(p, 'u', xu_n, nu_layers)
    c['ai'], c['pi'] = tower_fwd(p, 'i', xi_n, ni_layers)
    c['nu'], c['qu'], c['rhou'] = l2_normalize(c['au'][-1])
    c['ni'], c['qi'], c['rhoi'] = l2_normalize(c['ai'][-1])
    c['s'] = (c['nu'] * c['ni']).sum(axis=1)
    c['cos'] = np.maximum(xu.dtype.type(1e-6), c['s'])
    pred = c['cos'] * p['extra_w'][0] if 'extra_w' in p else c['cos']
    return pred, c


def loss_and_grads(p, xu, xi, y, reg_rate, nu_layers, ni_layers, l2_norm_vectors=True, broadcast_targets=False):
    dt = xu.dtype
    B = len(y)
    pred, c = forward(p, xu, xi, nu_layers, ni_layers, l2_norm_vectors)
    y = np.asarray(y, dt)
    if broadcast_targets:                       # (B,) targets vs (B,1) predictions -> (B,B); mean over both axes
        lval = co.bce_elem(y[None, :], pred[:, None], dt).mean(axis=-1).mean()
        dpred = co.bce_grad(y[None, :], pred[:, None], dt).sum(axis=1) / dt.type(B * B)
    else:
        lval = co.bce_elem(y, pred, dt).mean()
        dpred = co.bce_grad(y, pred, dt) / dt.type(B)
    g = {}
    if 'extra_w' in p:
        g['extra_w'] = np.array([(dpred * c['cos']).sum()], dt)
        dpred = dpred * p['extra_w'][0]
    ds = np.where(c['s'] > 1e-6, dpred, dt.type(0))[:, None]
    for tower, n_layers, dn, key in (('u', nu_layers, ds * c['ni'], 'u'), ('i', ni_layers, ds * c['nu'], 'i')):
        acts, pres = (c['au'], c['pu']) if tower == 'u' else (c['ai'], c['pi'])
        n, q, rho = (c['nu'], c['qu'], c['rhou']) if tower == 'u' else (c['ni'], c['qi'], c['rhoi'])
        da = l2_normalize_bwd(dn, n, q, rho)
        for l in reversed(range(n_layers)):
            dz = da * (pres[l] > 0)
            g[f'{tower}{l}_k'] = acts[l].T @ dz + dt.type(2.0 * reg_rate) * p[f'{tower}{l}_k']
            g[f'{tower}{l}_b'] = dz.sum(axis=0)
            da = dz @ p[f'{tower}{l}_k'].T
    reg = sum(dt.type(reg_rate) * (v * v).sum() for k, v in p.items() if k.endswith('_k'))
    return lval + reg, g, pred


def adam_state(p):
    return {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in p.items()}


def step(p, state, step_idx, xu, xi, y, lr, reg_rate, nu_layers, ni_layers, l2_norm_vectors=True, broadcast_targets=False):
    dt = xu.dtype
    lval, g, _ = loss_and_grads(p, xu, xi, y, reg_rate, nu_layers, ni_layers, l2_norm_vectors, broadcast_targets)
    groups = ([['extra_w']] if 'extra_w' in p else []) + [[k for k in p if k[0] == tower and k[1].isdigit()] for tower in ('u', 'i')]
    for j, names in enumerate(groups):          # one apply_gradients per registered item, trainable_weights first
        a = dt.type(co.adam_alpha(lr, len(groups) * step_idx + j + 1))
        for name in names:
            m, v = state[name]
            m[...] = m + (g[name] - m) * dt.type(co.ADAM_OMB1)
            v[...] = v + (g[name] * g[name] - v) * dt.type(co.ADAM_OMB2)
            p[name][...] = p[name] - (m * a) / (np.sqrt(v) + dt.type(co.ADAM_EPS))
    return lval
