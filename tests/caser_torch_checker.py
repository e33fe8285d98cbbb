"""TEST INFRASTRUCTURE — a second checker for Caser: the forward of caser.py:97-120 written with torch operations and differentiated by
torch.autograd (where the reference has tf.GradientTape), the update the library's dense Keras-Adam kernel (drx_adam_dense: one apply
per registered layer, t = (6 + L) * step + j + 1, l2 on embeddings and kernels).  Through round 5 this file was a product backend
(drecpy_amd/engine_caser_wide.py) for constructor arguments outside the fused HIP kernel's domain; the product now REJECTS those
arguments (Recommender/caser.py) and nothing under drecpy_amd/ imports torch.autograd for a built-in model.  It stays here because it
is an independent statement of the same arithmetic: tests hold it against oracle/caser_oracle.py and hold the HIP kernels against it.
Same interface as drecpy_amd.engine_caser.CaserEngine."""
import numpy as np
import torch

from drecpy_amd import _lib
from drecpy_amd._lib import check, lib, stream_ptr
from drecpy_amd.engine import ADAM_B1, ADAM_B2, ADAM_EPS


def _act(kind):
    return {'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid, 'linear': (lambda v: v), None: (lambda v: v)}[kind]


class CaserTorchChecker:
    table_update = 'dense'

    def __init__(self, n_users, n_items, L=5, T=3, neg_ratio=3, d=50, n_v=4, n_h=16, device='cuda:0', act_h='relu', act_mlp='relu'):
        if not torch.cuda.is_available():
            raise _lib.DrxError('drecpy_amd needs a ROCm GPU (MI355X); there is no CPU fallback.')
        lib()
        if act_h not in ('relu', 'tanh', 'sigmoid', 'linear', None) or act_mlp not in ('relu', 'tanh', 'sigmoid', 'linear', None):
            raise _lib.DrxError(f'Caser engine: activations relu / tanh / sigmoid / linear are supported, got act_h={act_h!r}, act_mlp={act_mlp!r}')
        self.device = torch.device(device)
        self.U, self.N, self.L, self.T, self.d, self.n_v, self.n_h = n_users, n_items, L, T, d, n_v, n_h
        self.Tp = T + T * neg_ratio
        self.nx = n_v + L * n_h
        self.act_h, self.act_mlp = _act(act_h), _act(act_mlp)
        z = dict(dtype=torch.float32, device=self.device)
        # the reference's own shapes (oracle/caser_oracle.py:init_params)
        self.p = {'user_emb': torch.zeros(n_users, d, **z), 'item_emb': torch.zeros(n_items, d, **z),
                  'conv_v_k': torch.zeros(L, d, n_v, **z), 'conv_v_b': torch.zeros(n_v, **z)}
        for i in range(L):
            self.p[f'conv_h{i}_k'] = torch.zeros(i + 1, d, n_h, **z)
            self.p[f'conv_h{i}_b'] = torch.zeros(n_h, **z)
        self.p.update({'dense0_k': torch.zeros(self.nx, d, **z), 'dense0_b': torch.zeros(d, **z),
                       'W1': torch.zeros(n_items, 2 * d, **z), 'b1': torch.zeros(n_items, 1, **z)})
        self.state = {n: (torch.zeros_like(t), torch.zeros_like(t)) for n, t in self.p.items()}
        # registration order of caser.py:47-70 = order of the per-step Adam applies
        self.order = [['user_emb'], ['item_emb'], ['conv_v_k', 'conv_v_b']] + [[f'conv_h{i}_k', f'conv_h{i}_b'] for i in range(L)] + \
                     [['dense0_k', 'dense0_b'], ['W1'], ['b1']]
        self.n_layers = 6 + L
        self.lr, self.reg = 1e-3, 1e-3
        self.beta1, self.beta2, self.eps = ADAM_B1, ADAM_B2, ADAM_EPS
        from drecpy_amd.Recommender.trainables import TrainableLayer
        names = ['user_embeddings', 'item_embeddings', 'conv_v'] + [f'convs_h[{i}]' for i in range(L)] + ['dense_0', 'dense_1_W', 'dense_1_b']
        self.layers = [TrainableLayer(nm, (lambda ks=ks: [self.p[k] for k in ks])) for nm, ks in zip(names, self.order)]

    @staticmethod
    def regularised(name):
        return name.endswith('_k') or name in ('user_emb', 'item_emb', 'W1')

    def tensors(self):
        return self.p

    def set_params(self, p):
        for k, t in self.p.items():
            t.copy_(torch.as_tensor(np.asarray(p[k], dtype=np.float32)).reshape(t.shape).to(self.device))

    def get_params(self):
        return {k: t.detach().cpu().numpy().copy() for k, t in self.p.items()}

    def snapshot(self):
        return {'p': {n: t.clone() for n, t in self.p.items()}}

    def restore(self, snap, with_optimizer=False):
        for n, t in self.p.items():
            t.copy_(snap['p'][n])

    def prepare_batch(self, uids, before, after):
        return {'uids': np.asarray(uids), 'before': np.asarray(before), 'after': np.asarray(after), 'B': len(uids)}

    def _idx(self, a):
        return a.to(self.device, torch.long) if torch.is_tensor(a) else torch.as_tensor(np.asarray(a, dtype=np.int64)).to(self.device)

    def _alphas(self, step_idx):
        """Keras-Adam lr_t of the n_layers applies of step `step_idx` (t = n_layers * step + j + 1), fp32 like optimizer_v2/adam.py."""
        f = np.float32
        t = (self.n_layers * step_idx + 1 + np.arange(self.n_layers)).astype(np.float32)
        return (f(self.lr) * np.sqrt(f(1.0) - np.power(f(self.beta2), t)) / (f(1.0) - np.power(f(self.beta1), t))).astype(np.float32).tolist()

    def _hidden(self, P, uid, bef, keep=None, rate=0.0):
        """concat(dense_0 output, user embedding) [B, 2d] (caser.py:97-118)"""
        L = self.L
        E = P['item_emb'][bef]                                            # [B, L, d]
        out = [torch.einsum('btc,tcf->bf', E, P['conv_v_k']) + P['conv_v_b']]
        for i in range(L):
            k = P[f'conv_h{i}_k']                                         # [i + 1, d, n_h]
            c = torch.stack([torch.einsum('bsc,scf->bf', E[:, t:t + i + 1], k) for t in range(L - i)], dim=1) + P[f'conv_h{i}_b']
            out.append(self.act_h(c).max(dim=1).values)                   # max over time (the first maximum takes the gradient)
        x = torch.cat(out, dim=1)
        if keep is not None:
            x = torch.where(keep, x / (1.0 - rate), torch.zeros_like(x))
        z = self.act_mlp(x @ P['dense0_k'] + P['dense0_b'])
        return torch.cat([z, P['user_emb'][uid]], dim=1)

    def step(self, step_idx, uids, before=None, after=None, keep=None, rate=0.0, want_loss=False, mask_seed=0):
        if isinstance(uids, dict):
            uids, before, after = uids['uids'], uids['before'], uids['after']
        uid, bef, aft = self._idx(uids), self._idx(before), self._idx(after)
        B, Tp = aft.shape
        kp = None
        if keep is not None:
            kp = (keep if torch.is_tensor(keep) else torch.as_tensor(np.asarray(keep))).to(self.device).bool()
        elif rate > 0:
            # (the fused kernel evaluates a counter-based mask; here the mask of a generator seeded by the same (seed, step) key)
            gen = torch.Generator(device=self.device)
            gen.manual_seed(int(mask_seed) & ((1 << 63) - 1))
            kp = torch.rand(B, self.nx, generator=gen, device=self.device) >= rate
        P = {k: t.detach().requires_grad_(True) for k, t in self.p.items()}
        cat = self._hidden(P, uid, bef, kp, rate)
        scores = torch.einsum('bk,bjk->bj', cat, P['W1'][aft]) + P['b1'][aft][:, :, 0]
        pred = torch.sigmoid(scores)
        y = torch.zeros(B, Tp, device=self.device)
        y[:, :self.T] = 1.0
        eps = 1e-7
        pc = pred.clamp(eps, 1 - eps)
        loss = (-(y * torch.log(pc + eps) + (1 - y) * torch.log(1 - pc + eps))).mean(dim=-1).mean()
        names = list(P)
        grads = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
        g = {n: (gr if gr is not None else torch.zeros_like(self.p[n])).contiguous() for n, gr in zip(names, grads)}
        alpha = self._alphas(step_idx)
        l2c = 2.0 * self.reg
        reg_loss = 0.0
        if want_loss:
            reg_loss = float(sum(self.reg * float((self.p[n] * self.p[n]).sum().item()) for n in names if self.regularised(n)))
        st = stream_ptr(self.device)
        for j, layer in enumerate(self.order):
            for n in layer:
                pt, (m, v) = self.p[n], self.state[n]
                check(lib().drx_adam_dense(pt.data_ptr(), m.data_ptr(), v.data_ptr(), g[n].data_ptr(), pt.numel(), alpha[j],
                                           l2c if self.regularised(n) else 0.0, self.beta1, self.beta2, self.eps, st), 'drx_adam_dense')
        if want_loss:
            return float(loss.item()) + reg_loss
        return None

    def scores_all(self, uids, before):
        with torch.no_grad():
            cat = self._hidden(self.p, self._idx(uids), self._idx(before))
            return cat @ self.p['W1'].t() + self.p['b1'][:, 0]
