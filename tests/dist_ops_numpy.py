"""NumPy statement of the per-rank work of the row-sharded step (same interface as drecpy_amd.dist.HipShardOps), used by
the world-size-2 gloo tests to exercise drecpy_amd/dist.py's exchange logic on CPU.  TEST INFRASTRUCTURE: float64, loops."""
import types

import numpy as np
import torch

from oracle import cdae_oracle as co
from drecpy_amd.dist import item_key, items_per_rank

NONE = 0xFFFFFFFF


def np_batch(uid, iid, y, indptr, q, mask_seed=0, keep=None):
    uid = np.asarray(uid, np.int64)
    deg = indptr[uid + 1] - indptr[uid]
    keep_off = np.zeros(len(uid) + 1, np.int64)
    keep_off[1:] = np.cumsum(deg)
    return types.SimpleNamespace(B=len(uid), uid=uid, iid=np.asarray(iid, np.int64), y=np.asarray(y, np.float64),
                                 keep_off=keep_off, keep=keep, mask_seed=mask_seed, q=float(np.float32(q)),
                                 n_touch_slots=int(keep_off[-1]))


class NumpyShardOps:
    def __init__(self, n_users_local, n_items, k, rank, world, indptr, indices, lr, reg):
        self.rank, self.world, self.k = rank, world, k
        self.ipr = items_per_rank(n_items, world)
        self.n_items, self.n_users_local = n_items, n_users_local
        self.indptr, self.indices = np.asarray(indptr, np.int64), np.asarray(indices, np.int64)
        self.lr, self.reg = lr, reg
        self.engine = None
        self.uk0 = world * 2 * self.ipr

    def set_params(self, W, W_, V, b, b_):
        f = np.float64
        self.W, self.W2T, self.V, self.b, self.b2 = W.astype(f), W_.T.astype(f).copy(), V.astype(f), b.astype(f), b_.astype(f)
        self.acc = {n: np.full_like(getattr(self, n), co.ADAGRAD_INIT) for n in ('W', 'W2T', 'V', 'b', 'b2')}

    def get_params(self):
        return {'W': self.W, 'W_': self.W2T.T, 'V': self.V, 'b': self.b, 'b_': self.b2}

    def optim(self, step):
        return None

    def _kept(self, bt, b):
        u = bt.uid[b]
        s, e = self.indptr[u], self.indptr[u + 1]
        if bt.keep is not None:
            kf = bt.keep[bt.keep_off[b]:bt.keep_off[b + 1]].astype(bool)
        else:
            kf = co.drx_hash_u32(bt.mask_seed, np.full(e - s, b), np.arange(e - s)) >= co.q_threshold(bt.q)
        return self.indices[s:e], kf

    def touches(self, bt):
        T = bt.n_touch_slots + 2 * bt.B
        keys = np.full(T, NONE, np.int64)
        bpos = np.zeros(T, np.int64)
        for b in range(bt.B):
            base = bt.keep_off[b] + 2 * b
            items, kf = self._kept(bt, b)
            for jj, (n, k) in enumerate(zip(items, kf)):
                if k:
                    keys[base + jj] = item_key(int(n), self.ipr, False)
                bpos[base + jj] = b
            d = len(items)
            keys[base + d] = item_key(int(bt.iid[b]), self.ipr, True)
            keys[base + d + 1] = self.uk0 + bt.uid[b]
            bpos[base + d] = bpos[base + d + 1] = b
        self._bt = bt
        return torch.from_numpy(keys), torch.arange(T), torch.from_numpy(bpos)

    def index(self, keys, vals):
        k = keys.numpy()
        order = np.argsort(k, kind='stable')
        ks, vs = k[order], vals.numpy()[order]
        real = ks != NONE
        uniq, inv = np.unique(ks[real], return_inverse=True)
        slot_sorted = np.full(len(ks), -1, np.int64)
        slot_sorted[real] = inv
        slot_of_pos = np.full(len(ks), NONE, np.int64)
        slot_of_pos[vs[real]] = inv
        bounds = [int(np.searchsorted(uniq, o * 2 * self.ipr)) for o in range(self.world + 1)] + [len(uniq)]
        return {'keys_s': torch.from_numpy(ks), 'vals_s': torch.from_numpy(vs), 'slot_sorted': torch.from_numpy(slot_sorted),
                'slot_of_pos': torch.from_numpy(slot_of_pos), 'uniq_keys': torch.from_numpy(uniq), 'bounds': bounds}

    def _local(self, key):
        t = key - self.rank * 2 * self.ipr
        return (t >= self.ipr), (t - self.ipr if t >= self.ipr else t)

    def gather_rows(self, req):
        r = req.numpy()
        rows = np.zeros((len(r), self.k))
        b2v = np.zeros(len(r))
        for i, key in enumerate(r):
            is_out, row = self._local(int(key))
            rows[i] = self.W2T[row] if is_out else self.W[row]
            b2v[i] = self.b2[row] if is_out else 0.0
        return torch.from_numpy(rows), torch.from_numpy(b2v)

    def fwd_bwd(self, bt, slot_of_pos, rows_cache, b2_cache, b_norm, loss_kind):
        sp, rc, bc = slot_of_pos.numpy(), rows_cache.numpy(), b2_cache.numpy()
        B, K = bt.B, self.k
        dt = np.dtype(np.float64)
        ctx = {'dz1': np.zeros((B, K)), 'g2': np.zeros((B, K)), 'dz2': np.zeros(B), 'lossb': np.zeros(B)}
        s = 1.0 / (1.0 - bt.q)
        for b in range(B):
            base = bt.keep_off[b] + 2 * b
            d = bt.keep_off[b + 1] - bt.keep_off[b]
            acc = np.zeros(K)
            for jj in range(d):
                if sp[base + jj] != NONE:
                    acc += rc[sp[base + jj]]
            h = co.sigmoid(s * acc + self.V[bt.uid[b]] + self.b)
            so = sp[base + d]
            w2 = rc[so]
            p = co.sigmoid(h @ w2 + bc[so])
            y = bt.y[b]
            if loss_kind == 0:
                ctx['lossb'][b] = co.bce_elem(np.float64(y), p, dt)
                dp = co.bce_grad(np.float64(y), p, dt) / b_norm
            else:
                ctx['lossb'][b] = (p - y) ** 2
                dp = 2 * (p - y) / b_norm
            dz2 = dp * p * (1 - p)
            ctx['dz2'][b] = dz2
            ctx['g2'][b] = dz2 * h
            ctx['dz1'][b] = dz2 * w2 * h * (1 - h)
        return ctx

    def _update(self, name, row, g, b_norm, reg=True):
        p = getattr(self, name)
        if reg:
            g = g + self.reg / b_norm * p[row]
        self.acc[name][row] = self.acc[name][row] + g * g
        p[row] = p[row] - float(np.float32(self.lr)) * g / (np.sqrt(self.acc[name][row]) + co.ADAGRAD_EPS)

    def reduce(self, idx, bpos, q_item, b_norm, q, opt, ctx):
        ks, vs, ss, bp = idx['keys_s'].numpy(), idx['vals_s'].numpy(), idx['slot_sorted'].numpy(), bpos.numpy()
        gc, gb2c = np.zeros((q_item, self.k)), np.zeros(q_item)
        s = 1.0 / (1.0 - q)
        gv = {}
        for key, pos, slot in zip(ks, vs, ss):
            if key == NONE:
                continue
            b = bp[pos]
            if key >= self.uk0:
                gv[key - self.uk0] = gv.get(key - self.uk0, 0) + ctx['dz1'][b]
            elif (key % (2 * self.ipr)) >= self.ipr:
                gc[slot] += ctx['g2'][b]
                gb2c[slot] += ctx['dz2'][b]
            else:
                gc[slot] += s * ctx['dz1'][b]
        for u, g in gv.items():
            self._update('V', u, g, b_norm)
        return torch.from_numpy(gc), torch.from_numpy(gb2c)

    def apply(self, recv_keys, recv_rows, recv_b2, b_norm, opt, recv_counts=None):
        k, r, s = recv_keys.numpy(), recv_rows.numpy(), recv_b2.numpy()
        tot, tots = {}, {}
        for i, key in enumerate(k):              # arrival order = source-rank order
            tot[int(key)] = tot.get(int(key), 0) + r[i]
            tots[int(key)] = tots.get(int(key), 0) + s[i]
        for key in sorted(tot):
            is_out, row = self._local(key)
            self._update('W2T' if is_out else 'W', row, tot[key], b_norm)
            if is_out:
                self._update('b2', row, tots[key], b_norm, reg=False)

    def bias_grad(self, B, ctx):
        return torch.from_numpy(np.concatenate([ctx['dz1'].sum(axis=0), [ctx['lossb'].sum()]]))

    def bias_apply(self, grad, b_norm, opt):
        self._update('b', slice(None), grad.numpy()[:self.k], b_norm, reg=False)
