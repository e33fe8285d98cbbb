"""NumPy statement of the per-rank work of the row-sharded step (same interface as drecpy_amd.dist.HipShardOps, same wire format:
include/drx.h "EXCHANGE BUFFER"), used by the world-size-2 gloo tests to exercise drecpy_amd/dist.py's exchange logic on CPU.
TEST INFRASTRUCTURE: float64, loops."""
import types

import numpy as np
import torch

from oracle import cdae_oracle as co
from drecpy_amd.dist import chunk_floats, items_per_rank, pad32, wire_chunks, wire_key, wire_local, wire_shift

NONE = 0xFFFFFFFF


def np_batch(uid, iid, y, indptr, q, mask_seed=0, keep=None):
    uid = np.asarray(uid, np.int64)
    deg = indptr[uid + 1] - indptr[uid]
    keep_off = np.zeros(len(uid) + 1, np.int64)
    keep_off[1:] = np.cumsum(deg)
    return types.SimpleNamespace(B=len(uid), uid=uid, iid=np.asarray(iid, np.int64), y=np.asarray(y, np.float64),
                                 keep_off=keep_off, keep=keep, mask_seed=mask_seed, q=float(np.float32(q)),
                                 n_touch_slots=int(keep_off[-1]))


def _chunk_offsets(counts, ld, skip=None, world=1, own_last=False):
    """(first key index, float offset of the rows, float offset of the scalars) of every piece of an exchange buffer whose pieces —
    segments on the owner's side, units on the requester's — hold `counts` rows.  skip: the rank whose pieces are NOT in the buffer
    (self-bypass) — entry None, or, with own_last (requester side), placed behind all the others in order"""
    out, k, f = [], 0, 0
    for i, c in enumerate(counts):
        if skip is not None and i % world == skip:
            out.append(None)
        else:
            out.append((k, f, f + int(c) * ld))
            f += int(c) * ld + pad32(c)
        k += int(c)
    if own_last and skip is not None:
        k = 0
        for i, c in enumerate(counts):
            if i % world == skip:
                out[i] = (k, f, f + int(c) * ld)
                f += int(c) * ld + pad32(c)
            k += int(c)
    return out


class NumpyShardOps:
    def __init__(self, n_users_local, n_items, k, rank, world, indptr, indices, lr, reg, self_bypass=True, chunks=1):
        self.rank, self.world, self.k = rank, world, k
        self.self_bypass = self_bypass
        self.ld = k
        self.ipr = items_per_rank(n_items, world)
        self.shift = wire_shift(self.ipr)
        self.chunks = wire_chunks(self.ipr, chunks)
        self.cshift = self.shift - (self.chunks.bit_length() - 1)         # log2 of a unit's key span
        self.n_items, self.n_users_local = n_items, n_users_local
        self.indptr, self.indices = np.asarray(indptr, np.int64), np.asarray(indices, np.int64)
        self.lr, self.reg = lr, reg
        self.engine = None

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
        return self.indices[s:e][kf]

    # -- parameter-independent
    def _wire(self, n, is_out):
        return wire_key(int(n), self.ipr, is_out, self.world, self.chunks)

    def prepare(self, bt):
        """Distinct wire keys of the batch's item rows, ascending, every UNIT's run (chunk-major, then owner) closed by a sentinel;
        counts owner-major ([owner][chunk]: what the count exchange sends)."""
        keys = set()
        for b in range(bt.B):
            for n in self._kept(bt, b):
                keys.add(self._wire(n, False))
            keys.add(self._wire(bt.iid[b], True))
        uniq, per_unit = [], []
        for v in range(self.world * self.chunks):
            mine = sorted(k for k in keys if (k >> self.cshift) == v)
            uniq += mine + [NONE]
            per_unit.append(len(mine) + 1)
        counts = [per_unit[c * self.world + o] for o in range(self.world) for c in range(self.chunks)]
        pos = {k: i for i, k in enumerate(uniq) if k != NONE}
        return {'uniq': torch.tensor(uniq, dtype=torch.int64), 'counts': counts, 'pos': pos, 'per_unit': per_unit}

    def owner_index(self, req, recv_counts, slot=0, chunk=0):
        return None

    def _local(self, key):
        o, c, is_out, row = wire_local(int(key), self.ipr, self.world, self.chunks)
        assert o == self.rank
        return is_out, row

    def xsplits(self, counts):
        f = chunk_floats(counts, self.ld)
        return [0 if (self.self_bypass and i % self.world == self.rank) else x for i, x in enumerate(f)]

    def _skip(self):
        return self.rank if self.self_bypass else None

    # -- parameter-dependent
    def gather_rows(self, req, recv_counts, out=None):
        r = req.numpy()
        out = np.zeros(sum(self.xsplits(recv_counts)))
        for ch, c in zip(_chunk_offsets(recv_counts, self.ld, self._skip(), self.world), recv_counts):
            if ch is None:
                continue
            k0, f0, s0 = ch
            for i in range(int(c)):
                key = int(r[k0 + i])
                if key == NONE:
                    continue
                is_out, row = self._local(key)
                out[f0 + i * self.ld:f0 + (i + 1) * self.ld] = self.W2T[row] if is_out else self.W[row]
                out[s0 + i] = self.b2[row] if is_out else 0.0
        return torch.from_numpy(out)

    def _update(self, name, row, g, b_norm, reg=True):
        p = getattr(self, name)
        if reg:
            g = g + self.reg / b_norm * p[row]
        self.acc[name][row] = self.acc[name][row] + g * g
        p[row] = p[row] - float(np.float32(self.lr)) * g / (np.sqrt(self.acc[name][row]) + co.ADAGRAD_EPS)

    def local_step(self, bt, P, rows_cache, b_norm, loss_kind, opt, events=None):
        cache = rows_cache.numpy()
        B, K = bt.B, self.k
        dt = np.dtype(np.float64)
        offs = _chunk_offsets(P['per_unit'], self.ld, self._skip(), self.world, own_last=True)

        def where(key):
            v = key >> self.cshift
            k0, f0, s0 = offs[v]
            i = P['pos'][key] - k0
            return f0 + i * self.ld, s0 + i

        def row_of(key):                       # an own row comes from the tables, any other from the cache
            if self.self_bypass and (key >> self.cshift) % self.world == self.rank:
                is_out, row = self._local(key)
                return (self.W2T[row], self.b2[row]) if is_out else (self.W[row], 0.0)
            r0, s0 = where(key)
            return cache[r0:r0 + K], cache[s0]
        gsend = np.zeros(sum(chunk_floats(P['per_unit'], self.ld)))
        s = 1.0 / (1.0 - bt.q)
        gv, gb, loss = {}, np.zeros(K), 0.0
        for b in range(B):
            kept = [self._wire(n, False) for n in self._kept(bt, b)]
            acc = np.zeros(K)
            for key in kept:
                acc += row_of(key)[0]
            u = int(bt.uid[b])
            h = co.sigmoid(s * acc + self.V[u] + self.b)
            ko = self._wire(bt.iid[b], True)
            ro, so = where(ko)
            w2, b2v = row_of(ko)
            p = co.sigmoid(h @ w2 + b2v)
            y = bt.y[b]
            if loss_kind == 0:
                loss += co.bce_elem(np.float64(y), p, dt)
                dp = co.bce_grad(np.float64(y), p, dt) / b_norm
            else:
                loss += (p - y) ** 2
                dp = 2 * (p - y) / b_norm
            dz2 = dp * p * (1 - p)
            dz1 = dz2 * w2 * h * (1 - h)
            gsend[ro:ro + K] += dz2 * h
            gsend[so] += dz2
            for key in kept:
                r0, _ = where(key)
                gsend[r0:r0 + K] += s * dz1
            gv[u] = gv.get(u, 0) + dz1
            gb += dz1
        for u, g in gv.items():
            self._update('V', u, g, b_norm)
        for (k0, f0, s0), c in zip(offs, P['per_unit']):            # the sentinel rows (every unit's): bias gradient, loss sum
            i = int(c) - 1
            gsend[f0 + i * self.ld:f0 + (i + 1) * self.ld] = gb
            gsend[s0 + i] = loss
        return torch.from_numpy(gsend)

    def apply(self, req, grecv, recv_counts, table, b_norm, opt, want_loss=False, own=None, chunk=0):
        """one exchange chunk: the rows of its key range; b and the loss with the LAST chunk (its sentinels)"""
        k, g = req.numpy(), grecv.numpy()
        tot, tots = {}, {}
        gb, loss = np.zeros(self.k), 0.0
        offs = _chunk_offsets(recv_counts, self.ld, self._skip(), self.world)
        k0 = 0
        for s, (ch, c) in enumerate(zip(offs, recv_counts)):       # segment order = micro-batch, then source rank
            if ch is None:                                          # the piece this rank "sent" to itself: still in its own buffer
                og, f0 = own[s // self.world]
                src = og.numpy()
                s0 = f0 + int(c) * self.ld
            else:
                src, (_, f0, s0) = g, ch
            for i in range(int(c)):
                key = int(k[k0 + i])
                row, sc = src[f0 + i * self.ld:f0 + (i + 1) * self.ld], src[s0 + i]
                if key == NONE:
                    gb, loss = gb + row, loss + sc
                else:
                    tot[key] = tot.get(key, 0) + row
                    tots[key] = tots.get(key, 0) + sc
            k0 += int(c)
        for key in sorted(tot):
            is_out, row = self._local(key)
            self._update('W2T' if is_out else 'W', row, tot[key], b_norm)
            if is_out:
                self._update('b2', row, tots[key], b_norm, reg=False)
        if chunk == self.chunks - 1:
            self._update('b', slice(None), gb, b_norm, reg=False)
        return loss / b_norm if want_loss else None
