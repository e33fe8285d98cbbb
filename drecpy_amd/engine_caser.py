"""Device-side state of one Caser model and the calls into libdrx.so (drx_caser_*, drx_scatter_rows, drx_adam_*).

One training step = the body of the reference fit() loop for Caser (recommender_abc.py:190-204 over caser.py:86-120), three launches:
  drx_caser_step_small      forward, Keras BCE, backward (k_caser_tile: tiles of 16 samples on the matrix cores) -> one gradient row per
                            item / user lookup, the score gradients and hidden rows of the dense_1 lookups, the workgroups' partial sums of
                            the small-weight gradients; then those sums and the conv / dense_0 kernels' and biases' l2 + Keras Adam
  drx_rows_csr_adam_multi   the lookups' rows -> gradient + L2 (Keras l2(reg): 2*reg*w) + Keras Adam of item_emb, user_emb and dense_1_W
                            (+ dense_1_b; its rows formed as score gradient x hidden row where they are summed) in one pass per table, from
                            the lookups grouped by row on the host (drx_batch_csr); one lr_t per registered layer
(table_update == 'scatter', for batches already on the device: drx_scatter_rows x3 into a zeroed gradient arena + drx_adam_dense x4)
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AdamSegments, CaserArgs, CaserDims, check, lib, ptr, stream_ptr
from .engine import ADAM_B1, ADAM_B2, ADAM_EPS, CdaeEngine, _round_up


class CaserEngine:
    ACTIVATIONS = {'relu': 0, 'tanh': 1, 'sigmoid': 2, 'linear': 3, None: 3}

    def __init__(self, n_users, n_items, L=5, T=3, neg_ratio=3, d=50, n_v=4, n_h=16, device='cuda:0', act_h='relu', act_mlp='relu'):
        if not torch.cuda.is_available():
            raise _lib.DrxError('drecpy_amd needs a ROCm GPU (MI355X); there is no CPU fallback.')
        lib()
        if not (1 <= L <= 8 and 1 <= d <= 64):
            raise _lib.DrxError(f'Caser engine: 1 <= L <= 8 and 1 <= d <= 64 are supported (lane = embedding channel of one wavefront), got L={L}, d={d}')
        if act_h not in self.ACTIVATIONS or act_mlp not in self.ACTIVATIONS:
            raise _lib.DrxError(f'Caser engine: activations {sorted(k for k in self.ACTIVATIONS if k)} are supported, got act_h={act_h!r}, act_mlp={act_mlp!r}')
        self.device = torch.device(device)
        self.U, self.N, self.L, self.T, self.d, self.n_v, self.n_h = n_users, n_items, L, T, d, n_v, n_h
        self.Tp = T + T * neg_ratio
        self.ld, self.ld2 = _round_up(d, 4), _round_up(2 * d, 4)
        self.nx = n_v + L * n_h
        ld = self.ld
        off = 0
        D = CaserDims()
        D.L, D.T, D.Tp, D.d, D.ld, D.ld2, D.n_v, D.n_h = L, T, self.Tp, d, ld, self.ld2, n_v, n_h
        self.seg = []                    # (name, start, len, regularised, layer index)
        D.off_kv = off; self.seg.append(('conv_v_k', off, L * n_v * ld, True, 2)); off += L * n_v * ld
        D.off_bv = off; self.seg.append(('conv_v_b', off, n_v, False, 2)); off += _round_up(n_v, 4)
        for i in range(L):
            D.off_kh[i] = off; self.seg.append((f'conv_h{i}_k', off, (i + 1) * n_h * ld, True, 3 + i)); off += (i + 1) * n_h * ld
            D.off_bh[i] = off; self.seg.append((f'conv_h{i}_b', off, n_h, False, 3 + i)); off += _round_up(n_h, 4)
        D.off_wd = off; self.seg.append(('dense0_k', off, self.nx * ld, True, 3 + L)); off += self.nx * ld
        D.off_bd = off; self.seg.append(('dense0_b', off, ld, False, 3 + L)); off += ld
        D.n_small = off
        D.act_h, D.act_mlp = self.ACTIVATIONS[act_h], self.ACTIVATIONS[act_mlp]
        self.D = D
        # (the kernels keep a tile's scratch and, where they fit, the small weights in a workgroup's LDS: drx_caser_grid answers 0 when
        #  n_v + L * n_h is too large for that — Recommender/caser.py then takes the generic engine)
        if int(lib().drx_caser_grid(C.byref(D), 16)) < 1:
            raise _lib.DrxError(f'Caser engine: n_v + L * n_h = {self.nx} units (n_v={n_v}, n_h={n_h}, L={L}, d={d}) do not fit the kernels\' LDS tiles')
        self.n_layers = 6 + L
        z = dict(dtype=torch.float32, device=self.device)
        self.item_emb = torch.zeros(n_items, ld, **z)
        self.user_emb = torch.zeros(n_users, ld, **z)
        self.W1 = torch.zeros(n_items, self.ld2, **z)
        self.b1 = torch.zeros(n_items, **z)
        self.sw = torch.zeros(off, **z)
        self.state = {n: (torch.zeros_like(t), torch.zeros_like(t)) for n, t in self.tensors().items()}
        # the dense gradient tables of the scatters live in ONE buffer, zeroed by one kernel per step
        scat = [n for n in ('item_emb', 'W1', 'b1', 'user_emb') if n in self.tensors()]
        sizes = [(self.tensors()[n].numel() + 63) // 64 * 64 for n in scat]
        self._grad_arena = torch.zeros(sum(sizes), dtype=torch.float32, device=self.device)
        self._grads, o = {}, 0
        for n, sz in zip(scat, sizes):
            self._grads[n] = self._grad_arena[o:o + self.tensors()[n].numel()].view(self.tensors()[n].shape)
            o += sz
        for n, t in self.tensors().items():
            if n not in self._grads:
                self._grads[n] = torch.zeros_like(t)
        self._scratch = None
        self.lr, self.reg = 1e-3, 1e-3
        self.beta1, self.beta2, self.eps = ADAM_B1, ADAM_B2, ADAM_EPS
        from .Recommender.trainables import TrainableLayer
        # the 6 + L Keras layers caser.py:47-70 registers, in that order (= the order of the per-step Adam applies)
        seg_views = lambda j: [self.sw[st:st + n] for _, st, n, _, layer in self.seg if layer == j]
        self.layers = [TrainableLayer('user_embeddings', lambda: [self.user_emb]), TrainableLayer('item_embeddings', lambda: [self.item_emb]),
                       TrainableLayer('conv_v', lambda: seg_views(2))]
        self.layers += [TrainableLayer(f'convs_h[{i}]', (lambda i=i: seg_views(3 + i))) for i in range(L)]
        self.layers += [TrainableLayer('dense_0', lambda: seg_views(3 + L)), TrainableLayer('dense_1_W', lambda: [self.W1]),
                        TrainableLayer('dense_1_b', lambda: [self.b1])]

    def tensors(self):
        return {'user_emb': self.user_emb, 'item_emb': self.item_emb, 'W1': self.W1, 'b1': self.b1, 'sw': self.sw}

    # ---- parameters in the reference's layout (oracle/caser_oracle.py:init_params) -----------------------------
    def set_params(self, p):
        d, ld = self.d, self.ld
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device)
        for x in self.tensors().values():
            x.zero_()
        self.user_emb[:, :d] = t(p['user_emb'])
        self.item_emb[:, :d] = t(p['item_emb'])
        self.W1[:, :2 * d] = t(p['W1'])
        self.b1.copy_(t(p['b1']).reshape(-1))
        for name, start, n, _, _ in self.seg:
            if name.endswith('_k') and name.startswith('conv'):
                k = t(p[name]).permute(0, 2, 1).contiguous()            # [s, d, f] -> [s, f, d]
                view = self.sw[start:start + n].view(k.shape[0], k.shape[1], ld)
                view[:, :, :d] = k
            elif name == 'dense0_k':
                self.sw[start:start + n].view(self.nx, ld)[:, :d] = t(p[name])
            elif name == 'dense0_b':
                self.sw[start:start + d] = t(p[name])
            else:
                self.sw[start:start + n] = t(p[name])

    def get_params(self):
        d, ld = self.d, self.ld
        c = lambda x: x.detach().cpu().numpy().copy()
        p = {'user_emb': c(self.user_emb[:, :d]), 'item_emb': c(self.item_emb[:, :d]), 'W1': c(self.W1[:, :2 * d]),
             'b1': c(self.b1).reshape(-1, 1)}
        for name, start, n, _, _ in self.seg:
            if name.endswith('_k') and name.startswith('conv'):
                s_ = n // (ld * (self.n_v if name == 'conv_v_k' else self.n_h))
                f = self.n_v if name == 'conv_v_k' else self.n_h
                p[name] = c(self.sw[start:start + n].view(s_, f, ld)[:, :, :d].permute(0, 2, 1))
            elif name == 'dense0_k':
                p[name] = c(self.sw[start:start + n].view(self.nx, ld)[:, :d])
            elif name == 'dense0_b':
                p[name] = c(self.sw[start:start + d])
            else:
                p[name] = c(self.sw[start:start + n])
        return p

    def snapshot(self):
        return {'p': {n: t.clone() for n, t in self.tensors().items()}}

    def restore(self, snap, with_optimizer=False):
        for n, t in self.tensors().items():
            t.copy_(snap['p'][n])

    # ---- helpers --------------------------------------------------------------------------------------------------
    def _dev_i32(self, a):
        if torch.is_tensor(a):
            return a.to(self.device, torch.int32).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)

    def _scatter(self, keys, T, src, ld, n_rows, out, src_s=None, out_s=None, stream=None):
        """keys / src / src_s / out / out_s: device addresses.  `out` (and `out_s`) must already be zero: step() clears the whole
        gradient arena once."""
        sb = self.__dict__.setdefault('_scatter_need', {})
        need = sb.get((ld, T, n_rows))
        if need is None:
            need = sb[(ld, T, n_rows)] = lib().drx_scatter_scratch_bytes(ld, T, n_rows)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(int(need * 1.2) + 1024, dtype=torch.uint8, device=self.device)
        check(lib().drx_scatter_rows(keys, T, src, None, None, src_s, ld, n_rows, out, out_s, self._scratch.data_ptr(),
                                     self._scratch.numel(), stream if stream is not None else stream_ptr(self.device)), 'drx_scatter_rows')

    def _adam(self, name, grad, alpha, l2c, stream=None):
        p = getattr(self, name)
        m, v = self.state[name]
        check(lib().drx_adam_dense(p.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), p.numel(), alpha, l2c, self.beta1, self.beta2,
                                   self.eps, stream if stream is not None else stream_ptr(self.device)), 'drx_adam_dense')

    def _alphas(self, step_idx):
        """Keras-Adam lr_t of the n_layers applies of step `step_idx` (t = n_layers * step + j + 1), fp32 like optimizer_v2/adam.py."""
        f = np.float32
        t = (self.n_layers * step_idx + 1 + np.arange(self.n_layers)).astype(np.float32)
        return (f(self.lr) * np.sqrt(f(1.0) - np.power(f(self.beta2), t)) / (f(1.0) - np.power(f(self.beta1), t))).astype(np.float32).tolist()

    def _args(self, uid, before, after=None, keep=None, rate=0.0, mask_seed=0):
        A = CaserArgs()
        A.item_emb, A.user_emb, A.W1, A.b1, A.sw = (t.data_ptr() for t in (self.item_emb, self.user_emb, self.W1, self.b1, self.sw))
        A.uid, A.before = uid.data_ptr(), before.data_ptr()
        A.after = after.data_ptr() if after is not None else None
        A.keep = keep.data_ptr() if keep is not None else None
        A.rate, A.B = float(rate), int(uid.numel())
        A.mask_seed = int(mask_seed) & (2 ** 64 - 1)
        return A

    # how the four lookup tables (user / item embeddings, dense_1_W + dense_1_b) are updated: 'csr' = drx_rows_csr_adam, gradient and
    # dense Adam in one pass per table from the lookups grouped by row on the host (prepare_batch); 'scatter' = drx_scatter_rows into
    # a zeroed gradient arena + drx_adam_dense (the only way for batches that are already on the device)
    table_update = 'csr'

    # batches already on the device: group their lookups by table row on the device (True) or take the scatter + dense-Adam update
    device_csr = True

    def prepare_device_batch(self, uid, bef, aft):
        """A batch that is already on the device, its lookups grouped by table row on the CURRENT stream (which may be a run-ahead
        stream: nothing here depends on the parameters) — what step() takes in place of (uids, before, after).  The three slots of the
        ring are reused only behind the step that last read them (an event recorded by that step on ITS stream)."""
        uid, bef, aft = self._dev_i32(uid), self._dev_i32(bef), self._dev_i32(aft)
        prep = {'uid': uid, 'before': bef, 'after': aft, 'B': uid.numel()}
        prep['csr_dev'] = self._csr_on_device(uid, bef, aft, prep)
        return prep

    # ---- the same with the batch's id tensors OWNED by the ring (Caser.fit(device_sampler=True): the sampler writes into them) ----------
    # Everything a draw and a step hand to the library is then the same from visit to visit of a slot: the argument structs are built
    # once per slot (at 4096 windows the host's Python, not the 79 us of training kernels, bounded the step: r06,
    # profiles/r06_caser_host_profile.txt).
    def device_slot(self, B):
        """The next ring slot for a batch of B windows drawn on the device: {'uid', 'before', 'after'} int32 tensors to fill on the CURRENT
        stream (which has been made to wait for the step that last read the slot), then group_slot(slot)."""
        st = self.__dict__.setdefault('_dev_slots', {})
        if st.get('B') != B:
            i32 = dict(dtype=torch.int32, device=self.device)
            sizes = (self.N + 1, B * self.L, self.N + 1, B * self.Tp, self.U + 1, B)       # ptrE, ordE, ptrW, ordW, ptrU, ordU
            offs, total = [], 0
            for n in sizes:
                offs.append(total)
                total += (4 * n + 15) & ~15
            st.clear()
            st.update({'B': B, 'i': 0, 'slots': []})
            for k in range(3):
                sl = {'k': k, 'B': B, 'uid': torch.empty(B, **i32), 'before': torch.empty(B, self.L, **i32),
                      'after': torch.empty(B, self.Tp, **i32), 'buf': torch.empty(total, dtype=torch.uint8, device=self.device), 'free': None}
                base = sl['buf'].data_ptr()
                ls = (_lib.CsrList * 3)()
                for l, (keys, T, n_rows, kp_, ko_) in zip(ls, ((sl['before'], B * self.L, self.N, 0, 1), (sl['after'], B * self.Tp, self.N, 2, 3),
                                                            (sl['uid'], B, self.U, 4, 5))):
                    l.keys, l.T, l.n_rows, l.row_ptr, l.order = keys.data_ptr(), T, n_rows, base + offs[kp_], base + offs[ko_]
                sl['lists'] = ls
                sl['csr_dev'] = [base + o for o in offs]
                st['slots'].append(sl)
            need = int(lib().drx_batch_csr_device_bytes(st['slots'][0]['lists'], 3))
            if need <= 0:
                raise _lib.DrxError('drx_batch_csr_device_bytes: invalid lists')
            st['scratch'] = torch.empty(need, dtype=torch.uint8, device=self.device)
            st['scratch_args'] = (st['scratch'].data_ptr(), st['scratch'].numel())
        sl = st['slots'][st['i'] % 3]
        st['i'] += 1
        if sl['free'] is not None:
            torch.cuda.current_stream(self.device).wait_event(sl['free'])
            sl['free'] = None
        return sl

    def group_slot(self, sl):
        """the lookups of a filled slot grouped by table row (drx_batch_csr_device) on the current stream; returns what step() takes"""
        check(lib().drx_batch_csr_device(sl['lists'], 3, *self._dev_slots['scratch_args'], stream_ptr(self.device)), 'drx_batch_csr_device')
        return sl

    def _step_slot(self, step_idx, sl, keep, rate, mask_seed):
        """step() on a ring slot (no loss wanted): the argument structs of the slot's earlier visit with this step's scalars"""
        L_ = lib()
        B = sl['B']
        c = sl.get('step')
        n_dE, n_dW1, n_dPu = B * self.L * self.ld, B * self.ld2, B * self.ld
        wk = getattr(self, '_step_bufs', None)
        if wk is None or wk[0] != (B, True):               # (the work buffers of a batch size, shared with the general path)
            z = dict(dtype=torch.float32, device=self.device)
            grid = L_.drx_caser_grid(C.byref(self.D), B)
            rows = torch.zeros(n_dE + n_dW1 + n_dPu, **z)
            wk = self._step_bufs = ((B, True), rows, torch.empty(B * self.Tp, **z), torch.empty(grid, self.D.n_small, **z), torch.empty(grid, **z),
                                    torch.empty(self.D.n_small + 1, **z))
        key = (self.item_emb.data_ptr(), self.sw.data_ptr(), self.state['sw'][0].data_ptr(), self.state['W1'][0].data_ptr(), self.reg, self.beta1,
               self.beta2, self.eps, id(self.D), id(wk))
        if c is None or c['key'] != key:
            _, rows, db1, gpart, lpart, gsw = wk
            base = rows.data_ptr()
            p_dE, p_dW1, p_dPu = base, base + 4 * n_dE, base + 4 * (n_dE + n_dW1)
            A = CaserArgs()
            A.item_emb, A.user_emb, A.W1, A.b1, A.sw = (t.data_ptr() for t in (self.item_emb, self.user_emb, self.W1, self.b1, self.sw))
            A.uid, A.before, A.after = sl['uid'].data_ptr(), sl['before'].data_ptr(), sl['after'].data_ptr()
            A.B = int(B)
            A.dE, A.db1, A.dPu, A.gsw_part, A.loss_part = p_dE, db1.data_ptr(), p_dPu, gpart.data_ptr(), lpart.data_ptr()
            A.dW1, A.cat_out = None, p_dW1
            l2c = 2.0 * self.reg
            sg = AdamSegments()
            sg.n = len(self.seg)
            for i, (_, start, n, regd, layer) in enumerate(self.seg):
                sg.start[i], sg.len[i], sg.l2_coef[i] = start, n, (l2c if regd else 0.0)
            ptrE, ordE, ptrW, ordW, ptrU, ordU = sl['csr_dev']
            st_ = self.state
            tabs = (_lib.CsrAdamTable * 3)()
            for t, (rp, od, src, scale, group, ld, n_rows, name, sname) in zip(tabs, (
                    (ptrU, ordU, p_dPu, None, 0, self.ld, self.U, 'user_emb', None),
                    (ptrE, ordE, p_dE, None, 0, self.ld, self.N, 'item_emb', None),
                    (ptrW, ordW, p_dW1, db1.data_ptr(), self.Tp, self.ld2, self.N, 'W1', 'b1'))):
                t.row_ptr, t.order, t.src, t.scale, t.group, t.ld, t.n_rows = rp, od, src, scale, group, ld, n_rows
                t.p, (t.m, t.v) = getattr(self, name).data_ptr(), (x.data_ptr() for x in st_[name])
                if sname is not None:
                    t.p_s, (t.m_s, t.v_s) = getattr(self, sname).data_ptr(), (x.data_ptr() for x in st_[sname])
                t.l2_coef = l2c
            m, v = self.state['sw']
            c = sl['step'] = {'key': key, 'A': A, 'sg': sg, 'tabs': tabs, 'keep': wk,
                              'small': (C.byref(self.D), C.byref(A), gsw.data_ptr(), self.sw.data_ptr(), m.data_ptr(), v.data_ptr(), C.byref(sg),
                                        self.beta1, self.beta2, self.eps),
                              'layers': [layer for (_, _, _, _, layer) in self.seg]}
        A, sg, tabs = c['A'], c['sg'], c['tabs']
        kp = None
        if keep is not None:
            kp = torch.as_tensor(np.ascontiguousarray(keep, dtype=np.uint8)).to(self.device) if not torch.is_tensor(keep) \
                else keep.to(self.device, torch.uint8).contiguous()
        A.keep = kp.data_ptr() if kp is not None else None
        A.rate = float(rate)
        A.mask_seed = int(mask_seed) & (2 ** 64 - 1)
        alpha = self._alphas(step_idx)
        for i, layer in enumerate(c['layers']):
            sg.alpha[i] = alpha[layer]
        tabs[0].alpha, tabs[1].alpha, tabs[2].alpha, tabs[2].alpha_s = alpha[0], alpha[1], alpha[4 + self.L], alpha[5 + self.L]
        stream = stream_ptr(self.device)
        check(L_.drx_caser_step_small(*c['small'], stream), 'drx_caser_step_small')
        check(L_.drx_rows_csr_adam_multi(tabs, 3, self.beta1, self.beta2, self.eps, stream), 'drx_rows_csr_adam_multi')
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        sl['free'] = ev
        return None

    def _csr_on_device(self, uid, bef, aft, prep=None):
        B = uid.numel()
        sizes = (self.N + 1, B * self.L, self.N + 1, B * self.Tp, self.U + 1, B)           # ptrE, ordE, ptrW, ordW, ptrU, ordU
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (4 * n + 15) & ~15
        st = self.__dict__.setdefault('_dev_csr', {})
        if st.get('key') != (B, total):
            st['key'] = (B, total)
            st['ring'] = [torch.empty(total, dtype=torch.uint8, device=self.device) for _ in range(3)]
            st['free'] = [None] * 3                        # recorded by the step that read the slot, on its stream
            st['i'] = 0
            st['scratch'] = None
        k = st['i'] % 3
        buf = st['ring'][k]
        st['i'] += 1
        if st['free'][k] is not None:
            torch.cuda.current_stream(self.device).wait_event(st['free'][k])
            st['free'][k] = None
        if prep is not None:
            prep['slot'] = k
        base = buf.data_ptr()
        ls = (_lib.CsrList * 3)()
        for l, (keys, T, n_rows, kp_, ko_) in zip(ls, ((bef, B * self.L, self.N, 0, 1), (aft, B * self.Tp, self.N, 2, 3), (uid, B, self.U, 4, 5))):
            l.keys, l.T, l.n_rows, l.row_ptr, l.order = keys.data_ptr(), T, n_rows, base + offs[kp_], base + offs[ko_]
        if st['scratch'] is None:
            need = int(lib().drx_batch_csr_device_bytes(ls, 3))
            if need <= 0:
                raise _lib.DrxError('drx_batch_csr_device_bytes: invalid lists')
            st['scratch'] = torch.empty(need, dtype=torch.uint8, device=self.device)
        check(lib().drx_batch_csr_device(ls, 3, st['scratch'].data_ptr(), st['scratch'].numel(), stream_ptr(self.device)), 'drx_batch_csr_device')
        st['keep'] = (uid, bef, aft)                       # (alive until the next step has been queued behind this one)
        return [base + o for o in offs]

    def prepare_batch(self, uids, before, after):
        """Host half of a step (Caser.fit() runs it on the sampler's worker thread): the batch as int32 arrays plus, for each of the
        three lookup lists, the lookups grouped by the table row they name (drx_batch_csr) — one host buffer for one upload."""
        B = len(uids)
        L_ = lib()
        sizes = (B, B * self.L, B * self.Tp, self.N + 1, B * self.L, self.N + 1, B * self.Tp, self.U + 1, B)
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (4 * n + 15) & ~15
        buf = np.empty(total, np.uint8)
        view = lambda k: buf[offs[k]:offs[k] + 4 * sizes[k]].view(np.int32)
        view(0)[:] = uids
        view(1)[:] = np.asarray(before).reshape(-1)
        view(2)[:] = np.asarray(after).reshape(-1)
        base = buf.ctypes.data
        for keys, T, n_rows, kp, ko in ((1, B * self.L, self.N, 3, 4), (2, B * self.Tp, self.N, 5, 6), (0, B, self.U, 7, 8)):
            check(L_.drx_batch_csr(base + offs[keys], T, n_rows, base + offs[kp], base + offs[ko]), 'drx_batch_csr')
        return {'buf': buf, 'offs': offs, 'B': B}

    def _upload(self, buf):
        st = self.__dict__.setdefault('_ring', {'i': 0, 'host': [None] * 4, 'ev': [None] * 4})
        total = buf.nbytes
        k = st['i'] % 4
        st['i'] += 1
        if st['host'][k] is None or st['host'][k].numel() < total:
            st['host'][k] = torch.empty(int(total * 1.5) + 4096, dtype=torch.uint8, pin_memory=True)
            st['ev'][k] = torch.cuda.Event()
        else:
            st['ev'][k].synchronize()                 # the copy that last read this pinned slot has finished
        st['host'][k].numpy()[:total] = buf
        dev = torch.empty(total, dtype=torch.uint8, device=self.device)
        dev.copy_(st['host'][k][:total], non_blocking=True)
        st['ev'][k].record(torch.cuda.current_stream(self.device))
        return dev

    # ---- one training step ---------------------------------------------------------------------------------------
    def step(self, step_idx, uids, before=None, after=None, keep=None, rate=0.0, want_loss=False, mask_seed=0):
        """uids, before, after: the batch (arrays or device tensors) — or `uids` = what prepare_batch returned for it.
        keep: explicit dropout keep mask [B, n_v + L*n_h] or None; with None and rate > 0 the kernel evaluates the counter-based
        mask drx_hash_u32(mask_seed, b, j) >= rate * 2^32 itself (no 344 K host random numbers per batch of 4096)."""
        L_ = lib()
        csr = None
        slot = None
        if isinstance(uids, dict) and 'lists' in uids:             # a ring slot (device_slot / group_slot)
            if not want_loss and self.table_update == 'csr':
                return self._step_slot(step_idx, uids, keep, rate, mask_seed)
            sl = uids                                               # (with the loss: the general path below, on the slot's tensors)
            uids = {'uid': sl['uid'], 'before': sl['before'], 'after': sl['after'], 'B': sl['B'], 'csr_dev': sl['csr_dev'], 'ring_slot': sl}
        if isinstance(uids, dict) and 'csr_dev' in uids:           # prepare_device_batch: ids and grouped lookups already on the device
            prep = uids
            uid, bef, aft, B, csr, slot = prep['uid'], prep['before'], prep['after'], prep['B'], prep['csr_dev'], prep.get('slot')
            uid_p, bef_p, aft_p = uid.data_ptr(), bef.data_ptr(), aft.data_ptr()
            kp = None
            if keep is not None:
                kp = torch.as_tensor(np.ascontiguousarray(keep, dtype=np.uint8)).to(self.device) if not torch.is_tensor(keep) \
                    else keep.to(self.device, torch.uint8).contiguous()
        elif isinstance(uids, dict) or (self.table_update == 'csr' and not any(torch.is_tensor(a) for a in (uids, before, after))):
            prep = uids if isinstance(uids, dict) else self.prepare_batch(uids, before, after)
            dev = self._upload(prep['buf'])
            B = prep['B']
            pb = [dev.data_ptr() + o for o in prep['offs']]
            uid_p, bef_p, aft_p = pb[0], pb[1], pb[2]
            csr = pb[3:]
            kp = None
            if keep is not None:
                kp = torch.as_tensor(np.ascontiguousarray(keep, dtype=np.uint8)).to(self.device, non_blocking=True) if not torch.is_tensor(keep) \
                    else keep.to(self.device, torch.uint8).contiguous()
        else:
            if any(torch.is_tensor(a) for a in (uids, before, after, keep)):
                uid, bef, aft = self._dev_i32(uids), self._dev_i32(before), self._dev_i32(after)
                kp = None
                if keep is not None:
                    kp = torch.as_tensor(np.ascontiguousarray(keep, dtype=np.uint8)).to(self.device) if not torch.is_tensor(keep) \
                        else keep.to(self.device, torch.uint8).contiguous()
                if self.table_update == 'csr' and self.device_csr:
                    # a batch that is already on the device (Caser.fit(device_sampler=True)): the lookups are grouped by table row THERE
                    # (drx_batch_csr_device: one stable sort for the three lists) and the step takes the same two launches as a host batch
                    csr = self._csr_on_device(uid, bef, aft)
            else:                                          # host batch: one asynchronous copy for all of it
                if getattr(self, '_stage', None) is None:
                    from ._staging import StagedUpload
                    self._stage = StagedUpload(self.device)
                arrays = [np.ascontiguousarray(uids, dtype=np.int32), np.ascontiguousarray(before, dtype=np.int32),
                          np.ascontiguousarray(after, dtype=np.int32)]
                if keep is not None:
                    arrays.append(np.ascontiguousarray(keep, dtype=np.uint8))
                _owner, views = self._stage(arrays)
                uid, bef, aft = views[:3]
                kp = views[3] if keep is not None else None
            B = uid.numel()
            assert bef.shape == (B, self.L) and aft.shape == (B, self.Tp)
            uid_p, bef_p, aft_p = uid.data_ptr(), bef.data_ptr(), aft.data_ptr()
        z = dict(dtype=torch.float32, device=self.device)
        stream = stream_ptr(self.device)
        grid = L_.drx_caser_grid(C.byref(self.D), B)
        # dense_1's gradient rows are outer products, score gradient x [dense_0 output | user row]: with the 'csr' update the kernel writes
        # the B hidden rows and drx_rows_csr_adam_outer forms the B * Tp rows where it sums them (19.7 MB less to write and to read back
        # per step of 4096); the 'scatter' update takes them written out
        outer = csr is not None
        n_dE, n_dW1, n_dPu = B * self.L * self.ld, (B if outer else B * self.Tp) * self.ld2, B * self.ld
        # device work buffers of a batch size: steps run in order on one stream, so they are reused from step to step
        wk = getattr(self, '_step_bufs', None)
        if wk is None or wk[0] != (B, outer):
            rows = torch.zeros(n_dE + n_dW1 + n_dPu, **z)    # the padding columns of the gradient rows stay zero: the kernel never writes them
            wk = self._step_bufs = ((B, outer), rows, torch.empty(B * self.Tp, **z), torch.empty(grid, self.D.n_small, **z), torch.empty(grid, **z),
                                    torch.empty(self.D.n_small + 1, **z))
        _, rows, db1, gpart, lpart, gsw = wk
        base = rows.data_ptr()
        p_dE, p_dW1, p_dPu = base, base + 4 * n_dE, base + 4 * (n_dE + n_dW1)       # (p_dW1: the rows, or the B hidden rows)
        A = CaserArgs()
        A.item_emb, A.user_emb, A.W1, A.b1, A.sw = (t.data_ptr() for t in (self.item_emb, self.user_emb, self.W1, self.b1, self.sw))
        A.uid, A.before, A.after = uid_p, bef_p, aft_p
        A.keep = kp.data_ptr() if kp is not None else None
        A.rate, A.B = float(rate), int(B)
        A.mask_seed = int(mask_seed) & (2 ** 64 - 1)
        A.dE, A.db1, A.dPu, A.gsw_part, A.loss_part = p_dE, db1.data_ptr(), p_dPu, gpart.data_ptr(), lpart.data_ptr()
        A.dW1, A.cat_out = (None, p_dW1) if outer else (p_dW1, None)
        reg_loss = None
        if want_loss:                                   # Keras l2(reg) on the pre-update weights
            sq = _lib.sumsq([self.user_emb, self.item_emb, self.W1] + [self.sw[start:start + n] for _, start, n, regd, _ in self.seg if regd])
            reg_loss = self.reg * sq
        alpha = self._alphas(step_idx)
        l2c = 2.0 * self.reg
        sg = AdamSegments()
        sg.n = len(self.seg)
        for i, (_, start, n, regd, layer) in enumerate(self.seg):
            sg.start[i], sg.len[i], sg.alpha[i], sg.l2_coef[i] = start, n, alpha[layer], (l2c if regd else 0.0)
        m, v = self.state['sw']
        # forward / backward, then the small weights' Adam in the launch that sums the workgroups' partial gradients
        check(L_.drx_caser_step_small(C.byref(self.D), C.byref(A), gsw.data_ptr(), self.sw.data_ptr(), m.data_ptr(), v.data_ptr(), C.byref(sg),
                                      self.beta1, self.beta2, self.eps, stream), 'drx_caser_step_small')
        if csr is not None:
            # the three lookup tables (+ dense_1_b) in one launch: gradient rows summed per table row, l2, Keras Adam
            ptrE, ordE, ptrW, ordW, ptrU, ordU = csr
            st_ = self.state
            tabs = (_lib.CsrAdamTable * 3)()
            for t, (rp, od, src, scale, group, ld, n_rows, name, a, sname, a_s) in zip(tabs, (
                    (ptrU, ordU, p_dPu, None, 0, self.ld, self.U, 'user_emb', alpha[0], None, 0.0),
                    (ptrE, ordE, p_dE, None, 0, self.ld, self.N, 'item_emb', alpha[1], None, 0.0),
                    (ptrW, ordW, p_dW1, db1.data_ptr(), self.Tp, self.ld2, self.N, 'W1', alpha[4 + self.L], 'b1', alpha[5 + self.L]))):
                t.row_ptr, t.order, t.src, t.scale, t.group, t.ld, t.n_rows = rp, od, src, scale, group, ld, n_rows
                t.p, (t.m, t.v) = getattr(self, name).data_ptr(), (x.data_ptr() for x in st_[name])
                if sname is not None:
                    t.p_s, (t.m_s, t.v_s) = getattr(self, sname).data_ptr(), (x.data_ptr() for x in st_[sname])
                t.alpha, t.alpha_s, t.l2_coef = a, a_s, l2c
            check(L_.drx_rows_csr_adam_multi(tabs, 3, self.beta1, self.beta2, self.eps, stream), 'drx_rows_csr_adam_multi')
            if slot is not None:                            # (the ring slot of a device-prepared batch may be rebuilt behind this point)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._dev_csr['free'][slot] = ev
            if isinstance(uids, dict) and uids.get('ring_slot') is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                uids['ring_slot']['free'] = ev
        else:
            self._grad_arena.zero_()
            g = self._grads
            self._scatter(bef_p, B * self.L, p_dE, self.ld, self.N, g['item_emb'].data_ptr(), stream=stream)
            self._scatter(aft_p, B * self.Tp, p_dW1, self.ld2, self.N, g['W1'].data_ptr(), src_s=db1.data_ptr(), out_s=g['b1'].data_ptr(),
                          stream=stream)
            self._scatter(uid_p, B, p_dPu, self.ld, self.U, g['user_emb'].data_ptr(), stream=stream)
            self._adam('user_emb', g['user_emb'], alpha[0], l2c, stream)
            self._adam('item_emb', g['item_emb'], alpha[1], l2c, stream)
            self._adam('W1', g['W1'], alpha[4 + self.L], l2c, stream)
            self._adam('b1', g['b1'], alpha[5 + self.L], 0.0, stream)
        if want_loss:
            return float((gsw[-1] + reg_loss).item())
        return None

    # ---- inference: scores of ALL items for each (user, last-L-items) row (caser.py:128-137) --------------------
    def scores_all(self, uids, before):
        uid, bef = self._dev_i32(uids), self._dev_i32(before)
        B = uid.numel()
        cat = torch.zeros(B, self.ld2, dtype=torch.float32, device=self.device)
        A = self._args(uid, bef)
        A.cat_out = cat.data_ptr()
        check(lib().drx_caser_hidden(C.byref(self.D), C.byref(A), stream_ptr(self.device)), 'drx_caser_hidden')
        out = torch.empty(B, self.N, dtype=torch.float32, device=self.device)
        check(lib().drx_rows_dot(ptr(cat), B, ptr(self.W1), self.N, self.ld2, ptr(self.b1), ptr(out),
                                 stream_ptr(self.device)), 'drx_rows_dot')
        return out
