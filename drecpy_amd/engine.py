"""Device-side state of one CDAE model and the calls into libdrx.so.

Holds the parameter tables, optimizer slots, the positives-CSR of the training set and the scratch
arena as torch tensors on one MI355X (torch is the allocator/stream provider only) and forwards
every computation to the hand-written HIP kernels through the C ABI of include/drx.h.
"""
import ctypes as C
import math

import os
import time

import numpy as np
import torch

from . import _lib
from ._lib import Batch, CdaeParams, History, Optim, check, lib, ptr, stream_ptr

VAR_ORDER = ('W', 'W2T', 'V', 'b', 'b2')      # registration order W, W_, V, b, b_ (cdae.py:43)
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7  # tf.keras.optimizers.Adam defaults
ADAGRAD_INIT, ADAGRAD_EPS = 0.1, 1e-7         # tf.keras.optimizers.Adagrad defaults


def _round_up(x, m):
    return (x + m - 1) // m * m


class CdaeEngine:
    def __init__(self, n_users, n_items, k, device='cuda:0'):
        if not torch.cuda.is_available():
            raise _lib.DrxError('drecpy_amd needs a ROCm GPU (MI355X); there is no CPU fallback.')
        lib()
        self.device = torch.device(device)
        self.n_users, self.n_items, self.k = int(n_users), int(n_items), int(k)
        self.ld = _round_up(self.k, 4)
        z = dict(dtype=torch.float32, device=self.device)
        self.W = torch.zeros(self.n_items, self.ld, **z)
        self.W2T = torch.zeros(self.n_items, self.ld, **z)
        self.V = torch.zeros(self.n_users, self.ld, **z)
        self.b = torch.zeros(self.ld, **z)
        self.b2 = torch.zeros(self.n_items, **z)
        self._params = CdaeParams(self.n_users, self.n_items, self.k, self.ld, *[ptr(t) for t in self.tables()])
        self.opt = None
        self.s1 = self.s2 = None
        self.hist_indptr = self.hist_indices = None
        self._hist = None
        self._scratch = None
        self._dense_scratch, self._dense_scratch_B, self._dense_clean = None, None, False
        self._loss = torch.zeros(2, **z)

    # ---- parameters -------------------------------------------------------------------------
    def tables(self):
        return [self.W, self.W2T, self.V, self.b, self.b2]

    def init_glorot(self, seed):
        """GlorotUniform for all five variables, biases included (cdae.py:35-41; App. A.1).
        TF's generator cannot be reproduced, so the stream is numpy's PCG64 seeded with `seed`."""
        rng = np.random.default_rng(seed)

        def glorot(shape):
            fi, fo = (shape[0], shape[0]) if len(shape) == 1 else (shape[0], shape[1])
            lim = math.sqrt(6.0 / (fi + fo))
            return rng.uniform(-lim, lim, size=shape).astype(np.float32)
        self.set_params(W=glorot((self.n_items, self.k)), W_=glorot((self.k, self.n_items)),
                        V=glorot((self.n_users, self.k)), b=glorot((self.k,)), b_=glorot((self.n_items,)))

    def init_glorot_device(self, seed):
        """Same distribution as init_glorot, drawn on the GPU (benchmark-sized tables: no 5 GB host copy)."""
        gen = torch.Generator(device=self.device)
        gen.manual_seed(int(seed))
        k = self.k

        def fill(t, rows, fi, fo):
            lim = math.sqrt(6.0 / (fi + fo))
            t.zero_()
            view = t[:, :k] if t.dim() == 2 else t[:rows]
            view.copy_((torch.rand(view.shape, generator=gen, device=self.device, dtype=torch.float32) * 2 - 1) * lim)
        fill(self.W, self.n_items, self.n_items, k)
        fill(self.W2T, self.n_items, k, self.n_items)
        fill(self.V, self.n_users, self.n_users, k)
        fill(self.b, k, k, k)
        fill(self.b2, self.n_items, self.n_items, self.n_items)

    def set_params(self, W, W_, V, b, b_):
        """Weights in the reference's orientation: W [N,K], W_ [K,N], V [U,K], b [K], b_ [N]."""
        k = self.k
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device)
        self.W.zero_(); self.W2T.zero_(); self.V.zero_(); self.b.zero_()
        self.W[:, :k] = t(W)
        self.W2T[:, :k] = t(W_).t()
        self.V[:, :k] = t(V)
        self.b[:k] = t(b)
        self.b2.copy_(t(b_))

    def get_params(self):
        k = self.k
        c = lambda x: x.detach().cpu().numpy().copy()
        return {'W': c(self.W[:, :k]), 'W_': c(self.W2T[:, :k].t()), 'V': c(self.V[:, :k]), 'b': c(self.b[:k]),
                'b_': c(self.b2)}

    def snapshot(self):
        """Device copies of parameters and optimizer slots (used by _store_epoch_weights/_revert_weights)."""
        snap = {'p': [t.clone() for t in self.tables()]}
        if self.s1 is not None:
            snap['s1'] = [t.clone() for t in self.s1]
            snap['s2'] = [t.clone() for t in self.s2] if self.s2 is not None else None
        return snap

    def restore(self, snap, with_optimizer=False):
        for dst, src in zip(self.tables(), snap['p']):
            dst.copy_(src)
        if with_optimizer and 's1' in snap:
            for dst, src in zip(self.s1, snap['s1']):
                dst.copy_(src)
            if snap['s2'] is not None:
                for dst, src in zip(self.s2, snap['s2']):
                    dst.copy_(src)

    # ---- training-set positives (CSR) ---------------------------------------------------------
    def set_recorded_pairs(self, indptr, indices):
        """CSR (columns ascending) of EVERY (user, item) pair the training set records, whatever its value: the device PointSampler then
        draws its negatives among the pairs ABSENT from the frame, as the reference does (point_sampler.py:56), instead of among the
        non-positives.  Only needed where the frame records pairs below the interaction threshold; None = the positives."""
        if indptr is None:
            self._recorded = None
            return
        ip = torch.as_tensor(np.asarray(indptr, dtype=np.int64)).to(self.device)
        ix = torch.as_tensor(np.asarray(indices, dtype=np.int32)).to(self.device)
        if ix.numel() == 0:
            ix = torch.zeros(1, dtype=torch.int32, device=self.device)
        assert ip.numel() == self.n_users + 1
        self._recorded = (History(ptr(ip), ptr(ix)), ip, ix)

    TRANSPOSE_MAX_NNZ, TRANSPOSE_MAX_ROWS = 1 << 27, 1 << 22
    sample_by_user = True    # device-sampled batches in user order where the history's transpose exists (an attribute: A/B runs set it, bench.py --no-sample-by-user)
    share_users = True       # include/drx.h DRX_BATCH_SHARE_USERS: the triples of one user share their gather and their gradient (bench.py --no-share-users)

    def set_history(self, indptr, indices, with_transpose=True):
        self.hist_indptr = torch.as_tensor(np.asarray(indptr, dtype=np.int64)).to(self.device) \
            if not torch.is_tensor(indptr) else indptr.to(self.device, torch.int64)
        self.hist_indices = torch.as_tensor(np.asarray(indices, dtype=np.int32)).to(self.device) \
            if not torch.is_tensor(indices) else indices.to(self.device, torch.int32)
        if self.hist_indices.numel() == 0:
            self.hist_indices = torch.zeros(1, dtype=torch.int32, device=self.device)
        assert self.hist_indptr.numel() == self.n_users + 1
        self._hist = History(ptr(self.hist_indptr), ptr(self.hist_indices))
        self._hist_t = None
        nnz = int(self.hist_indptr[-1].item())
        if with_transpose and 0 < nnz <= self.TRANSPOSE_MAX_NNZ and 2 * self.n_items + self.n_users <= self.TRANSPOSE_MAX_ROWS:
            # the item-major rank of every history entry (the permutation that transposes the history, inverted): batches whose rows
            # collect long runs of touches (MovieLens shapes) are then prepared through this static order instead of sorting millions of
            # pairs per step (include/drx.h DrxHistory::t_rank).  Once per dataset, with torch ops on the device (set-up, not the hot path).
            ip, idx = self.hist_indptr, self.hist_indices[:nnz].long()
            rows = torch.repeat_interleave(torch.arange(self.n_users, device=self.device), (ip[1:] - ip[:-1]))
            pos = torch.arange(nnz, device=self.device) - ip[rows]
            order = torch.argsort(idx * self.n_users + rows)
            rank = torch.empty(nnz, dtype=torch.int32, device=self.device)
            rank[order] = torch.arange(nnz, dtype=torch.int32, device=self.device)
            self._hist_t = (rank, rows[order].to(torch.int32).contiguous(), pos[order].to(torch.int32).contiguous(),
                            idx[order].to(torch.int32).contiguous())
            self._hist = History(ptr(self.hist_indptr), ptr(self.hist_indices), ptr(rank), nnz, *[ptr(t) for t in self._hist_t[1:]])

    # ---- optimizer --------------------------------------------------------------------------
    def init_optimizer(self, kind, lr, reg_rate, beta1=ADAM_B1, beta2=ADAM_B2, eps=None, initial_accumulator=ADAGRAD_INIT):
        """kind: 'adam' (Keras Adam, reference default recommender_abc.py:153), 'adagrad', or — sampled mode only —
        'rowwise_adagrad' (one accumulator per table row: acc += mean_k(g^2); an engine extension, oracle/cdae_oracle.py).
        The hyper-parameters default to tf.keras's; drecpy_amd.optimizers objects passed to fit(optimizer=...) set them."""
        if kind not in ('adam', 'adagrad', 'rowwise_adagrad'):
            raise _lib.DrxError(f'unknown optimizer "{kind}" (adam, adagrad, rowwise_adagrad)')
        self.opt_kind = {'adam': _lib.OPT_ADAM, 'adagrad': _lib.OPT_ADAGRAD, 'rowwise_adagrad': _lib.OPT_ROWWISE_ADAGRAD}[kind]
        self.lr, self.reg_rate = float(lr), float(reg_rate)
        self.beta1, self.beta2 = float(beta1), float(beta2)
        self.opt_eps = float(eps) if eps is not None else (ADAM_EPS if kind == 'adam' else ADAGRAD_EPS)
        self._optim_struct = self._alpha_tab = None
        if kind == 'adam':
            self.s1 = [torch.zeros_like(t) for t in self.tables()]
            self.s2 = [torch.zeros_like(t) for t in self.tables()]
        else:
            self.s1 = [torch.full_like(t, float(initial_accumulator)) for t in self.tables()]
            self.s2 = None

    def _optim(self, alphas):
        # the struct is kept between steps (the slot tensors live as long as the optimizer): only the five lr_t change
        o = getattr(self, '_optim_struct', None)
        if o is None or self._optim_for is not self.s1:
            o = Optim()
            o.kind = self.opt_kind
            o.lr, o.reg_rate = self.lr, self.reg_rate
            o.beta1, o.beta2 = self.beta1, self.beta2
            o.eps = self.opt_eps
            for j in range(5):
                o.s1[j] = self.s1[j].data_ptr()
                o.s2[j] = self.s2[j].data_ptr() if self.s2 is not None else 0
            self._optim_struct, self._optim_for = o, self.s1
        o.alpha[0], o.alpha[1], o.alpha[2], o.alpha[3], o.alpha[4] = alphas
        return o

    @staticmethod
    def adam_alpha(lr, t, beta1=ADAM_B1, beta2=ADAM_B2):
        """Keras-Adam lr_t for the 1-based step t, in fp32 like optimizer_v2/adam.py."""
        f = np.float32
        return float(f(lr) * np.sqrt(f(1.0) - np.power(f(beta2), f(t))) / (f(1.0) - np.power(f(beta1), f(t))))

    _ALPHA_CHUNK = 512

    def _dense_alphas(self, step):
        """The five lr_t of dense step `step` (t = 5 * step + j + 1), from a table computed 512 steps at a time with the same
        fp32 arithmetic as adam_alpha (numpy scalar arithmetic per step costs more than launching the step)."""
        c = getattr(self, '_alpha_tab', None)
        base = step - step % self._ALPHA_CHUNK
        if c is None or c[0] != base or c[1] != self.lr:
            c = self._alpha_tab = (base, self.lr, self._alpha_table(base, self._ALPHA_CHUNK).tolist())
        return c[2][step - base]

    def _alpha_table(self, first_step, n_steps):
        """float32 [n_steps, 5]: lr_t of dense steps first_step .. first_step + n_steps - 1."""
        f = np.float32
        t = (5 * first_step + 1 + np.arange(5 * n_steps)).astype(np.float32)
        tab = f(self.lr) * np.sqrt(f(1.0) - np.power(f(self.beta2), t)) / (f(1.0) - np.power(f(self.beta1), t))
        return np.ascontiguousarray(tab.astype(np.float32).reshape(-1, 5))

    # ---- batches ----------------------------------------------------------------------------
    def _dev(self, a, dtype):
        if a is None:
            return None
        if torch.is_tensor(a):
            return a.to(self.device, dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a)).to(self.device, dtype)

    _STAGE_SLOTS = 4

    def _make_batch_staged(self, uid, iid, y, keep_off, keep, q, mask_seed, n_touch_slots):
        """Host (numpy) batch -> device in ONE asynchronous copy: the arrays are packed into a pinned staging buffer (a ring
        of _STAGE_SLOTS, so the host can run ahead of the device) and land in one device buffer that the Batch points into."""
        B = len(uid)
        parts = [('uid', np.ascontiguousarray(uid, dtype=np.int32)), ('keep_off', np.ascontiguousarray(keep_off, dtype=np.int32))]
        if iid is not None:
            parts.append(('iid', np.ascontiguousarray(iid, dtype=np.int32)))
        if y is not None:
            parts.append(('y', np.ascontiguousarray(y, dtype=np.float32)))
        if keep is not None:
            k8 = np.ascontiguousarray(keep, dtype=np.uint8)
            parts.append(('keep', k8 if k8.size else np.zeros(1, np.uint8)))
        offs, total = {}, 0
        for name, a in parts:
            offs[name] = total
            total += (a.nbytes + 15) & ~15
        st = self.__dict__.setdefault('_stage', {'i': 0, 'host': [None] * self._STAGE_SLOTS, 'ev': [None] * self._STAGE_SLOTS})
        k = st['i'] % self._STAGE_SLOTS
        st['i'] += 1
        if st['host'][k] is None or st['host'][k].numel() < total:
            st['host'][k] = torch.empty(int(total * 1.5) + 4096, dtype=torch.uint8, pin_memory=True)
            st['ev'][k] = None
        if st['ev'][k] is not None:
            st['ev'][k].synchronize()                 # the copy that last read this pinned slot (4 batches ago) has finished
        hv = st['host'][k].numpy()
        for name, a in parts:
            hv[offs[name]:offs[name] + a.nbytes] = a.view(np.uint8).reshape(-1)
        dev = torch.empty(total, dtype=torch.uint8, device=self.device)      # owned by the batch (the keep-alive tuple)
        dev.copy_(st['host'][k][:total], non_blocking=True)
        st['ev'][k] = torch.cuda.Event()
        st['ev'][k].record(torch.cuda.current_stream(self.device))
        base = dev.data_ptr()
        at = lambda name: (base + offs[name]) if name in offs else None
        bt = Batch(B, at('uid'), at('iid'), at('y'), at('keep_off'), at('keep'), int(mask_seed) & (2 ** 64 - 1), float(q),
                   int(n_touch_slots), self._batch_flags(n_touch_slots))
        return bt, (dev,)

    # ---- batches a producer thread writes straight into pinned memory (reference-mode fit()) -----------------------------
    def stage_acquire(self, B, keep_capacity):
        """A pinned staging slot for a user-only batch with explicit keep flags, laid out [uid int32 B | keep_off int32 B+1 |
        keep uint8 <= keep_capacity]: returns (slot, uid view, keep_off view, keep view) — numpy views a worker thread may fill
        (e.g. through drx_rng_corruption_keep) while the main thread does something else.  The step that last read the slot
        has finished when this returns."""
        off_ko = (4 * B + 15) & ~15
        off_kp = (off_ko + 4 * (B + 1) + 15) & ~15
        total = off_kp + max(int(keep_capacity), 1)
        st = self.__dict__.setdefault('_slots', {'i': 0, 'host': [None] * 8, 'ev': [None] * 8, 'busy': [False] * 8, 'views': [None] * 8})
        k = st['i'] % 8
        st['i'] += 1
        if st['host'][k] is None or st['host'][k].numel() < total:
            st['host'][k] = torch.empty(int(total * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
            st['ev'][k] = torch.cuda.Event()
            st['busy'][k] = False
            st['views'][k] = None
        if st['busy'][k]:
            st['ev'][k].synchronize()
            st['busy'][k] = False
        v = st['views'][k]
        if v is None or v[0] != (B, int(keep_capacity)):      # the views of a slot are rebuilt only when the batch shape changes
            hv = st['host'][k].numpy()
            extra = (np.empty(B, np.int32), np.empty(B, np.float64), np.empty(B, np.uint8))
            uid_v, ko_v, kp_v = hv[:4 * B].view(np.int32), hv[off_ko:off_ko + 4 * (B + 1)].view(np.int32), hv[off_kp:off_kp + max(int(keep_capacity), 1)]
            ptrs = (uid_v.ctypes.data, extra[0].ctypes.data, extra[1].ctypes.data, extra[2].ctypes.data, ko_v.ctypes.data, kp_v.ctypes.data,
                    len(kp_v))
            v = st['views'][k] = ((B, int(keep_capacity)), (k, off_ko, off_kp), uid_v, ko_v, kp_v, extra, ptrs)
        return v[1], v[2], v[3], v[4]

    def stage_extra(self, slot):
        """Per-slot host arrays (iid int32 [B], value float64 [B], is_negative uint8 [B]) for what else a draw produces."""
        return self._slots['views'][slot[0]][5]

    def stage_pointers(self, slot):
        """Addresses of the slot's arrays in the argument order of drx_drawahead_submit: uid, iid, value, is_negative, keep_off,
        keep, keep capacity (numpy's .ctypes accessor costs a microsecond per array and call)."""
        return self._slots['views'][slot[0]][6]

    def batch_in_slot(self, slot, B, n_keep, q):
        """Batch struct over a filled staging slot, WITHOUT a copy: pinned host memory is addressable from the device, and a
        batch of 64 users is a few KB that each workgroup of the gather kernel reads once — cheaper than an upload, which the
        runtime performs only once the stream has drained.  Call stage_release(slot) after queueing the step that reads it."""
        k, off_ko, off_kp = slot
        base = self._slots['host'][k].data_ptr()
        return Batch(B, base, None, None, base + off_ko, base + off_kp, 0, float(q), int(n_keep))

    def stage_release(self, slot):
        """The work queued so far on the current stream is the last to read the slot (stage_acquire waits for it)."""
        st = self._slots
        st['ev'][slot[0]].record(torch.cuda.current_stream(self.device))
        st['busy'][slot[0]] = True

    def _batch_flags(self, n_touch_slots):
        # DRX_BATCH_SHARE_USERS: takes effect only where a list is prepared through the history's transpose (MovieLens shapes)
        return _lib.BATCH_SHARE_USERS if (self.share_users and getattr(self, '_hist_t', None) is not None) else 0

    def make_batch(self, uid, iid=None, y=None, keep_off=None, keep=None, q=0.0, mask_seed=0, n_touch_slots=None):
        """Uploads (if needed) one batch and returns (Batch struct, keep-alive tensors)."""
        if (keep_off is not None and n_touch_slots is not None and self.device.type == 'cuda'
                and not any(torch.is_tensor(a) for a in (uid, iid, y, keep_off, keep))):
            return self._make_batch_staged(uid, iid, y, keep_off, keep, q, mask_seed, n_touch_slots)
        uid = self._dev(uid, torch.int32)
        B = int(uid.numel())
        if keep_off is None:
            keep_off = _lib.batch_offsets(self.hist_indptr, uid)
        else:
            keep_off = self._dev(keep_off, torch.int32)
        if n_touch_slots is None:
            n_touch_slots = int(keep_off[-1].item())
        iid = self._dev(iid, torch.int32)
        y = self._dev(y, torch.float32)
        keep = self._dev(keep, torch.uint8)
        if keep is not None and keep.numel() == 0:
            keep = torch.zeros(1, dtype=torch.uint8, device=self.device)
        bt = Batch(B, ptr(uid), ptr(iid), ptr(y), ptr(keep_off), ptr(keep), int(mask_seed) & (2 ** 64 - 1), float(q),
                   int(n_touch_slots), self._batch_flags(n_touch_slots))
        return bt, (uid, iid, y, keep_off, keep)

    def _ensure_scratch(self, B, n_touch_slots, dense=False):
        if dense:
            return self._ensure_dense_scratch(B)
        need = lib().drx_cdae_scratch_bytes(C.byref(self._params), B, n_touch_slots, 0)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = None
            self._scratch = torch.empty(int(need * 1.1) + 1024, dtype=torch.uint8, device=self.device)
        return self._scratch

    def _ensure_dense_scratch(self, B):
        """Scratch of the reference-mode step: ZERO-initialised, one buffer per batch size (a last partial batch or a second model
        configuration alternates with the usual size without re-allocating either).  The dense step leaves its batch-membership
        arrays clean (the kernel that reads an entry clears it), so consecutive steps pass DRX_DENSE_AUX_CLEAN and skip the
        memset; a buffer whose step raised (or faulted asynchronously: see dense_fault()) is zeroed again before its next use."""
        pool = self.__dict__.setdefault('_dense_pool', {})
        ent = pool.get(B)
        if ent is None:
            if len(pool) >= 4:                      # a handful of batch sizes at most: drop the oldest
                pool.pop(next(iter(pool)))
            need = lib().drx_cdae_scratch_bytes(C.byref(self._params), B, 0, 1)
            ent = pool[B] = [torch.zeros(int(need) + 1024, dtype=torch.uint8, device=self.device), True]
        elif not ent[1]:
            ent[0].zero_()
            ent[1] = True
        self._dense_scratch, self._dense_scratch_B, self._dense_clean = ent[0], B, True
        self._dense_ent = ent
        return ent[0]

    def dense_fault(self):
        """Call after catching a device fault: every dense scratch is zeroed before its next use (a kernel that died half-way may have
        left batch-membership bits set that the AUX_CLEAN contract promises are clear)."""
        for ent in self.__dict__.get('_dense_pool', {}).values():
            ent[1] = False

    # ---- calls ------------------------------------------------------------------------------
    def forward(self, uid, keep_off=None, keep=None, q=0.0, mask_seed=0, want_pred=True):
        """h [B,K], pred [B,N] for the given users (cdae.py:73-76).  q == 0 and keep is None gives the
        inference path of cdae.py:67-71 (uncorrupted, unscaled)."""
        bt, alive = self.make_batch(uid, keep_off=keep_off, keep=keep, q=q, mask_seed=mask_seed, n_touch_slots=0)
        h = torch.empty(bt.B, self.ld, dtype=torch.float32, device=self.device)
        pred = torch.empty(bt.B, self.n_items, dtype=torch.float32, device=self.device) if want_pred else None
        check(lib().drx_cdae_forward(C.byref(self._params), C.byref(self._hist), C.byref(bt), ptr(h), ptr(pred),
                                     stream_ptr(self.device)), 'drx_cdae_forward')
        return h[:, :self.k], pred

    def step_dense(self, step, bt, loss='bce', targets='reference', want_loss=False):
        """One reference-mode fit() iteration; `step` is the 0-based batch index (Adam t = 5*step+j+1)."""
        o = self._optim(self._dense_alphas(step))
        sc = self._ensure_dense_scratch(bt.B)
        self._dense_clean = self._dense_ent[1] = False       # (stays False if the call raises: the next step then starts from fresh zeros)
        check(lib().drx_cdae_step_dense(
            C.byref(self._params), C.byref(o), C.byref(self._hist), C.byref(bt),
            _lib.LOSS_BCE if loss == 'bce' else _lib.LOSS_MSE,
            (_lib.TARGETS_REFERENCE if targets == 'reference' else _lib.TARGETS_PER_ROW) | _lib.DENSE_AUX_CLEAN,
            ptr(sc), sc.numel(), ptr(self._loss) if want_loss else None, stream_ptr(self.device)),
            'drx_cdae_step_dense')
        self._dense_clean = self._dense_ent[1] = True
        return self._loss if want_loss else None

    _FIT_SLOTS = 16

    def fit_dense(self, drawahead, cursor, B, q, keep_capacity, first_step, n_steps, loss='bce', targets='reference'):
        """n_steps reference-mode iterations in ONE library call (drx_cdae_fit_dense): the draw-ahead workers of `drawahead` fill
        pinned staging slots, the library queues the steps.  cursor: int64[4] numpy array (sampler ticket, corruption-stream
        position, words consumed per generator), updated in place.  Returns when the steps have run."""
        L = lib()
        need = int(L.drx_cdae_fit_slot_bytes(B, int(keep_capacity)))
        slots = getattr(self, '_fit_slots', None)
        if slots is None or slots[0] < need:
            slots = self._fit_slots = (need, torch.empty(need * self._FIT_SLOTS, dtype=torch.uint8, pin_memory=True),
                                       torch.empty(2 * need, dtype=torch.uint8, device=self.device))
        o = self._optim([0.0] * 5)
        sc = self._ensure_dense_scratch(B)
        alphas = self._alpha_table(first_step, n_steps)
        self._dense_clean = self._dense_ent[1] = False
        check(L.drx_cdae_fit_dense(
            C.byref(self._params), C.byref(o), C.byref(self._hist), drawahead, cursor.ctypes.data, B, float(q), int(keep_capacity),
            _lib.LOSS_BCE if loss == 'bce' else _lib.LOSS_MSE, _lib.TARGETS_REFERENCE if targets == 'reference' else _lib.TARGETS_PER_ROW,
            n_steps, alphas.ctypes.data, slots[1].data_ptr(), slots[0], self._FIT_SLOTS, ptr(slots[2]), slots[2].numel(), ptr(sc),
            sc.numel(), stream_ptr(self.device)),
            'drx_cdae_fit_dense')
        self._dense_clean = self._dense_ent[1] = True

    def _retire(self, t):
        """A buffer about to be dropped while work on OTHER streams may still read it: its memory is not reused before every stream
        that exists on this device now has passed this point (the caching allocator only orders reuse on the allocating stream)."""
        t.record_stream(torch.cuda.default_stream(self.device))
        t.record_stream(torch.cuda.current_stream(self.device))

    def prep_buffer(self, bt, out=None):
        """A buffer large enough for the prepared touch list of `bt` (`out` itself when it is)."""
        need = lib().drx_cdae_prep_bytes(C.byref(self._params), bt.B, bt.n_touch_slots)
        if out is None or out.numel() < need:
            out = None
            out = torch.empty(int(need * 1.03) + 4096, dtype=torch.uint8, device=self.device)
        return out

    def prep_result_bytes(self, bt):
        """Leading bytes of a prepared buffer that a step reads (include/drx.h, drx_cdae_prep_result_bytes)."""
        return int(lib().drx_cdae_prep_result_bytes(C.byref(self._params), bt.B, bt.n_touch_slots))

    def prepare_sparse(self, bt, out=None):
        """Builds and sorts the touch list of a batch (drx_cdae_sparse_prepare) on the CURRENT stream.  The list does not
        depend on the parameters, so this may run on a side stream for batch t+1 while batch t trains."""
        out = self.prep_buffer(bt, out)
        check(lib().drx_cdae_sparse_prepare(C.byref(self._params), C.byref(self._hist), C.byref(bt), ptr(out), out.numel(),
                                            stream_ptr(self.device)), 'drx_cdae_sparse_prepare')
        return out

    def prepare_part(self, bt, part, parts, slot=None):
        """This rank's share of a batch's touch list (drx_cdae_sparse_prepare_part) on the CURRENT stream: a byte tensor that
        the caller gathers from all ranks and hands to prepare_assemble().  slot: reuse the (grow-only) buffer of that name —
        the result is then a view of it, valid until the slot's next use."""
        P = C.byref(self._params)
        nb = int(lib().drx_cdae_prep_part_out_bytes(P, bt.B, bt.n_touch_slots, parts))
        if slot is None:
            out = torch.empty(nb, dtype=torch.uint8, device=self.device)
        else:
            bufs = self.__dict__.setdefault('_part_bufs', {})
            if slot not in bufs or bufs[slot].numel() < nb:
                if bufs.get(slot) is not None:
                    self._retire(bufs[slot])
                bufs[slot] = None
                bufs[slot] = torch.empty(int(nb * 1.05) + 4096, dtype=torch.uint8, device=self.device)
            out = bufs[slot][:nb]
        need = lib().drx_cdae_prep_part_bytes(P, bt.B, bt.n_touch_slots, parts)
        if getattr(self, '_pscratch', None) is None or self._pscratch.numel() < need:
            if getattr(self, '_pscratch', None) is not None:
                self._retire(self._pscratch)
            self._pscratch = None
            self._pscratch = torch.empty(int(need * 1.1) + 1024, dtype=torch.uint8, device=self.device)
        check(lib().drx_cdae_sparse_prepare_part(P, C.byref(self._hist), C.byref(bt), part, parts, ptr(out), out.numel(),
                                                 ptr(self._pscratch), self._pscratch.numel(), stream_ptr(self.device)),
              'drx_cdae_sparse_prepare_part')
        return out

    def prepare_assemble(self, bt, all_parts, parts, out=None, overflow=None):
        """The parts of all ranks (rank order, contiguous) -> a prepared buffer like prepare_sparse()'s.  Returns (buffer,
        overflow flag tensor [1] int32 on the device: 1 = a part did not fit, the list is incomplete)."""
        need = lib().drx_cdae_prep_bytes(C.byref(self._params), bt.B, bt.n_touch_slots)
        if out is None or out.numel() < need:
            out = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        if overflow is None:
            overflow = torch.zeros(1, dtype=torch.int32, device=self.device)
        check(lib().drx_cdae_sparse_prepare_assemble(C.byref(self._params), C.byref(bt), ptr(all_parts), parts, ptr(out), out.numel(),
                                                     ptr(overflow), stream_ptr(self.device)), 'drx_cdae_sparse_prepare_assemble')
        return out, overflow

    def kshard_forward(self, bt, prepared=None):
        """Forward half of the column-sharded step on this engine's columns: (h [B, ld], partial dot products [B]).
        prepared: the batch's prepared list (its launch order of history lengths is used, like the single-GPU forward kernel)."""
        h = torch.empty(bt.B, self.ld, dtype=torch.float32, device=self.device)
        d = torch.empty(bt.B, dtype=torch.float32, device=self.device)
        check(lib().drx_cdae_kshard_forward_prepared(C.byref(self._params), C.byref(self._hist), C.byref(bt), ptr(prepared),
                                                     prepared.numel() if prepared is not None else 0, ptr(h), ptr(d),
                                                     stream_ptr(self.device)), 'drx_cdae_kshard_forward_prepared')
        return h, d

    def step_sparse(self, step, bt, loss='bce', want_loss=False, events=None, prepared=None, kshard=None):
        """One sampled-output step (sparse Adagrad / lazy Adam on touched rows).
        events: optional list of 6 recorded-once torch.cuda.Event(enable_timing=True); their raw hipEvent_t are
        re-recorded by the library around each phase (include/drx.h, drx_cdae_step_sparse_timed).
        prepared: buffer returned by prepare_sparse() for this batch (else the touch list is built inline)."""
        a = self.adam_alpha(self.lr, step + 1, self.beta1, self.beta2)
        o = self._optim([a] * 5)
        sc = self._ensure_scratch(bt.B, bt.n_touch_slots)
        lk = _lib.LOSS_BCE if loss == 'bce' else _lib.LOSS_MSE
        lo = ptr(self._loss) if want_loss else None
        arr = (C.c_void_p * len(events))(*[e.cuda_event for e in events]) if events is not None else None
        if kshard is not None:            # column-sharded step: (h, all-reduced dot products) replace the forward half
            h, dot_total = kshard
            check(lib().drx_cdae_kshard_step(C.byref(self._params), C.byref(o), C.byref(self._hist), C.byref(bt), lk, ptr(h),
                                             ptr(dot_total), ptr(prepared) if prepared is not None else None,
                                             prepared.numel() if prepared is not None else 0, ptr(sc), sc.numel(), lo, arr,
                                             stream_ptr(self.device)), 'drx_cdae_kshard_step')
            return self._loss if want_loss else None
        if prepared is not None:
            check(lib().drx_cdae_step_sparse_prepared(C.byref(self._params), C.byref(o), C.byref(self._hist), C.byref(bt),
                                                      lk, ptr(prepared), prepared.numel(), ptr(sc), sc.numel(), lo, arr,
                                                      stream_ptr(self.device)), 'drx_cdae_step_sparse_prepared')
        elif events is None:
            check(lib().drx_cdae_step_sparse(C.byref(self._params), C.byref(o), C.byref(self._hist), C.byref(bt), lk,
                                             ptr(sc), sc.numel(), lo, stream_ptr(self.device)), 'drx_cdae_step_sparse')
        else:
            check(lib().drx_cdae_step_sparse_timed(C.byref(self._params), C.byref(o), C.byref(self._hist), C.byref(bt),
                                                   lk, ptr(sc), sc.numel(), lo, arr, stream_ptr(self.device)),
                  'drx_cdae_step_sparse_timed')
        return self._loss if want_loss else None

    def sample_device(self, B, neg_ratio, seed, n_items=None, out=None, mailbox=None, tag=0):
        """Throughput-mode PointSampler on the GPU: returns device tensors (uid, iid, y, keep_off); `out` reuses them.
        mailbox: a pinned int64 tensor of one element that receives (tag << 32) | keep_off[B] straight from the last kernel."""
        if out is None:
            out = (torch.empty(B, dtype=torch.int32, device=self.device), torch.empty(B, dtype=torch.int32, device=self.device),
                   torch.empty(B, dtype=torch.float32, device=self.device), torch.empty(B + 1, dtype=torch.int32, device=self.device))
        uid, iid, y, keep_off = out
        # with the history's transpose (MovieLens shapes) the batch is handed out in user order: its touch list then has every row's
        # touches sample-ascending (include/drx.h drx_point_sample_by_user)
        by_user = getattr(self, '_hist_t', None) is not None and self.sample_by_user
        need = lib().drx_point_sample_by_user_scratch_bytes(B, self.n_users) if by_user else lib().drx_point_sample_scratch_bytes(B)
        # one scratch per stream: draws queued on different streams (a pipeline's run-ahead stream, a caller's own) run side by side
        pool = self.__dict__.setdefault('_sscratch', {})
        key = torch.cuda.current_stream(self.device).cuda_stream
        if pool.get(key) is None or pool[key].numel() < need:
            pool[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        scratch = pool[key]
        rec = getattr(self, '_recorded', None)
        fn = lib().drx_point_sample_by_user if by_user else lib().drx_point_sample_recorded
        check(fn(C.byref(self._hist), C.byref(rec[0]) if rec is not None else None, self.n_users,
                 n_items or self.n_items, B, neg_ratio, int(seed) & (2 ** 64 - 1), ptr(uid), ptr(iid), ptr(y),
                 ptr(keep_off), ptr(scratch), scratch.numel(), ptr(mailbox), int(tag) & 0xFFFFFFFF,
                 stream_ptr(self.device)),
              'drx_point_sample_by_user' if by_user else 'drx_point_sample_recorded')
        return out

    def topk(self, scores, k, cand_mask=None):
        """Row-wise top-k with heapq.nlargest((score, iid)) ordering (cdae.py:103)."""
        scores = scores.contiguous()
        R, n = scores.shape
        out_idx = torch.empty(R, k, dtype=torch.int32, device=self.device)
        out_val = torch.empty(R, k, dtype=torch.float32, device=self.device)
        sb = lib().drx_topk_scratch_bytes_k(R, n, k)
        sc = torch.empty(sb, dtype=torch.uint8, device=self.device) if sb else None
        check(lib().drx_topk(ptr(scores), ptr(cand_mask), R, n, k, ptr(out_idx), ptr(out_val), ptr(sc), sb,
                             stream_ptr(self.device)), 'drx_topk')
        return out_idx, out_val


def pack_mask_bits(mask_bool):
    """bool [R,n] (numpy) -> uint32 words; bit (i & 31) of word (i >> 5) is flat element i."""
    flat = np.asarray(mask_bool, dtype=bool).ravel()
    pad = (-len(flat)) % 32
    if pad:
        flat = np.concatenate([flat, np.zeros(pad, bool)])
    return np.packbits(flat, bitorder='little').view(np.uint32).copy()


class DeviceBatchSource:
    """batch_of(s) for pipelines that index batches ahead of time: step s's triples are drawn by the device PointSampler on a
    high-priority stream when batch s - AHEAD is requested, so that by the time batch s itself is asked for its touch count
    is already in pinned memory and nothing waits.  Batches must be requested in increasing order (as the pipelines do)."""

    AHEAD = 2

    def __init__(self, eng, batch_size, neg_ratio, q, sample_seed_of, mask_seed_of, n_items=None, slots=8, stream_slot=2):
        self.eng, self.B, self.neg_ratio, self.q = eng, int(batch_size), int(neg_ratio), float(q)
        self.sample_seed_of, self.mask_seed_of, self.n_items = sample_seed_of, mask_seed_of, n_items
        # stream_slot: which run-ahead stream of the process-wide pool draws (2: one of its own; the row layout passes 0 — the stream its
        # other run-ahead stages use: a rank there also has the communicator's stream and torch.distributed's, and a process's streams
        # share four hardware queues)
        self.stream = run_ahead_stream(eng.device, int(stream_slot))
        self.stream.wait_stream(torch.cuda.current_stream(eng.device))
        self.n_slots = slots                       # a slot is reused `slots` steps later: more than any pipeline looks ahead
        self.ring = [None] * slots
        self.count = [torch.empty(1, dtype=torch.int32, pin_memory=True) for _ in range(slots)]
        self.ready = [torch.cuda.Event() for _ in range(slots)]
        self.free = [None] * slots                 # recorded by release(): the step that used the slot's batch is queued
        self.made = {}
        self.drawn = -1

    def release(self, s):
        """The consumer has queued step s on the current stream: its batch's slot may be overwritten once that point is reached."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.eng.device))
        self.free[s % self.n_slots] = ev

    def _draw_up_to(self, s):
        while self.drawn < s:
            self.drawn += 1
            k = self.drawn % self.n_slots
            if self.free[k] is not None:
                self.stream.wait_event(self.free[k])
            with torch.cuda.stream(self.stream):
                self.ring[k] = self.eng.sample_device(self.B, self.neg_ratio, self.sample_seed_of(self.drawn), n_items=self.n_items,
                                                      out=self.ring[k])
                self.count[k].copy_(self.ring[k][3][-1:], non_blocking=True)
                self.ready[k].record(self.stream)

    def __call__(self, s):
        if s not in self.made:
            self._draw_up_to(s + self.AHEAD)
            k = s % self.n_slots
            self.ready[k].synchronize()             # drawn AHEAD requests ago
            uid, iid, y, keep_off = self.ring[k]
            torch.cuda.current_stream(self.eng.device).wait_event(self.ready[k])
            self.made[s] = self.eng.make_batch(uid, iid, y, keep_off=keep_off, q=self.q, mask_seed=self.mask_seed_of(s),
                                               n_touch_slots=int(self.count[k][0]))
            self.made.pop(s - self.n_slots + self.AHEAD + 1, None)
        return self.made[s][0]


class StreamEvent:
    """Ordering between two streams of one device without the host-visibility part of a torch.cuda.Event: created by
    drx_event_create (hipEventDisableTiming | hipEventDisableSystemFence), so a record is an agent-scope release instead of an L2
    write-back between two training kernels (measured: two torch event records per step cost 7.6 us, two of these 5)."""

    def __init__(self):
        self._h = lib().drx_event_create()
        if not self._h:
            raise _lib.DrxError('drx_event_create failed')

    def record(self, stream):
        check(lib().drx_event_record(self._h, C.c_void_p(stream.cuda_stream)), 'drx_event_record')

    def wait(self, stream):
        """`stream` waits for the work recorded by the last record()."""
        check(lib().drx_stream_wait_event(C.c_void_p(stream.cuda_stream), self._h), 'drx_stream_wait_event')

    def __del__(self):
        try:
            lib().drx_event_destroy(self._h)
        except Exception:
            pass


_RUN_AHEAD = {}
# CUs of every XCD the run-ahead streams are confined to by default (0: the whole chip, high priority).
PREP_CUS_PER_XCD = 0


def _ping_pong_us(main, side, rounds=40):
    """Microseconds per round of the run-ahead pattern between two streams: `side` waits for the training stream's last event, launches,
    records; the training stream waits for that, launches, records."""
    dev = side.device
    x, y = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    ev = torch.cuda.Event()
    ev.record(main)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(rounds):
        side.wait_event(ev)
        with _on_stream(side, main):
            x.add_(1)
        e2 = torch.cuda.Event()
        e2.record(side)
        main.wait_event(e2)
        y.add_(1)
        ev = torch.cuda.Event()
        ev.record(main)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / rounds * 1e6


# A run-ahead stream must not SHARE A HARDWARE QUEUE with the training stream: the runtime maps the streams a process uses onto a
# handful of queues, and a stream that lands on the training stream's queue turns every cross-stream event wait into head-of-line
# blocking — DMF.fit(device_sampler=True) at B = 256 ran at 0.33 instead of 0.11 ms/step for every second model of a process (r06,
# profiles/r06_stream_queues.log).  Such a stream still runs kernels beside the training stream; what gives it away is the
# round-trip time of the run-ahead pattern itself: 33 - 45 us per round on a queue of its own, 90 - 170 us on a shared one.
STREAM_PROBE_US = 70.0
STREAM_PROBE = True                 # (False: take the streams as they come — A/B, scripts/r06_stream_parity.py)


def _probed_stream(dev, main):
    """a high-priority stream whose run-ahead round trip with `main` is fast (the first of up to 6 candidates under STREAM_PROBE_US,
    else the fastest of them)"""
    best = None
    for _ in range(6 if STREAM_PROBE else 1):
        st = torch.cuda.Stream(dev, priority=-1)
        if not STREAM_PROBE:
            return st
        us = min(_ping_pong_us(main, st), _ping_pong_us(main, st))
        if us < STREAM_PROBE_US:
            return st
        if best is None or us < best[0]:
            best = (us, st)
    return best[1]


def _run_ahead_streams(dev, n, cus_per_xcd=0):
    """The n run-ahead streams of a device, created ONCE per process: every new torch stream is the next of a pool and the runtime
    spreads streams over a handful of hardware queues — the second pipeline of a process (bench.py's `configs` block after the headline
    run) got two streams that shared a queue and ran the ml-1m-shaped steps at 43 instead of 61 M triples/s (r03).
    cus_per_xcd == 0: high-priority torch streams over the whole chip; > 0: streams confined to that many CUs of every XCD
    (drx_stream_create_cu_slice: the preparation's launches queue for their slice instead of displacing training waves everywhere)."""
    return [_run_ahead_slot(dev, k, cus_per_xcd) for k in range(n)]


def _run_ahead_slot(dev, k, cus_per_xcd=0):
    """slot k of the pool — created (and probed) when first asked for, NOT together with the slots below it: every stream a process uses
    takes a place on one of a handful of hardware queues, and a row-sharded job that asked for slot 2 (its sampler) and thereby also
    brought slot 1 to life ended with five streams on four queues — the communicator's stream shared one, and the count exchanges waited
    for the training stream: 1.13 instead of 0.75 ms per step (r06cc)."""
    dev = torch.device(dev)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (dev.type, index, int(cus_per_xcd))
    have = _RUN_AHEAD.setdefault(key, {})
    if k not in have:
        if cus_per_xcd > 0:
            with torch.cuda.device(index):
                h = lib().drx_stream_create_cu_slice(int(cus_per_xcd))
            if not h:
                raise _lib.DrxError(f'drx_stream_create_cu_slice({cus_per_xcd}) failed')
            have[k] = torch.cuda.ExternalStream(h, device=torch.device('cuda', index))      # (lives as long as the process)
        else:
            have[k] = _probed_stream(torch.device('cuda', index), torch.cuda.current_stream(torch.device('cuda', index)))
    return have[k]


def run_ahead_stream(dev, k=0):
    """The k-th run-ahead stream of the process-wide pool (k = 0, 1: the preparation of the sampled step; 2: the device samplers' draws;
    3: deliveries).  Pipelines of different models share them: models train one at a time, and a process that uses few streams keeps
    every one of them on a hardware queue of its own."""
    return _run_ahead_slot(dev, k, 0)


class _on_stream:
    """`with torch.cuda.stream(side)` without most of its Python layers (25 - 30 us per block; two blocks per step of the pipelines
    below): torch.cuda.set_stream on the way in, and back to the stream that was current on the way out."""

    def __init__(self, side, home=None):
        self.side = side

    def __enter__(self):
        self.prev = torch.cuda.current_stream(self.side.device)
        torch.cuda.set_stream(self.side)
        return self

    def __exit__(self, *a):
        torch.cuda.set_stream(self.prev)
        return False


class SampledPipeline:
    """Keeps the training stream of the sampled-output mode free of everything that does not depend on the parameters.

    Step s trains on a batch the device PointSampler (drx_point_sample) drew two steps earlier and whose sorted touch
    list (drx_cdae_sparse_prepare) was built one step earlier, both on a high-priority side stream; the touch count of a
    batch reaches the host through pinned memory two steps before it is needed, so nothing waits.  Used by
    `CDAE.fit(mode='sampled', device_sampler=True)` and by bench.py — the same code path.

    sample_seed_of(s) / mask_seed_of(s): seeds of step s's triple draw and of its corruption mask.
    prep_ahead: how many steps ahead the touch list is prepared.  3 by default: the preparation shares the chip with the training
    kernels and then takes about as long as a step, so with one step of lead the training stream sometimes waited for it (measured at
    B = 65 536: 141 / 148 / 151 / 150 M triples/s at 1 / 2 / 3 / 4); more when preparing takes longer than a step, as when the ranks of a
    column-sharded job take turns preparing the list for all — dist.ColumnShardedCdae."""

    def __init__(self, eng, batch_size, neg_ratio, q, sample_seed_of, mask_seed_of, n_items=None, loss='bce', step_fn=None,
                 prepare_fn=None, prep_ahead=3, deliver_fn=None, side_streams=2, side_cus_per_xcd=None):
        self.eng, self.B, self.neg_ratio, self.q, self.loss = eng, int(batch_size), int(neg_ratio), float(q), loss
        # step_fn(s, bt, prepared, events, want_loss): what trains on a prepared batch (default: this engine's sparse step;
        # dist.ColumnShardedCdae.step for the column-sharded multi-GPU layout)
        self.step_fn = step_fn
        # prepare_fn(s, bt, out) -> prepared buffer, called with the side stream current (default: eng.prepare_sparse)
        self.prepare_fn = prepare_fn
        # deliver_fn(s, bt, out): second stage of a preparation, issued ONE step before the list is used on a stream of its own
        # (e.g. the broadcast of a list another rank built: issued late, so that no rank's collective waits for a sort)
        self.deliver_fn = deliver_fn
        self.comm = run_ahead_stream(eng.device, 3) if deliver_fn is not None else None
        self.sample_seed_of, self.mask_seed_of = sample_seed_of, mask_seed_of
        self.n_items = n_items
        dev = eng.device
        self.main = torch.cuda.current_stream(dev)
        # high priority: a normal stream may share a hardware queue with the training stream and inherit its barriers
        # TWO run-ahead streams, the work of step s on stream s % 2: a launch there mostly WAITS for room beside the training kernels
        # (DESIGN.md section 3), and the waits of two independent preparations overlap — step 0.355 -> 0.347 ms; a third stream gave
        # nothing (r03ah).  side_cus_per_xcd > 0: the streams are confined to a slice of the chip.
        self.sides = _run_ahead_streams(dev, max(1, int(side_streams)), PREP_CUS_PER_XCD if side_cus_per_xcd is None else int(side_cus_per_xcd))
        self.side = self.sides[0]
        # what the run-ahead work reads (histories, tables' shapes) may still be in flight on the caller's stream — a history generated on
        # the device a moment ago (scripts/stamps.py hit this: the sampler read row pointers that were not written yet and faulted)
        for st_ in self.sides:
            st_.wait_stream(self.main)
        if self.comm is not None:
            self.comm.wait_stream(self.main)
        self.D = D = max(1, int(prep_ahead))
        # batches are drawn one step before their list is prepared; two when lists are prepared far ahead (a draw then never
        # queues right behind a long preparation whose count the host is about to wait for)
        self.SA = D + 1 if D == 1 else D + 2
        self.RS, self.RP = self.SA + 1, D + 1                 # ring sizes: drawn batches, prepared lists
        # (allocated, not drawn: a draw queued HERE would run on the caller's stream beside the first run-ahead draws on the side
        # stream, sharing the sampler's scratch and these very tensors with them — r03: a batch whose row offsets belonged to another
        # draw than its users, seen as memory faults when a profiler stretched the window)
        B = self.B
        self.ring = [(torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
                      torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B + 1, dtype=torch.int32, device=dev))
                     for _ in range(self.RS)]
        # a batch's touch count reaches the host through a pinned mailbox the sampler's last kernel writes itself, tagged with the
        # step it belongs to (drx_point_sample): no copy kernel, no event, nothing for the host to synchronise with
        self.ring_T = [torch.full((1,), -1, dtype=torch.int64).pin_memory() for _ in range(self.RS)]
        self.ring_Tv = [t.numpy() for t in self.ring_T]
        self.ring_bt = [None] * self.RS
        self.prep = [None] * self.RP
        self.prep_done = [StreamEvent() for _ in range(self.RP)]
        self.built = [StreamEvent() for _ in range(self.RP)]
        # ONE record per step on the training stream: step j's event frees both its drawn-batch slot (for the draw of step j + RS)
        # and its prepared-list buffer (for the preparation of step j + RP)
        self.NE = max(self.RS, self.RP) + 1
        self.step_ev = [StreamEvent() for _ in range(self.NE)]
        for e in self.step_ev:
            e.record(self.main)
        self.next = 0
        for i in range(self.SA):
            self._sample(i)
        for i in range(D):
            self._prepare(i)
        if deliver_fn is not None:
            self._deliver(0)

    def _sample(self, s):
        k = s % self.RS
        if s >= self.RS:
            self.step_ev[(s - self.RS) % self.NE].wait(self.sides[s % len(self.sides)])   # the slot's previous batch (step s - RS) has been consumed
        with _on_stream(self.sides[s % len(self.sides)], self.main):
            self.eng.sample_device(self.B, self.neg_ratio, self.sample_seed_of(s), n_items=self.n_items, out=self.ring[k],
                                   mailbox=self.ring_T[k], tag=s + 1)

    def batch_of(self, s):
        k = s % self.RS
        if self.ring_bt[k] is None or self.ring_bt[k][0] != s:
            uid, iid, y, keep_off = self.ring[k]
            bt, alive = self.eng.make_batch(uid, iid, y, keep_off=keep_off, q=self.q, mask_seed=self.mask_seed_of(s),
                                            n_touch_slots=self._touch_count(k, s))
            self.ring_bt[k] = (s, bt, alive)
        return self.ring_bt[k][1]

    def _touch_count(self, k, s):
        """keep_off[B] of the batch drawn for step s, posted by the device at least one step ago: normally a plain read."""
        want, box = (s + 1) & 0xFFFFFFFF, self.ring_Tv[k]
        M = (1 << 64) - 1
        v = int(box[0]) & M
        if v >> 32 != want:
            import time
            t0 = time.perf_counter()
            while True:
                v = int(box[0]) & M
                if v >> 32 == want:
                    break
                if time.perf_counter() - t0 > 60.0:
                    raise _lib.DrxError(f'the device sampler did not deliver the batch of step {s} within 60 s')
        return v & 0xFFFFFFFF

    def _prepare(self, s):
        bt = self.batch_of(s)
        k = s % self.RP
        if s >= self.RP:
            self.step_ev[(s - self.RP) % self.NE].wait(self.sides[s % len(self.sides)])   # the buffer's previous user (step s - RP) has finished
        old = self.prep[k]
        with _on_stream(self.sides[s % len(self.sides)], self.main):
            if self.prepare_fn is not None:
                self.prep[k] = self.prepare_fn(s, bt, old)
            else:
                self.prep[k] = self.eng.prepare_sparse(bt, old)
        if old is not None and self.prep[k].data_ptr() != old.data_ptr():
            # the list grew into a new buffer.  The old one was allocated on the side stream, so the allocator would hand its memory to
            # the next side-stream allocation at once — but its last reader is a training step on the MAIN stream that, when the host
            # runs ahead of the device, has not even started (r03: a memory fault in scripts/stamps.py, whose first steps queue behind
            # a second of table initialisation).  record_stream defers the reuse until the main stream has passed this point.
            old.record_stream(self.main)
            if self.comm is not None:
                old.record_stream(self.comm)
        del old
        (self.prep_done if self.deliver_fn is None else self.built)[k].record(self.sides[s % len(self.sides)])

    def _deliver(self, s):
        bt = self.batch_of(s)
        k = s % self.RP
        self.built[k].wait(self.comm)
        with _on_stream(self.comm, self.main):
            self.deliver_fn(s, bt, self.prep[k])
        self.prep_done[k].record(self.comm)

    def run_step(self, events=None, want_loss=False):
        """Queues step `self.next` (and the run-ahead work of the following steps); returns the loss tensor or None."""
        s = self.next
        if self.deliver_fn is not None:
            self._deliver(s + 1)
        bt = self.batch_of(s)
        k = s % self.RP
        self.prep_done[k].wait(self.main)
        if self.step_fn is not None:
            out = self.step_fn(s, bt, self.prep[k], events, want_loss)
        else:
            out = self.eng.step_sparse(s, bt, self.loss, want_loss=want_loss, events=events, prepared=self.prep[k])
        self.step_ev[s % self.NE].record(self.main)
        # The run-ahead work of later steps is queued BEHIND the step itself: everything step s needs was queued iterations ago, and after
        # a host-side fence (an epoch callback, bench.py's timed windows) the training stream starts 50 - 60 us earlier (r06: windows of
        # 20 steps ran 0.8 % faster); in steady state the host is several steps ahead and the order does not matter.
        self._sample(s + self.SA)
        self._prepare(s + self.D)
        self.next = s + 1
        return out
