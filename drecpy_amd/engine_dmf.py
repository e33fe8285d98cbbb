"""Device-side state of one DMF model and the calls into libdrx.so (drx_dmf_*, drx_scatter_rows, drx_adam_*,
drx_score_pairs_bf16).

One training step = the body of the reference fit() loop for DMF (recommender_abc.py:190-204 over dmf.py:64-99):
  drx_dmf_fwd_bwd       both towers, cosine, clip, Keras BCE, backward -> dz0 row per distinct id + small grads
  drx_dmf_k0_update     first-layer kernel gradients (embedding-bag backward) + Keras l2 + dense Adam on both kernels in one walk
                        over the interaction matrix (user_nn t = 2s+1, item_nn t = 2s+2), deterministic
  drx_adam_segments     deeper kernels and all biases
(first_layer_update == 'scatter', for matrices too large to walk every step: touches from drx_dmf_fwd_bwd -> drx_scatter_rows x2
-> drx_adam_dense x2 instead of drx_dmf_k0_update)

`bind_prediction_scale(variable)` adds the ModifiedDMF extension of examples/extending_recommender_dmf.py:9-18 to the fused
step: one registered scalar multiplies every prediction, the loss becomes Keras' (B,B) broadcast (= BCE against the batch-mean
target) and the step makes three Adam applies, the scalar first (recommender_abc.py:194-196,328-334).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AdamSegments, DmfArgs, DmfDims, DmfK0Update, check, lib, ptr, stream_ptr
from .engine import ADAM_B1, ADAM_B2, ADAM_EPS, CdaeEngine, _round_up


class DmfEngine:
    def __init__(self, n_users, n_items, user_factors=(64, 32), item_factors=(64, 32), l2_norm_vectors=True, device='cuda:0'):
        if not torch.cuda.is_available():
            raise _lib.DrxError('drecpy_amd needs a ROCm GPU (MI355X); there is no CPU fallback.')
        lib()
        self.device = torch.device(device)
        self.U, self.N = n_users, n_items
        self.factors = [list(user_factors), list(item_factors)]
        if not all(1 <= len(f) <= 4 and 1 <= min(f) and max(f) <= 128 for f in self.factors):
            raise _lib.DrxError(f'DMF engine: towers of 1..4 layers of width 1..128 are supported (a lane of one wavefront holds hidden units '
                                f'k and k + 64), got user_factors={self.factors[0]}, item_factors={self.factors[1]}')
        # floats per row of the per-sample work rows and of the representations predict() hands out: 64 per unit slot of a lane
        self.W = 128 if max(max(f) for f in self.factors) > 64 else 64
        D = DmfDims()
        self.seg = []                      # (name, start, len, regularised, tower)
        off = 0
        for tw, pre in ((0, 'u'), (1, 'i')):
            f = self.factors[tw]
            D.n_layers[tw] = len(f)
            D.ld0[tw] = _round_up(f[0], 4)
            for l, fl in enumerate(f):
                D.f[tw][l] = fl
                if l >= 1:
                    D.off_k[tw][l] = off
                    self.seg.append((f'{pre}{l}_k', off, f[l - 1] * fl, True, tw)); off += _round_up(f[l - 1] * fl, 4)
                D.off_b[tw][l] = off
                self.seg.append((f'{pre}{l}_b', off, fl, False, tw)); off += _round_up(fl, 4)
        self._scale_slot = off                 # reserved for a registered prediction scale (bind_prediction_scale); unused: 1.0, no gradient
        off += 4
        D.n_small = off
        D.l2_norm_vectors = 1 if l2_norm_vectors else 0
        D.off_scale = -1
        self.D = D
        self.scale_var = None
        self.broadcast_targets = False
        self.beta1, self.beta2, self.eps = ADAM_B1, ADAM_B2, ADAM_EPS
        z = dict(dtype=torch.float32, device=self.device)
        self.K0u = torch.zeros(n_items, D.ld0[0], **z)
        self.K0i = torch.zeros(n_users, D.ld0[1], **z)
        self.sw = torch.zeros(off, **z)
        self.state = {n: (torch.zeros_like(t), torch.zeros_like(t)) for n, t in self.tensors().items()}
        nu, ni = self.K0u.numel(), self.K0i.numel()
        self._g_arena = torch.zeros(nu + ni, **z)            # both dense gradient tables: one buffer, one fill per step
        self._g = {'K0u': self._g_arena[:nu].view(self.K0u.shape), 'K0i': self._g_arena[nu:].view(self.K0i.shape)}
        self._scratch = None
        self.lr, self.reg = 1e-3, 1e-3
        # how the first-layer kernels are updated (set_interactions decides): 'scan' = drx_dmf_k0_update, one walk over all non-zeros
        # per step; 'scatter' = touches -> drx_scatter_rows -> drx_adam_dense, for matrices too large to walk every step
        self.first_layer_update = 'scan'
        self._maps = (torch.zeros(n_users, dtype=torch.int64, device=self.device), torch.zeros(n_items, dtype=torch.int64, device=self.device))
        self._stamp = 0
        from .Recommender.trainables import TrainableModel
        # what DMF._pre_fit registers (dmf.py:60): the two Sequential towers, each one apply_gradients per step
        self.user_nn = TrainableModel('user_nn', lambda: self._tower_weights(0))
        self.item_nn = TrainableModel('item_nn', lambda: self._tower_weights(1))

    def _tower_weights(self, tw):
        out = [(self.K0u, self.K0i)[tw][:, :self.factors[tw][0]]]
        for name, start, n, _, t in self.seg:
            if t == tw:
                out.append(self.sw[start:start + n])
        return out

    def bind_prediction_scale(self, variable, broadcast_targets=True):
        """Every prediction of the fused step (and of predict / score_matrix_bf16) is multiplied by `variable` (a registered scalar
        Variable), whose gradient and Adam update the step then also computes.  broadcast_targets: the loss is Keras' (B,B)
        broadcast of (B,) targets against (B,1) predictions — what ModifiedDMF's list of (1,)-tensors produces."""
        if variable.tensor.numel() != 1:
            raise _lib.DrxError('the prediction scale is one scalar')
        self.sw[self._scale_slot:self._scale_slot + 4].zero_()
        variable._rebind(self.sw[self._scale_slot:self._scale_slot + 1])
        variable._consumed_by = self
        self.D.off_scale = self._scale_slot
        self.scale_var = variable
        self.broadcast_targets = bool(broadcast_targets)
        self.state['sw'][0].zero_(); self.state['sw'][1].zero_()

    def tensors(self):
        return {'K0u': self.K0u, 'K0i': self.K0i, 'sw': self.sw}

    def set_interactions(self, csr, csc):
        """csr / csc = (indptr int64, indices, values) of the [U,N] interaction matrix and of its transpose, raw values."""
        d = self.device
        mk = lambda t: (torch.as_tensor(np.asarray(t[0], np.int64)).to(d), torch.as_tensor(np.asarray(t[1], np.int32)).to(d),
                        torch.as_tensor(np.asarray(t[2], np.float32)).to(d))
        self.csr, self.csc = mk(csr), mk(csc)
        self._h_indptr = (np.asarray(csr[0], np.int64).copy(), np.asarray(csc[0], np.int64).copy())   # host copies: batch offsets
        # (the scan update gives a quarter-wave to a first-layer gradient row: rows of up to 64 floats)
        self.first_layer_update = 'scan' if len(csr[1]) <= self.SCAN_MAX_NNZ and max(self.D.ld0[0], self.D.ld0[1]) <= 64 else 'scatter'
        # the l2 normaliser of every row / column depends on the dataset alone: once here, not per batch id and step
        self._rho = (torch.empty(self.U, dtype=torch.float32, device=d), torch.empty(self.N, dtype=torch.float32, device=d))
        for (ip, _, vals), n, out in ((self.csr, self.U, self._rho[0]), (self.csc, self.N, self._rho[1])):
            check(lib().drx_dmf_norms(C.byref(self.D), ptr(ip), ptr(vals), n, ptr(out), stream_ptr(self.device)), 'drx_dmf_norms')

        # drx_dmf_k0_update: row n of K0u (an item) walks COLUMN n of the matrix, row u of K0i (a user) walks ROW u; the longest walks
        # first (include/drx.h DrxDmfK0Update::row_order) — static per dataset
        walk = np.concatenate([np.diff(self._h_indptr[1]), np.diff(self._h_indptr[0])])
        # segment length of the gather's work items: 1024 non-zeros, more where the longest row / column would need over 255 segments
        self._seg_len = max(1024, -(-int(walk.max() if len(walk) else 0) // 255))
        self._k0_order = torch.as_tensor(np.argsort(-walk, kind='stable').astype(np.int32)).to(d)
        self._k0_long = int((walk > 1024).sum())            # (csrc/drx_dmf.hip kK0WaveMax: longer walks take a workgroup, the others a wave)

    SCAN_MAX_NNZ = 1 << 25      # 32 M non-zeros = 0.5 GB walked per step: beyond that the touches of a batch are the smaller job

    def set_params(self, p):
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device)
        keep_scale = float(self.sw[self._scale_slot].item()) if self.scale_var is not None else None
        for x in self.tensors().values():
            x.zero_()
        if self.scale_var is not None:        # ('extra_w' before a scale is bound is left to the binder: ModifiedDMF assigns it afterwards)
            self.sw[self._scale_slot] = float(np.asarray(p['extra_w']).reshape(-1)[0]) if 'extra_w' in p else keep_scale
        self.K0u[:, :self.factors[0][0]] = t(p['u0_k'])
        self.K0i[:, :self.factors[1][0]] = t(p['i0_k'])
        for name, start, n, _, _ in self.seg:
            self.sw[start:start + n] = t(p[name]).reshape(-1)

    def get_params(self):
        c = lambda x: x.detach().cpu().numpy().copy()
        p = {'u0_k': c(self.K0u[:, :self.factors[0][0]]), 'i0_k': c(self.K0i[:, :self.factors[1][0]])}
        for name, start, n, _, tw in self.seg:
            v = c(self.sw[start:start + n])
            if name.endswith('_k'):
                l = int(name[1])
                v = v.reshape(self.factors[tw][l - 1], self.factors[tw][l])
            p[name] = v
        if self.scale_var is not None:
            p['extra_w'] = c(self.sw[self._scale_slot:self._scale_slot + 1])
        return p

    def snapshot(self):
        return {'p': {n: t.clone() for n, t in self.tensors().items()}}

    def restore(self, snap, with_optimizer=False):
        for n, t in self.tensors().items():
            t.copy_(snap['p'][n])

    def _i32(self, a):
        if torch.is_tensor(a):
            return a.to(self.device, torch.int32).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)

    def _base_args(self, uid, iid):
        A = DmfArgs()
        A.K0u, A.K0i, A.sw = self.K0u.data_ptr(), self.K0i.data_ptr(), self.sw.data_ptr()
        A.u_indptr, A.u_indices, A.u_values = (t.data_ptr() for t in self.csr)
        A.i_indptr, A.i_indices, A.i_values = (t.data_ptr() for t in self.csc)
        A.uid, A.iid, A.B = uid.data_ptr(), iid.data_ptr(), int(uid.numel())
        return A

    def _offsets(self, ids, indptr):
        off = _lib.batch_offsets(indptr, ids if ids.dtype == torch.int32 else ids.to(torch.int32))
        return off, int(off[-1].item())

    def _scatter(self, keys, T, src, src_index, coef, ld, n_rows, out):
        """`out` must already be zero (step() clears the gradient arena once)."""
        if T == 0:
            return
        need = lib().drx_scatter_scratch_bytes(ld, T, n_rows)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(int(need * 1.2) + 1024, dtype=torch.uint8, device=self.device)
        check(lib().drx_scatter_rows(ptr(keys), T, ptr(src), ptr(src_index), ptr(coef), None, ld, n_rows, ptr(out), None,
                                     ptr(self._scratch), self._scratch.numel(), stream_ptr(self.device)), 'drx_scatter_rows')

    # what a step uploads, in this order, each array 16-byte aligned at a byte offset that depends on the batch size alone
    _BATCH_ARRAYS = ('du', 'di', 'y', 'off_u', 'off_i', 'inv_u', 'inv_i', 'gptr_u', 'gptr_i', 'grows_u', 'grows_i', 'order', 'zseg')
    _ORDER_EXTRA = 4096        # work items beyond one per distinct id (segments of long rows / columns): drx_dmf_work_order's order_cap

    @staticmethod
    def _batch_layout(B):
        lens = (B, B, B, B + 1, B + 1, B, B, B + 1, B + 1, B, B, 2 * B + DmfEngine._ORDER_EXTRA, 2 * B)
        offs, total = [], 0
        for n in lens:
            offs.append(total)
            total += (4 * n + 15) & ~15
        return offs, lens, total

    def prepare_batch(self, uids, iids, y):
        """Host half of a step, free of device work (DMF.fit() runs it on the sampler's worker thread): the distinct users / items of
        the batch (ascending), every sample's index among them, the samples per distinct id as a CSR (samples ascending: the order of
        the gradient sums) and the touch offsets of the distinct ids — drx_batch_distinct, counting through an id -> rank scratch
        instead of a sort — written into ONE host buffer at fixed offsets for ONE asynchronous upload."""
        B = len(uids)
        offs, lens, total = self._batch_layout(B)
        buf = np.empty(total, np.uint8)
        base = buf.ctypes.data
        at = dict(zip(self._BATCH_ARRAYS, offs))
        sc = self.__dict__.setdefault('_distinct_scratch', (np.full(self.U, -1, np.int32), np.full(self.N, -1, np.int32)))
        L_ = lib()
        nd = []
        for t, (ids, n, sfx) in enumerate(((uids, self.U, '_u'), (iids, self.N, '_i'))):
            ids32 = np.ascontiguousarray(ids, dtype=np.int32)
            r = L_.drx_batch_distinct(ids32.ctypes.data, B, n, self._h_indptr[t].ctypes.data, sc[t].ctypes.data, base + at['du' if t == 0 else 'di'],
                                      base + at['inv' + sfx], base + at['gptr' + sfx], base + at['grows' + sfx], base + at['off' + sfx])
            if r < 0:
                check(int(r), 'drx_batch_distinct')
            nd.append(int(r))
        y32 = buf[at['y']:at['y'] + 4 * B].view(np.float32)
        y32[:] = y
        off_u = buf[at['off_u']:at['off_u'] + 4 * (nd[0] + 1)].view(np.int32)
        off_i = buf[at['off_i']:at['off_i'] + 4 * (nd[1] + 1)].view(np.int32)
        # the gather's work items, LONGEST row / column first (include/drx.h DrxDmfArgs::work_order): a popular item's column has
        # thousands of non-zeros and is the launch's critical path when its workgroup happens to start late
        # the gather's work items: longest rows / columns first, long ones cut into segments of _seg_len non-zeros (include/drx.h
        # DrxDmfArgs::work_order; uncut when the segments would not fit the list: rare, correct all the same)
        n_part = C.c_int32(0)
        seg = self._seg_len
        n_work = int(L_.drx_dmf_work_order(base + at['off_u'], nd[0], base + at['off_i'], nd[1], seg, base + at['order'],
                                           2 * B + self._ORDER_EXTRA, base + at['zseg'], C.byref(n_part)))
        if n_work < 0:
            seg = 0
            n_work = int(L_.drx_dmf_work_order(base + at['off_u'], nd[0], base + at['off_i'], nd[1], 0, base + at['order'],
                                               2 * B + self._ORDER_EXTRA, base + at['zseg'], C.byref(n_part)))
            if n_work < 0:
                check(n_work, 'drx_dmf_work_order')
        return {'buf': buf, 'offs': offs, 'B': B, 'n_du': nd[0], 'n_di': nd[1], 'Tu': int(off_u[-1]), 'Ti': int(off_i[-1]),
                'n_work': n_work, 'seg_len': seg, 'n_part': int(n_part.value),
                'y_mean': float(y32.astype(np.float64).mean())}

    # ---- batches drawn and prepared ON THE DEVICE (DMF.fit(device_sampler=True): throughput mode) --------------------------------
    def set_sampler_frame(self, positives, recorded, vmin, vrange):
        """positives = (indptr, indices, values) of the pairs a positive draw may return (interaction >= threshold) with their
        values; recorded = (indptr, indices) of every pair the frame holds (a negative is a pair absent from it) or None when all
        recorded pairs are positives; vmin / vrange: the standardisation of recommender_abc.py:463-465 (vrange 0 = raw values)."""
        d = self.device
        ip, idx, val = positives
        self._pos = (torch.as_tensor(np.asarray(ip, np.int64)).to(d), torch.as_tensor(np.asarray(idx, np.int32)).to(d),
                     torch.as_tensor(np.asarray(val, np.float32)).to(d))
        self._rec = None
        if recorded is not None:
            self._rec = (torch.as_tensor(np.asarray(recorded[0], np.int64)).to(d), torch.as_tensor(np.asarray(recorded[1], np.int32)).to(d))
        self._vstd = (float(vmin), float(vrange))
        self._dev_ring, self._dev_i = {}, 0

    # host-prepared batches: the argument structs of a step cached per slot of the upload ring (False: built anew every step — A/B, tests)
    host_step_cache = True

    # device-prepared batches: the gather's work list (longest rows first, long ones cut into segments) built on the device too — False:
    # one work item per distinct id in their own order, as through r05 (52.6 against 28 us for the gather at B = 4096)
    device_work_order = True

    def _device_seg_len(self):
        """segment length of device-built work lists: the host's, raised until the list cannot overflow its _ORDER_EXTRA spare entries —
        every distinct id brings its own row or column once, so the extra entries are at most 2 nnz / seg_len"""
        nnz2 = int(self.csr[1].numel()) + int(self.csc[1].numel())
        return max(int(self._seg_len), -(-nnz2 // self._ORDER_EXTRA))

    def _ensure_dev_ring(self):
        if not hasattr(self, '_dev_ring'):
            self._dev_ring, self._dev_i = {}, 0

    def prepare_batch_device(self, B, neg_ratio, seed, triples=None):
        """One batch drawn by the device PointSampler (drx_point_sample_valued: the reference sampler's distribution, a counter-based
        stream) and prepared there (drx_dmf_batch_distinct_device) on the CURRENT stream — DMF.fit() runs it one step ahead on a side
        stream.  Returns what step() takes in place of prepare_batch()'s host arrays.  A ring of 3 buffers per batch size: the
        caller keeps at most two batches in flight.  triples: (uid, iid, y) device tensors to prepare INSTEAD of drawing (tests)."""
        from ._lib import History
        L_ = lib()
        self._ensure_dev_ring()
        k = self._dev_i % 3
        self._dev_i += 1
        slot = self._dev_ring.get((B, k))
        if slot is None:
            i32 = dict(dtype=torch.int32, device=self.device)
            need = int(L_.drx_dmf_distinct_scratch_bytes(B, self.U, self.N))
            slot = self._dev_ring[(B, k)] = {
                'ids': torch.empty(2, B, **i32), 'y': torch.empty(B, dtype=torch.float32, device=self.device),
                'arr': torch.empty(8, B + 4, **i32),          # du, di, inv_u, inv_i, gptr_u, gptr_i, grows_u, grows_i
                'nd': torch.empty(2, **i32), 'y_mean': torch.empty(1, dtype=torch.float32, device=self.device),
                'scratch': torch.empty(need, dtype=torch.uint8, device=self.device),
                # the gather's work list built on the device (drx_dmf_work_order_device): entries, zseg per work index, {entries, partial rows}
                'order': torch.empty(2 * B + self._ORDER_EXTRA, **i32), 'zseg': torch.empty(2 * B, **i32), 'nw': torch.zeros(2, **i32)}
        st = stream_ptr(self.device)
        uid, iid = slot['ids'][0], slot['ids'][1]
        # everything but the seed and the stream is the same from call to call on a slot: the arguments are built once (at B = 256 the
        # host, not the device, bounds fit(device_sampler=True): r06, profiles/r06_dmf_host_profile_B256.txt)
        call = slot.get('call')
        pos, rec = getattr(self, '_pos', None), getattr(self, '_rec', None)          # (set_sampler_frame; absent when only `triples` are prepared)
        if call is None or call['pos'] is not pos or call['rec'] is not rec:
            a = slot['arr']
            call = slot['call'] = {
                'pos': pos, 'rec': rec, 'draw_out': (ptr(uid), ptr(iid), ptr(slot['y'])),
                'distinct': (ptr(uid), ptr(iid), ptr(slot['y']), B, self.U, self.N, *[ptr(a[r]) for r in range(8)], ptr(slot['nd']),
                             ptr(slot['y_mean']), ptr(slot['scratch']), slot['scratch'].numel())}
            if pos is not None:
                H = History(ptr(pos[0]), ptr(pos[1]))
                R = History(ptr(rec[0]), ptr(rec[1])) if rec is not None else None
                call.update({'H': H, 'R': R, 'draw': (C.byref(H), C.byref(R) if R is not None else None, ptr(pos[2]), self._vstd[0], self._vstd[1],
                                                      self.U, self.N, B)})
        if triples is not None:
            uid.copy_(triples[0]); iid.copy_(triples[1]); slot['y'].copy_(triples[2])
        else:
            check(L_.drx_point_sample_valued(*call['draw'], int(neg_ratio), int(seed) & ((1 << 64) - 1), *call['draw_out'], st),
                  'drx_point_sample_valued')
        check(L_.drx_dmf_batch_distinct_device(*call['distinct'], st), 'drx_dmf_batch_distinct_device')
        if self.device_work_order:
            wo = call.get('order')
            if wo is None or wo[0] != self.csr[0].data_ptr():
                a = slot['arr']
                wo = call['order'] = (self.csr[0].data_ptr(), (ptr(self.csr[0]), ptr(self.csc[0]), ptr(a[0]), ptr(a[1]), ptr(slot['nd']),
                                                                  self._device_seg_len(), ptr(slot['order']), slot['order'].numel(), ptr(slot['zseg']),
                                                                  ptr(slot['nw'])))
            check(L_.drx_dmf_work_order_device(*wo[1], st), 'drx_dmf_work_order_device')
        return {'device': slot, 'B': B}

    def _upload_batch(self, prepared):
        """Host batch -> device in one asynchronous copy from a pinned ring (so the host can run ahead of the device); returns the
        device buffer and the addresses of the arrays in it."""
        st = self.__dict__.setdefault('_stage', {'i': 0, 'host': [None] * 4, 'ev': [None] * 4, 'dev': [None] * 4, 'cache': [None] * 4})
        buf = prepared['buf']
        total = buf.nbytes
        k = st['i'] % 4
        st['i'] += 1
        if st['host'][k] is None or st['host'][k].numel() < total:
            st['host'][k] = torch.empty(int(total * 1.5) + 4096, dtype=torch.uint8, pin_memory=True)
            st['ev'][k] = torch.cuda.Event()
            # the device side of the slot is kept too (r06): steps run in order on one stream, so a copy into it is queued behind the
            # step that last read it — and a slot's addresses stay the same from visit to visit (the step's argument structs are cached)
            st['dev'][k] = torch.empty(st['host'][k].numel(), dtype=torch.uint8, device=self.device)
            st['cache'][k] = None
        else:
            st['ev'][k].synchronize()                 # the copy that last read this pinned slot has finished
        st['host'][k].numpy()[:total] = buf
        dev = st['dev'][k]
        dev[:total].copy_(st['host'][k][:total], non_blocking=True)
        st['ev'][k].record(torch.cuda.current_stream(self.device))
        base = dev.data_ptr()
        prepared['stage_slot'] = k
        return dev, [base + o for o in prepared['offs']]

    def step(self, step_idx, uids, iids=None, y=None, want_loss=False, applies=None):
        """uids, iids, y: the batch (arrays or tensors) — or `uids` = what prepare_batch returned for it.
        applies = (n, j_user, j_item, j_scale): number of apply_gradients calls per step and the positions of user_nn, item_nn
        and the prediction scale among them (the registration order, recommender_abc.py:194-196); default: the reference DMF's
        (2, 0, 1), or (3, 1, 2, 0) with a bound scale."""
        L_ = lib()
        if isinstance(uids, dict) and uids.get('device') is not None and not want_loss:
            done = self._step_device_cached(L_, step_idx, uids, applies)
            if done:
                return None
        z = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        if not isinstance(uids, dict):
            if torch.is_tensor(uids) or torch.is_tensor(iids) or torch.is_tensor(y):
                c = lambda a: a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
                uids, iids, y = c(uids), c(iids), c(y)
            uids = self.prepare_batch(uids, iids, y)
        prep = uids
        on_device = prep.get('device') is not None
        if on_device:                       # drawn and prepared on the device: the distinct counts and the batch mean live there
            if self.first_layer_update != 'scan':
                raise _lib.DrxError('device-prepared DMF batches need the scan update of the first-layer kernels (set_interactions chose '
                                    '"scatter": more than SCAN_MAX_NNZ interactions)')
            sl = prep['device']
            a = sl['arr']
            p_du, p_di, p_invu, p_invi, p_gpu, p_gpi, p_gru, p_gri = (a[r].data_ptr() for r in range(8))
            p_y, p_offu, p_offi = sl['y'].data_ptr(), 0, 0
            B, Tu, Ti = prep['B'], 0, 0
            n_du = n_di = B                 # upper bounds: they size the launches; the kernels read nd_dev
        else:
            alive, ptrs = self._upload_batch(prep)
            if not want_loss and self.first_layer_update == 'scan' and self.host_step_cache:
                c = self._stage['cache'][prep['stage_slot']]
                if applies is None:
                    applies = (3, 1, 2, 0) if self.scale_var is not None else (2, 0, 1, None)
                if c is not None and c['B'] == prep['B'] and c['base'] == alive.data_ptr() and c['applies'] == tuple(applies) \
                        and c['key'] == self._step_cache_key():
                    A = c['A']                              # the slot's struct with this batch's scalars
                    A.n_du, A.n_di, A.n_work, A.seg_len = prep['n_du'], prep['n_di'], prep['n_work'], prep['seg_len']
                    if A.target_mode == 1:
                        A.y_mean = prep['y_mean']
                    self._launch_cached(L_, c, step_idx, tuple(applies))
                    return None
                self._stage['cache_wanted'] = (prep['stage_slot'], alive.data_ptr())
            p_du, p_di, p_y, p_offu, p_offi, p_invu, p_invi, p_gpu, p_gpi, p_gru, p_gri, p_order, p_zseg = ptrs
            B, Tu, Ti = prep['B'], prep['Tu'], prep['Ti']
            n_du, n_di = prep['n_du'], prep['n_di']
        ld0u, ld0i = self.D.ld0[0], self.D.ld0[1]
        stream = stream_ptr(self.device)
        grid = L_.drx_dmf_grid(B)
        # device work buffers of a batch size: steps run in order on one stream, so they are reused from step to step
        wk = getattr(self, '_step_bufs', None)
        if wk is None or wk[0] != B:
            wk = self._step_bufs = (B, torch.empty(B, ld0u, **z), torch.empty(B, ld0i, **z), torch.empty(grid, self.D.n_small, **z),
                                    torch.empty(grid, **z), torch.empty(self.D.n_small + 1, **z))
        _, dz0u, dz0i, gpart, lpart, gsw = wk
        scan = self.first_layer_update == 'scan'
        if not scan:
            tk_u, ts_u, tc_u = torch.empty(max(Tu, 1), **i32), torch.empty(max(Tu, 1), **i32), torch.empty(max(Tu, 1), **z)
            tk_i, ts_i, tc_i = torch.empty(max(Ti, 1), **i32), torch.empty(max(Ti, 1), **i32), torch.empty(max(Ti, 1), **z)
        A = DmfArgs()
        A.K0u, A.K0i, A.sw = self.K0u.data_ptr(), self.K0i.data_ptr(), self.sw.data_ptr()
        A.u_indptr, A.u_indices, A.u_values = (t.data_ptr() for t in self.csr)
        A.i_indptr, A.i_indices, A.i_values = (t.data_ptr() for t in self.csc)
        A.uid, A.iid, A.B = p_du, p_di, int(B)
        A.y, A.off_u, A.off_i = p_y, p_offu, p_offi
        A.inv_u, A.inv_i, A.gptr_u, A.gptr_i, A.grows_u, A.grows_i = p_invu, p_invi, p_gpu, p_gpi, p_gru, p_gri
        A.n_du, A.n_di = n_du, n_di
        if not on_device:
            A.work_order, A.n_work, A.seg_len, A.zseg = p_order, prep['n_work'], prep['seg_len'], p_zseg
            zp = getattr(self, '_zpart', None)
            if zp is None:
                zp = self._zpart = torch.empty(self._ORDER_EXTRA + 8, self.W, dtype=torch.float32, device=self.device)
            A.zpart = zp.data_ptr()
        if on_device:
            A.nd_dev, A.y_mean_dev = prep['device']['nd'].data_ptr(), prep['device']['y_mean'].data_ptr()
            if self.device_work_order and 'order' in prep['device']:
                sl_ = prep['device']
                zp = getattr(self, '_zpart', None)
                if zp is None:
                    zp = self._zpart = torch.empty(self._ORDER_EXTRA + 8, self.W, dtype=torch.float32, device=self.device)
                A.work_order, A.n_work, A.seg_len = sl_['order'].data_ptr(), sl_['order'].numel(), self._device_seg_len()
                A.zseg, A.zpart, A.n_work_dev = sl_['zseg'].data_ptr(), zp.data_ptr(), sl_['nw'].data_ptr()
        if self.broadcast_targets and self.scale_var is not None:
            A.target_mode = 1
            A.y_mean = 0.0 if on_device else prep['y_mean']
        A.dz0u, A.dz0i = dz0u.data_ptr(), dz0i.data_ptr()
        A.rho_u, A.rho_i = self._rho[0].data_ptr(), self._rho[1].data_ptr()
        if scan:
            self._stamp += 1
            A.map_u, A.map_i, A.stamp = self._maps[0].data_ptr(), self._maps[1].data_ptr(), self._stamp
        else:
            A.tkeys_u, A.tsrc_u, A.tcoef_u = tk_u.data_ptr(), ts_u.data_ptr(), tc_u.data_ptr()
            A.tkeys_i, A.tsrc_i, A.tcoef_i = tk_i.data_ptr(), ts_i.data_ptr(), tc_i.data_ptr()
        A.gsw_part, A.loss_part = gpart.data_ptr(), lpart.data_ptr()
        need = L_.drx_dmf_work_bytes(B)
        if getattr(self, '_work', None) is None or self._work.numel() < need:
            self._work = torch.empty(int(need) + 1024, dtype=torch.uint8, device=self.device)
        A.work = self._work.data_ptr()
        reg_loss = None
        if want_loss:
            sq = _lib.sumsq([self.K0u, self.K0i] + [self.sw[start:start + n] for _, start, n, regd, _ in self.seg if regd])
            reg_loss = self.reg * sq
        if applies is None:
            applies = (3, 1, 2, 0) if self.scale_var is not None else (2, 0, 1, None)
        n_app = applies[0]
        alpha = [CdaeEngine.adam_alpha(self.lr, n_app * step_idx + j + 1, self.beta1, self.beta2) if j is not None else 0.0 for j in applies[1:]]
        l2c = 2.0 * self.reg
        sg = AdamSegments()
        sg.n = len(self.seg)
        for i, (_, start, n, regd, tw) in enumerate(self.seg):
            sg.start[i], sg.len[i], sg.alpha[i], sg.l2_coef[i] = start, n, alpha[tw], (l2c if regd else 0.0)
        if self.scale_var is not None:                     # the registered scalar: its own lr_t, no regulariser
            i = sg.n
            sg.n += 1
            sg.start[i], sg.len[i], sg.alpha[i], sg.l2_coef[i] = self._scale_slot, 1, alpha[2], 0.0
        m, v = self.state['sw']
        # forward / backward with the small weights' Keras Adam in the launch that sums their partial gradients (r06: drx_dmf_step_small
        # — one launch less than drx_dmf_fwd_bwd + drx_adam_segments; gsw still holds the gradient and, last, the loss sum; the
        # first-layer update below reads none of the small weights)
        check(L_.drx_dmf_step_small(C.byref(self.D), C.byref(A), ptr(gsw), ptr(self.sw), ptr(m), ptr(v), C.byref(sg), self.beta1, self.beta2,
                                    self.eps, stream), 'drx_dmf_step_small')
        if scan:
            up = DmfK0Update()
            (mu, vu), (mi, vi) = self.state['K0u'], self.state['K0i']
            up.K0u, up.m_u, up.v_u, up.K0i, up.m_i, up.v_i = (t.data_ptr() for t in (self.K0u, mu, vu, self.K0i, mi, vi))
            up.n_items, up.n_users = self.N, self.U
            up.row_order, up.n_long = self._k0_order.data_ptr(), self._k0_long
            up.alpha_u, up.alpha_i, up.l2_coef, up.beta1, up.beta2, up.eps = alpha[0], alpha[1], l2c, self.beta1, self.beta2, self.eps
            check(L_.drx_dmf_k0_update(C.byref(self.D), C.byref(A), C.byref(up), stream), 'drx_dmf_k0_update')
        else:
            self._g_arena.zero_()
            self._scatter(tk_u, Tu, dz0u, ts_u, tc_u, ld0u, self.N, self._g['K0u'])
            self._scatter(tk_i, Ti, dz0i, ts_i, tc_i, ld0i, self.U, self._g['K0i'])
            for name, tw in (('K0u', 0), ('K0i', 1)):
                p = self.tensors()[name]
                m, v = self.state[name]
                check(L_.drx_adam_dense(ptr(p), ptr(m), ptr(v), ptr(self._g[name]), p.numel(), alpha[tw], l2c, self.beta1, self.beta2,
                                        self.eps, stream), 'drx_adam_dense')
        wanted_host = None if on_device else self.__dict__.get('_stage', {}).pop('cache_wanted', None)
        if (not on_device) and scan and wanted_host is not None and not want_loss:
            sg_tw = [(i, tw) for i, (_, _, _, _, tw) in enumerate(self.seg)] + ([(len(self.seg), 2)] if self.scale_var is not None else [])
            self._stage['cache'][wanted_host[0]] = {
                'B': B, 'base': wanted_host[1], 'applies': tuple(applies), 'key': self._step_cache_key(), 'A': A, 'up': up, 'sg': sg, 'gsw': ptr(gsw),
                'D': C.byref(self.D), 'Aref': C.byref(A), 'upref': C.byref(up), 'sgref': C.byref(sg), 'sg_tw': sg_tw, 'sw': ptr(self.sw),
                'm': ptr(m), 'v': ptr(v), 'keep': (gsw, gpart, lpart, dz0u, dz0i)}
        if on_device and scan and prep['device'].pop('step_cache_wanted', False):
            sg_tw = [(i, tw) for i, (_, _, _, _, tw) in enumerate(self.seg)] + ([(len(self.seg), 2)] if self.scale_var is not None else [])
            prep['device']['step_cache'] = {
                'applies': tuple(applies), 'key': self._step_cache_key(), 'A': A, 'up': up, 'sg': sg, 'gsw': ptr(gsw), 'D': C.byref(self.D),
                'Aref': C.byref(A), 'upref': C.byref(up), 'sgref': C.byref(sg), 'sg_tw': sg_tw, 'sw': ptr(self.sw), 'm': ptr(m), 'v': ptr(v),
                'keep': (gsw, gpart, lpart, dz0u, dz0i)}
        if want_loss:
            return float((gsw[-1] + reg_loss).item())
        return None

    def _step_device_cached(self, L_, step_idx, prep, applies):
        """step() for a device-prepared batch whose argument structs were built by an earlier step on the same ring slot: only the stamp
        and the learning rates change from step to step (fit(device_sampler=True) at B = 256 is bound by the host's Python, not by the
        75 us of kernels).  False when there is no valid cache for the slot: the caller takes the general path, which fills it."""
        sl = prep['device']
        c = sl.get('step_cache')
        if applies is None:
            applies = (3, 1, 2, 0) if self.scale_var is not None else (2, 0, 1, None)
        if c is None or c['applies'] != applies or c['key'] != self._step_cache_key():
            sl['step_cache_wanted'] = True
            return False
        self._launch_cached(L_, c, step_idx, applies)
        return True

    def _launch_cached(self, L_, c, step_idx, applies):
        """the three library calls of a scan-update step from cached argument structs: stamp and learning rates refreshed"""
        A, up, sg, gsw = c['A'], c['up'], c['sg'], c['gsw']
        stream = stream_ptr(self.device)
        self._stamp += 1
        A.stamp = self._stamp
        n_app = applies[0]
        alpha = [CdaeEngine.adam_alpha(self.lr, n_app * step_idx + j + 1, self.beta1, self.beta2) if j is not None else 0.0 for j in applies[1:]]
        for i, tw in c['sg_tw']:
            sg.alpha[i] = alpha[tw]
        # (the small weights' Adam in the launch that sums their partial gradients: drx_dmf_step_small — one launch less than
        # drx_dmf_fwd_bwd + drx_adam_segments, the same bits; the first-layer update reads none of them)
        check(L_.drx_dmf_step_small(c['D'], c['Aref'], gsw, c['sw'], c['m'], c['v'], c['sgref'], self.beta1, self.beta2, self.eps, stream),
              'drx_dmf_step_small')
        up.alpha_u, up.alpha_i = alpha[0], alpha[1]
        check(L_.drx_dmf_k0_update(c['D'], c['Aref'], c['upref'], stream), 'drx_dmf_k0_update')

    def _step_cache_key(self):
        """what a cached argument struct depends on besides its ring slot: the tensors a set_params / set_interactions / optimizer change
        would replace, and the scalars baked into the structs"""
        w = getattr(self, '_work', None)
        return (self.K0u.data_ptr(), self.sw.data_ptr(), self._maps[0].data_ptr(), self._rho[0].data_ptr(), self.csr[0].data_ptr(),
                self.state['sw'][0].data_ptr(), self.state['K0u'][0].data_ptr(), self._k0_order.data_ptr(), self._k0_long,
                w.data_ptr() if w is not None else 0, id(self.D), self.D.off_scale, self.first_layer_update, self.reg, self.beta1, self.beta2,
                self.eps, self.broadcast_targets, self.scale_var is not None, getattr(self, '_step_bufs', (None,))[0])

    def predict(self, uids, iids, want_reps=False, scaled=True):
        """max(1e-6, cosine) for each (uid, iid) pair (dmf.py:88-96) — times the bound prediction scale unless scaled=False;
        optionally the normalised tower outputs [B, self.W]."""
        uid, iid = self._i32(uids), self._i32(iids)
        B = uid.numel()
        pred = torch.empty(B, dtype=torch.float32, device=self.device)
        A = self._base_args(uid, iid)
        A.pred_out = pred.data_ptr()
        ru = ri = None
        if want_reps:
            ru = torch.empty(B, self.W, dtype=torch.float32, device=self.device)
            ri = torch.empty(B, self.W, dtype=torch.float32, device=self.device)
            A.rep_u_out, A.rep_i_out = ru.data_ptr(), ri.data_ptr()
        D = self.D
        if not scaled and self.D.off_scale >= 0:
            D = DmfDims.from_buffer_copy(self.D)
            D.off_scale = -1
        check(lib().drx_dmf_predict(C.byref(D), C.byref(A), stream_ptr(self.device)), 'drx_dmf_predict')
        return (pred, ru, ri) if want_reps else pred

    def score_matrix_bf16(self, uids):
        """[len(uids), N] cosine scores of the given users against ALL items on the matrix cores (bf16 operands)."""
        uid = self._i32(uids)
        n_u = uid.numel()
        all_items = torch.arange(self.N, dtype=torch.int32, device=self.device)
        _, _, ri = self.predict(torch.zeros(self.N, dtype=torch.int32, device=self.device), all_items, want_reps=True)
        _, ru, _ = self.predict(uid, torch.zeros(n_u, dtype=torch.int32, device=self.device), want_reps=True)
        pitch = _round_up(self.N, 32)                  # rows of 128-byte lines: a line then belongs to ONE tile of the scorer
        out = torch.empty(n_u, pitch, dtype=torch.float32, device=self.device)
        kdim = _round_up(self.factors[0][-1], 16)
        if kdim > 64:
            raise _lib.DrxError(f'score_matrix_bf16: representations of up to 64 factors (the scorer stages 64-wide bf16 tiles), got {self.factors[0][-1]}')
        scale = ptr(self.sw[self._scale_slot:]) if self.scale_var is not None else None
        check(lib().drx_score_pairs_bf16(ptr(ru), n_u, ptr(ri), self.N, self.W, kdim, scale, ptr(out), pitch, stream_ptr(self.device)),
              'drx_score_pairs_bf16')
        return out[:, :self.N]
