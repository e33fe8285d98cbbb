"""Row-sharded CDAE across the GPUs of one node (BASELINE.json configuration 4): one process per GPU,
`torch.distributed` over RCCL/xGMI (backend "nccl"); the same orchestration runs over gloo on CPU tensors in the world-size-2 .. 4 tests,
with the per-rank device work swapped for a NumPy statement (tests/dist_ops_numpy.py).

Sharding
  * users (V rows + optimizer state, their positives CSR, and the triples sampled for them): contiguous uid ranges —
    the 5 GB table of the 10M-user configuration never leaves its GPU and needs no exchange;
  * item rows (W, W2T, b2 + state): contiguous item ranges of `ipr = ceil(N / world)` rows;
  * hidden bias b (K floats): replicated; every rank adds the same `world` gradient rows in rank order (they travel as the
    sentinel rows of the gradient exchange: no all-reduce).

One step (every rank, its own B triples; losses / L2 are normalised by the GLOBAL batch so the step equals a single-GPU step on
the concatenated batch).  Parameter-independent, run ahead of the step (ShardedPipeline: on a side stream, 1 - 3 steps early):
  1. prepare: sorted touch list + span plan + sole-toucher marks + launch order (the single-GPU step's preparation), and — from a
     presence map of the wire keys, without waiting for the sort — the batch's DISTINCT item rows, their places in the exchange
     buffers and the per-unit counts                                                                         [drx_shard_prepare]
  2. all-to-all of the counts (results to pinned host memory), all-to-all(v) of the keys (4 B per distinct row); the owner enters
     the keys it received in its direct-address table                                                        [drx_shard_owner_index]
Parameter-dependent, on the training stream:
  3. owners gather the requested rows + output biases into float buffers; all-to-all(v)                      [drx_shard_gather_rows]
  4. forward / backward against the received rows, planned segmented reduction: ONE gradient row per distinct item row (hot Zipf
     rows are merged before they travel), V rows updated in place, bias gradient into the sentinel rows      [drx_shard_step_local]
  5. all-to-all(v) of the gradient buffer back to the owners
  6. owners sum what the ranks sent per row in rank order and apply the sparse optimizer; bias update        [drx_shard_apply]

EXCHANGE CHUNKS (r06; include/drx.h "WIRE keys"): an owner's key range is cut into C chunks and every exchange is C all-to-alls, chunk
c over the units c * world .. c * world + world - 1 (contiguous in every buffer).  The tail of step s and the head of step s + 1 are
then ONE pipeline over the chunks:
        communicator   GX_0  GX_1  GX_2  GX_3        RX'_0   RX'_1   RX'_2   RX'_3
        training       .     A_0 G'_0 A_1 G'_1 A_2 G'_2 A_3 G'_3  .  .   forward(s + 1)
(GX_c: gradient rows of chunk c to their owners; A_c: owner apply of chunk c; G'_c: gather of the rows step s + 1 asked for in chunk c —
they are final once A_c has run, every row lives in exactly one chunk; RX'_c: those rows to their requesters.)  Step s + 1's forward
kernel starts when RX'_{C-1} has landed.  Without chunks the four stages are serial: GX, A, G', RX'.

Exchange buffer geometry (include/drx.h): per unit, in unit order, n rows of ld floats then n scalars padded to 32 floats;
n = distinct rows + 1 sentinel.
"""
import ctypes as C
import math
import time

import numpy as np
import torch
import torch.distributed as dist

KEY_NONE = 0xFFFFFFFF


def items_per_rank(n_items, world):
    return (n_items + world - 1) // world


def wire_shift(ipr):
    """log2 of the wire-key span of one owner: the smallest power of two >= max(2 * ipr, 8192) (include/drx.h)."""
    s = 13
    while (1 << s) < 2 * ipr:
        s += 1
    return s


def wire_chunks(ipr, chunks):
    """The chunk count a shard description gets: a power of two, lowered until a unit spans whole 8192-key tiles (drx_shard_chunks)."""
    c = max(1, int(chunks))
    assert c & (c - 1) == 0, 'exchange chunks: a power of two'
    while c > 1 and wire_shift(ipr) - (c.bit_length() - 1) < 13:
        c >>= 1
    return c


def wire_key(n, ipr, is_out, world=1, chunks=1):
    """unit-major wire key of item n's W (is_out False) or W2T row (python ints): include/drx.h, csrc/drx_prep.hpp WireGeo"""
    cs = wire_shift(ipr) - (wire_chunks(ipr, chunks).bit_length() - 1)
    o = n // ipr
    l = 2 * (n - o * ipr) + (1 if is_out else 0)
    return ((((l >> cs) * world + o) << cs) | (l & ((1 << cs) - 1)))


def wire_local(key, ipr, world=1, chunks=1):
    """(owner, chunk, is W2T row, local item) of a wire key"""
    cs = wire_shift(ipr) - (wire_chunks(ipr, chunks).bit_length() - 1)
    v = key >> cs
    l = ((v // world) << cs) | (key & ((1 << cs) - 1))
    return v % world, v // world, bool(l & 1), l >> 1


def pad32(n):
    return (int(n) + 31) // 32 * 32


def chunk_floats(counts, ld):
    """float split sizes of an exchange among the peers that hold `counts` rows (sentinels included)"""
    return [int(c) * ld + pad32(c) for c in counts]


class HipShardOps:
    """Per-rank device work through the C ABI (include/drx.h, drx_shard_*)."""

    def __init__(self, n_users_local, n_items, k, rank, world, device, optimizer, lr, reg, self_bypass=True, chunks=1):
        from . import _lib
        from .engine import CdaeEngine
        self._lib = _lib
        self.L = _lib.lib()
        self.ipr = items_per_rank(n_items, world)
        self.rank, self.world = rank, world
        self.engine = CdaeEngine(n_users_local, self.ipr, k, device=device)
        self.engine.init_optimizer(optimizer, lr, reg)
        self.device = self.engine.device
        self.ld = self.engine.ld
        # the rank's own rows never pass through a collective (include/drx.h DRX_SHARD_SELF_BYPASS); off: every row travels (a
        # measurement aid: at world 1 it sends the whole exchange through the communicator)
        self.self_bypass = bool(self_bypass) and self.ipr * self.ld < (1 << 31)
        self.shard = _lib.Shard(world, rank, n_items, self.ipr, n_users_local, _lib.SHARD_SELF_BYPASS if self.self_bypass else 0,
                                wire_chunks(self.ipr, chunks))
        self.chunks = int(self.L.drx_shard_chunks(C.byref(self.shard)))
        if self.chunks < 1:
            raise _lib.DrxError('drx_shard_chunks: invalid shard description')
        wb = int(self.L.drx_shard_work_bytes(C.byref(self.shard)))
        if wb <= 0:
            raise _lib.DrxError('drx_shard_work_bytes: invalid shard description')
        self._work = torch.zeros(wb, dtype=torch.uint8, device=self.device)      # the presence map: zero between preparations
        self._scratch = None
        self._loss = torch.zeros(2, dtype=torch.float32, device=self.device)
        self._tables = {}
        self._cnt = {}

    def _stream(self):
        return self._lib.stream_ptr(self.device)

    def _e(self, n, dtype=torch.float32):
        return torch.empty(int(n), dtype=dtype, device=self.device)

    def _counts(self, counts):
        """host int32 array of a count list (ctypes arrays are kept per length: building one costs more than filling it)"""
        n = len(counts)
        a = self._cnt.get(n)
        if a is None:
            a = self._cnt[n] = (C.c_int32 * n)()
        a[:] = counts
        return a

    # -- parameter-independent
    def prepare(self, bt, out=None):
        """drx_shard_prepare on the CURRENT stream; returns the views a step and the exchanges need.  `out`: a buffer to reuse."""
        P, sh, lib = self.engine._params, self.shard, self.L
        need = int(lib.drx_shard_prep_bytes(C.byref(P), C.byref(sh), bt.B, bt.n_touch_slots))
        if out is None or out.numel() < need:
            out = torch.empty(int(need * 1.05) + 4096, dtype=torch.uint8, device=self.device)
        lay = (C.c_size_t * 4)()
        self._lib.check(lib.drx_shard_prep_layout(C.byref(P), C.byref(sh), bt.B, bt.n_touch_slots, lay), 'drx_shard_prep_layout')
        self._lib.check(lib.drx_shard_prepare(C.byref(P), C.byref(sh), C.byref(self.engine._hist), C.byref(bt), self._lib.ptr(out),
                                              out.numel(), self._lib.ptr(self._work), self._work.numel(), self._stream()),
                        'drx_shard_prepare')
        uniq = out[lay[0]:lay[0] + 4 * lay[2]].view(torch.int32)
        counts = out[lay[1]:lay[1] + 8 * self.world * self.chunks].view(torch.int64)       # owner-major: [owner][chunk]
        return {'buf': out, 'uniq': uniq, 'counts_dev': counts}

    def owner_index(self, req, recv_counts, slot=0, chunk=0):
        """The owner's table of the keys it received for one chunk (parameter-independent; one table per in-flight step `slot`,
        shared by the step's chunks)."""
        n_seg = len(recv_counts)
        nb = int(self.L.drx_shard_owner_table_bytes(C.byref(self.shard), n_seg))
        tab = self._tables.get(slot)
        if tab is None or tab.numel() < nb:
            tab = self._tables[slot] = torch.empty(nb, dtype=torch.uint8, device=self.device)
        self._lib.check(self.L.drx_shard_owner_index(C.byref(self.shard), self._lib.ptr(req), req.numel(), self._counts(recv_counts), n_seg,
                                                     int(chunk), self._lib.ptr(tab), tab.numel(), self._stream()), 'drx_shard_owner_index')
        return tab

    # -- parameter-dependent
    def xsplits(self, counts):
        """float split sizes of ONE CHUNK's exchange among the peers that hold `counts` rows: nothing travels to the rank itself when
        its own rows bypass the collectives"""
        f = chunk_floats(counts, self.ld)
        if self.self_bypass:
            f = [0 if (i % self.world) == self.rank else x for i, x in enumerate(f)]
        return f

    def gather_rows(self, req, recv_counts, out=None):
        """the rows one chunk's requests name, segment after segment"""
        n = max(32, sum(self.xsplits(recv_counts)))
        if out is None or out.numel() < n:
            out = self._e(n)
        self._lib.check(self.L.drx_shard_gather_rows(C.byref(self.engine._params), C.byref(self.shard), self._lib.ptr(req), req.numel(),
                                                     self._counts(recv_counts), len(recv_counts), self._lib.ptr(out), self._stream()),
                        'drx_shard_gather_rows')
        return out

    def local_step(self, bt, prep, rows_cache, b_norm, loss_kind, opt, events=None, out=None):
        """Forward / backward + reduction of one (micro-)batch: the gradient buffer on its way back (same geometry as rows_cache,
        the rank's own units — with the bypass — at its end)."""
        P = self.engine._params
        need = int(self.L.drx_shard_step_scratch_bytes(C.byref(P), bt.B, bt.n_touch_slots))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(int(need * 1.1) + 4096, dtype=torch.uint8, device=self.device)
        n = max(32, sum(sum(chunk_floats(sc, self.ld)) for sc in prep['send_counts']))
        gsend = out if (out is not None and out.numel() >= n) else self._e(n)
        arr = (C.c_void_p * len(events))(*[e.cuda_event for e in events]) if events is not None else None
        buf = prep['buf']
        self._lib.check(self.L.drx_shard_step_local(C.byref(P), C.byref(opt), C.byref(self.shard), C.byref(self.engine._hist), C.byref(bt),
                                                    self._lib.ptr(buf), buf.numel(), self._lib.ptr(rows_cache), self._lib.ptr(gsend),
                                                    int(b_norm), int(loss_kind), self._lib.ptr(self._scratch), self._scratch.numel(), arr,
                                                    self._stream()), 'drx_shard_step_local')
        return gsend

    def apply(self, req, grecv, recv_counts, table, b_norm, opt, want_loss=False, own=None, chunk=0):
        """One chunk of the owner apply.  own: per micro-batch (its gradient buffer, float offset of the rank's own piece of this chunk)"""
        og = oo = None
        if self.self_bypass:
            og = (C.c_void_p * len(own))(*[self._lib.ptr(g) for g, _ in own])
            oo = (C.c_int64 * len(own))(*[int(o) for _, o in own])
        self._lib.check(self.L.drx_shard_apply(C.byref(self.engine._params), C.byref(opt), C.byref(self.shard), int(b_norm),
                                               self._lib.ptr(req), self._lib.ptr(grecv), req.numel(), self._counts(recv_counts),
                                               len(recv_counts), int(chunk), self._lib.ptr(table), og, oo,
                                               self._lib.ptr(self._loss) if want_loss else None, self._stream()), 'drx_shard_apply')
        return self._loss if want_loss else None

    # -- the exchange phases issued from C (include/drx.h drx_shard_phase_*; one micro-batch per step, the library's communicator)
    def x_new(self, prep, counts_host):
        """DrxShardExchange of a prepared batch whose count exchange has landed in `counts_host` (pinned int64 [2, world * chunks])"""
        X = self._lib.ShardExchange()
        X.send_counts, X.recv_counts = counts_host[0].data_ptr(), counts_host[1].data_ptr()
        X.uniq = prep['uniq'].data_ptr()
        sizes = (C.c_int64 * 4)()
        self._lib.check(self.L.drx_shard_exchange_sizes(C.byref(self.engine._params), C.byref(self.shard), X.send_counts, X.recv_counts,
                                                        sizes), 'drx_shard_exchange_sizes')
        return X, tuple(int(v) for v in sizes)

    def x_keys(self, comm, X, req, slot):
        nb = int(self.L.drx_shard_owner_table_bytes(C.byref(self.shard), self.world))
        tab = self._tables.get(slot)
        if tab is None or tab.numel() < nb:
            tab = self._tables[slot] = torch.empty(nb, dtype=torch.uint8, device=self.device)
        X.req, X.table, X.table_bytes = req.data_ptr(), tab.data_ptr(), tab.numel()
        self._lib.check(self.L.drx_shard_phase_keys(C.byref(self.shard), comm, C.byref(X), self._stream()), 'drx_shard_phase_keys')
        return tab

    def x_rows(self, comm, X, chunk):
        self._lib.check(self.L.drx_shard_phase_rows(C.byref(self.engine._params), C.byref(self.shard), comm, C.byref(X), int(chunk),
                                                    self._stream()), 'drx_shard_phase_rows')

    def x_local(self, comm, X, bt, prep, b_norm, loss_kind, opt, events=None):
        P = self.engine._params
        need = int(self.L.drx_shard_step_scratch_bytes(C.byref(P), bt.B, bt.n_touch_slots))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(int(need * 1.1) + 4096, dtype=torch.uint8, device=self.device)
        arr = (C.c_void_p * len(events))(*[e.cuda_event for e in events]) if events is not None else None
        buf = prep['buf']
        self._lib.check(self.L.drx_shard_phase_local(C.byref(P), C.byref(opt), C.byref(self.shard), C.byref(self.engine._hist), C.byref(bt),
                                                     comm, C.byref(X), self._lib.ptr(buf), buf.numel(), int(b_norm), int(loss_kind),
                                                     self._lib.ptr(self._scratch), self._scratch.numel(), arr, self._stream()),
                        'drx_shard_phase_local')

    def x_tail(self, comm, X, X_next, b_norm, opt, want_loss=False):
        self._lib.check(self.L.drx_shard_phase_tail(C.byref(self.engine._params), C.byref(opt), C.byref(self.shard), comm, C.byref(X),
                                                    C.byref(X_next) if X_next is not None else None, int(b_norm),
                                                    self._lib.ptr(self._loss) if want_loss else None, self._stream()), 'drx_shard_phase_tail')
        return self._loss if want_loss else None

    def optim(self, step):
        a = self.engine.adam_alpha(self.engine.lr, step + 1, self.engine.beta1, self.engine.beta2)
        return self.engine._optim([a] * 5)

    def make_batch(self, *a, **k):
        return self.engine.make_batch(*a, **k)

    def set_params(self, W, W_, V, b, b_):
        self.engine.set_params(W=W, W_=W_, V=V, b=b, b_=b_)

    def get_params(self):
        return self.engine.get_params()


class _Done:
    """handle of an exchange that has already happened (or was issued in stream order)"""
    def wait(self):
        pass


_DONE = _Done()


class TorchTransport:
    """all-to-all(v) through torch.distributed (RCCL on GPU tensors: backend "nccl"; gloo on CPU tensors or — cpu_staging — staged
    through the host for ranks that share one GPU in tests).  a2a() returns a handle; handle.wait() makes the CURRENT stream wait for
    the rows (a no-op for exchanges that ran synchronously)."""

    def __init__(self, group, world, collectives, cpu_staging):
        self.group, self.world, self.collectives, self.cpu_staging = group, world, collectives, cpu_staging

    def a2a(self, send, send_splits, out, recv_splits, overlap=False):
        n_s, n_r = int(sum(send_splits)), int(sum(recv_splits))
        dst, send = out[:n_r], send[:n_s]
        if not self.collectives:
            dst.copy_(send)
            return _DONE
        if self.world == 1 and n_r == 0 and n_s == 0:
            return _DONE                               # (a 1-rank communicator with the self-bypass: nothing to exchange)
        if (self.cpu_staging and send.is_cuda) or not send.is_cuda:
            o = torch.empty(n_r, dtype=send.dtype)
            dist.all_to_all_single(o, send.contiguous().cpu(), output_split_sizes=[int(c) for c in recv_splits],
                                   input_split_sizes=[int(c) for c in send_splits], group=self.group)
            dst.copy_(o)
            return _DONE
        # async_op: the training stream keeps running kernels (another chunk's apply, another micro-batch) while the rows travel;
        # on a single exchange the asynchronous form only adds a stream hop
        work = dist.all_to_all_single(dst, send, output_split_sizes=[int(c) for c in recv_splits],
                                      input_split_sizes=[int(c) for c in send_splits], group=self.group, async_op=overlap)
        return work if overlap else _DONE


class _Ticket:
    """handle of an exchange queued on the library's communicator; keeps the buffers alive until a stream has waited for it"""
    __slots__ = ('xfer', 't', 'bufs')

    def __init__(self, xfer, t, bufs):
        self.xfer, self.t, self.bufs = xfer, t, bufs

    def wait(self):
        if self.t is not None:
            x = self.xfer
            x.check(x.L.drx_comm_wait(x.comm, self.t, x.stream_ptr(x.device)), 'drx_comm_wait')
            self.bufs = None                          # (the waiting stream is behind the exchange now: its later work may reuse them)


class RcclTransport:
    """all-to-all(v) through the library's own RCCL communicator (include/drx.h drx_comm_*, csrc/drx_comm.hip): one ncclGroup of send /
    recv pairs per exchange, enqueued from C on the communicator's stream, ordered with the CURRENT torch stream by events.  r06: the
    chunked schedule issues 2 + 2 C exchanges per step; through torch.distributed each cost tens of microseconds of Python and the host
    became the bound (profiles/r06b_*).  One rank per GPU (RCCL refuses two ranks on one device): tests with several ranks on one GPU
    keep TorchTransport over gloo.  `group`: an initialised torch.distributed group of the same ranks, used ONCE to hand rank 0's
    unique id to the others (world 1: nothing)."""

    def __init__(self, world, rank, device, group=None, threaded=True):
        from . import _lib
        self._lib, self.L, self.check, self.stream_ptr = _lib, _lib.lib(), _lib.check, _lib.stream_ptr
        self.world, self.rank, self.device = world, rank, torch.device(device)
        ident = C.create_string_buffer(128)
        if rank == 0:
            self._ok(self.L.drx_comm_unique_id(ident), 'drx_comm_unique_id')
        if world > 1:
            box = [ident.raw if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            self._ok(self.L.drx_comm_create(ident, world, rank, 1 if threaded else 0, C.byref(h)), 'drx_comm_create')
        self.comm = h
        self._arr = [(C.c_int64 * world)() for _ in range(4)]
        self.issued = 0

    def _ok(self, rc, what):
        if rc < 0:
            raise self._lib.DrxError(f'{what}: {self._lib.lib().drx_comm_last_error().decode()} (code {rc})')

    def a2a(self, send, send_splits, out, recv_splits, overlap=False):
        so, sb, ro, rb = self._arr
        es, er = send.element_size(), out.element_size()
        a = 0
        for p, n in enumerate(send_splits):
            so[p], sb[p] = a * es, int(n) * es
            a += int(n)
        a = 0
        for p, n in enumerate(recv_splits):
            ro[p], rb[p] = a * er, int(n) * er
            a += int(n)
        t = self.L.drx_comm_alltoallv(self.comm, send.data_ptr(), so, sb, out.data_ptr(), ro, rb, self.stream_ptr(self.device))
        self._ok(t, 'drx_comm_alltoallv')
        self.issued += 1
        h = _Ticket(self, t, (send, out))
        if not overlap:
            h.wait()
        return h

    def close(self):
        if self.comm:
            self.L.drx_comm_destroy(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedCdae:
    """One rank of the row-sharded sampled-mode CDAE.  `ops` = per-rank compute backend (HipShardOps on a GPU)."""

    def __init__(self, n_users_total, n_items, k, rank, world, device, hist_indptr, hist_indices, seed=10, lr=0.05, reg=1e-3,
                 optimizer='adagrad', ops=None, group=None, loss='bce', q=0.2, cpu_staging=False, force_collectives=False,
                 self_bypass=True, chunks=1, transport=None, phases=None):
        self.rank, self.world, self.group = rank, world, group
        # world 1 normally bypasses torch.distributed; `force_collectives` sends every exchange through the process group
        # anyway (a 1-rank RCCL communicator exercises the exact call sequence of the N-rank step on one GPU)
        self.collectives = world > 1 or force_collectives
        self.cpu_staging = cpu_staging        # tests: gloo has no device all-to-all; stage the exchange through the host
        self.n_items, self.k = n_items, k
        self.ipr = items_per_rank(n_items, world)
        self.user_lo = n_users_total * rank // world
        self.user_hi = n_users_total * (rank + 1) // world
        self.n_users_total = n_users_total
        n_local = self.user_hi - self.user_lo
        self.ops = ops if ops is not None else HipShardOps(n_local, n_items, k, rank, world, device, optimizer, lr, reg, self_bypass, chunks)
        self.chunks = int(getattr(self.ops, 'chunks', 1))          # (what the shard's key range allows: <= the request)
        self.engine = getattr(self.ops, 'engine', None)
        self.ld = getattr(self.ops, 'ld', k)
        self.loss_kind = 0 if loss == 'bce' else 1
        self.q = q
        # transport: an object with a2a() (TorchTransport / RcclTransport), or 'rccl' = the library's own communicator (one rank per
        # GPU, or world 1 with force_collectives), default = torch.distributed
        if transport == 'rccl':
            transport = self._own_communicator(world, rank, device, group) if self.collectives else None
        self.xfer = transport if transport is not None else TorchTransport(group, world, self.collectives, cpu_staging)
        # phases: the exchanges of a step issued by the library (drx_shard_phase_*: four calls per step, split sizes read from the count
        # exchange's pinned mailbox) instead of call by call from here.  Needs the library's communicator; steps of several micro-batches
        # keep the call-by-call form.  Default: on with transport='rccl'
        can = isinstance(self.xfer, RcclTransport) and isinstance(self.ops, HipShardOps)
        if phases and not can:
            raise ValueError("phases=True needs transport='rccl' (the library's own communicator) and the HIP shard ops")
        self.phases = can if phases is None else bool(phases)
        if self.engine is not None:
            # (the engine's tables are this rank's SHARD — its item rows are local, the history's item ids global: no transpose)
            self.engine.set_history(hist_indptr, hist_indices, with_transpose=False)
            self._init_random(seed)
        self.last_loss = None
        self._cur = self._main = None         # ShardedPipeline: the stream its run-ahead stages are queued on / the training stream
        self.wait_s = 0.0                     # host time spent waiting for count exchanges (stays ~0 when pipelined)

    @staticmethod
    def _own_communicator(world, rank, device, group):
        """RcclTransport, or None (= torch.distributed) when ANY rank cannot open librccl — the ranks agree first (a rank that raised
        alone would leave the others waiting inside the communicator's set-up)."""
        from . import _lib
        probe = C.create_string_buffer(128)
        ok = 1 if _lib.lib().drx_comm_unique_id(probe) == 0 else 0
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=torch.device(device) if dist.get_backend(group) == 'nccl' else 'cpu')
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            ok = int(flag.item())
        if not ok:
            import warnings
            warnings.warn('drecpy_amd: librccl could not be opened on every rank (' + _lib.lib().drx_comm_last_error().decode() +
                          '); the exchanges go through torch.distributed')
            return None
        return RcclTransport(world, rank, device, group)

    def _init_random(self, seed):
        """GlorotUniform of the GLOBAL shapes, drawn per shard on the device (cdae.py:35-41)."""
        e = self.engine
        gen = torch.Generator(device=e.device)
        gen.manual_seed(int(seed) * 1000003 + self.rank)
        k = self.k

        def fill(t, fi, fo):
            lim = math.sqrt(6.0 / (fi + fo))
            t.zero_()
            view = t[:, :k] if t.dim() == 2 else t
            view.copy_((torch.rand(view.shape, generator=gen, device=e.device, dtype=torch.float32) * 2 - 1) * lim)
        fill(e.W, self.n_items, k)
        fill(e.W2T, k, self.n_items)
        fill(e.V, self.n_users_total, k)
        fill(e.b2, self.n_items, self.n_items)
        gb = torch.Generator(device=e.device)
        gb.manual_seed(int(seed))                     # b is replicated: same draw on every rank
        e.b.zero_()
        e.b[:k] = (torch.rand(k, generator=gb, device=e.device) * 2 - 1) * math.sqrt(3.0 / k)

    def set_params_global(self, W, W_, V, b, b_):
        """Slices global (reference-orientation) weights into this rank's shards."""
        ipr, r = self.ipr, self.rank
        lo, hi = r * ipr, min(self.n_items, (r + 1) * ipr)
        dt = np.asarray(W).dtype
        Wl = np.zeros((ipr, self.k), dt); Wl[:hi - lo] = W[lo:hi]
        W_l = np.zeros((self.k, ipr), dt); W_l[:, :hi - lo] = W_[:, lo:hi]
        b_l = np.zeros(ipr, dt); b_l[:hi - lo] = b_[lo:hi]
        self.ops.set_params(Wl, W_l, np.asarray(V[self.user_lo:self.user_hi], dt), np.asarray(b, dt), b_l)

    def gather_params_global(self):
        """The whole model on every rank (reference orientation: W [N,K], W_ [K,N], V [U,K], b [K], b_ [N]) — for evaluation or
        export after training; an infrequent host-side gather, not part of a step."""
        mine = self.ops.get_params()
        N = self.n_items
        if self.world == 1:
            return {'W': mine['W'][:N], 'W_': mine['W_'][:, :N], 'V': mine['V'], 'b': mine['b'], 'b_': mine['b_'][:N]}
        if not all(isinstance(mine[k], np.ndarray) for k in ('W', 'W_', 'V', 'b_')):          # (test doubles with other containers)
            parts = [None] * self.world
            dist.all_gather_object(parts, mine, group=self.group)
            return {'W': np.concatenate([p['W'] for p in parts], axis=0)[:N], 'W_': np.concatenate([p['W_'] for p in parts], axis=1)[:, :N],
                    'V': np.concatenate([p['V'] for p in parts], axis=0), 'b': parts[0]['b'], 'b_': np.concatenate([p['b_'] for p in parts])[:N]}
        # tensors, not pickled objects: at BASELINE configuration 4 a rank's V shard alone is 640 MB (all_gather_object would serialise it,
        # pad every rank's blob to the largest and keep both copies).  Shards enter with their sharded axis first, padded to the largest
        # shard (user ranges differ by one row when world does not divide U; item tables are ipr rows on every rank).
        W = self.world
        dev = torch.device(self.engine.device) if (self.engine is not None and dist.get_backend(self.group) == 'nccl') else torch.device('cpu')
        rows = {'W': [self.ipr] * W, 'W_': [self.ipr] * W, 'b_': [self.ipr] * W,
                'V': [self.n_users_total * (r + 1) // W - self.n_users_total * r // W for r in range(W)]}
        out = {'b': mine['b']}
        for name in ('W', 'W_', 'V', 'b_'):
            a = np.ascontiguousarray(mine[name].T if name == 'W_' else mine[name])
            n_max = max(rows[name])
            t = torch.zeros((n_max,) + a.shape[1:], dtype=torch.as_tensor(a[:0]).dtype, device=dev)
            t[:a.shape[0]].copy_(torch.as_tensor(a))
            parts = [torch.empty_like(t) for _ in range(W)]
            dist.all_gather(parts, t, group=self.group)
            full = torch.cat([p[:rows[name][r]] for r, p in enumerate(parts)], dim=0).cpu().numpy()
            out[name] = full.T if name == 'W_' else full
        out['W'], out['W_'], out['b_'] = out['W'][:N], out['W_'][:, :N], out['b_'][:N]
        return out

    # ---- exchanges ---------------------------------------------------------------------------------------
    def _a2a(self, send, send_counts, recv_counts, out=None, overlap=False):
        """all-to-all(v) of a 1-D tensor with per-peer split sizes; returns (received tensor, handle).  The returned tensor is never
        empty (at least 32 elements are allocated: an exchange may be empty when a rank's own rows bypass it, and the library wants
        a pointer); its first sum(recv_counts) elements are the received ones."""
        n = int(sum(recv_counts))
        if out is None:
            out = torch.empty(max(n, 32), dtype=send.dtype, device=send.device)
        return out, self.xfer.a2a(send, send_counts, out, recv_counts, overlap)

    # ---- the parameter-independent stages ---------------------------------------------------------------------
    # They may run ahead of the training stream (ShardedPipeline below):
    #   prepare(bt)            local: touch list, plan, marks, distinct rows, per-unit counts            [no collective]
    #   exchange_counts(P)     all-to-all of the per-(owner, chunk) counts; the result goes to pinned host memory
    #   exchange_keys(P)       per chunk an all-to-all(v) of the distinct keys each owner is asked for (needs the counts on the host)
    #   index_owner(Ps)        the owner's table of the keys it received for all micro-batches of a step
    # Every rank must call the stages in the same program order: they all run on one communicator.
    def prepare(self, bt, consumer_stream=None, out=None):
        P = self.ops.prepare(bt, out) if out is not None else self.ops.prepare(bt)
        P.setdefault('buf', None)
        P['consumer'] = consumer_stream             # the stream that will run the step, when the stages run ahead on another one
        P['event'] = None                           # recorded behind the LAST run-ahead stage (index_owner): what the step waits for
        return P

    def _split_counts(self, P, send_flat, recv_flat):
        """owner-major flat count vectors ([peer][chunk]) -> per chunk, per peer"""
        W, Cn = self.world, self.chunks
        P['send_counts'] = [[int(send_flat[o * Cn + c]) for o in range(W)] for c in range(Cn)]
        P['recv_counts'] = [[int(recv_flat[s * Cn + c]) for s in range(W)] for c in range(Cn)]

    def exchange_counts(self, P):
        """How many rows (sentinel included) every rank asks of every owner in every chunk.  On the device path nothing here waits on
        the host: the send counts are a view of the prepared buffer, both count vectors land in pinned memory."""
        W, Cn = self.world, self.chunks
        if 'counts_dev' in P and not self.cpu_staging:
            send = P['counts_dev']
            recv = torch.empty_like(send)
            P['counts_x'] = self.xfer.a2a(send, [Cn] * W, recv, [Cn] * W)        # (waited for by the current stream inside)
            host = torch.empty(2, W * Cn, dtype=torch.int64, pin_memory=True)
            host[0].copy_(send, non_blocking=True)
            host[1].copy_(recv, non_blocking=True)
            P['counts_host'] = host
            P['counts_event'] = torch.cuda.Event()
            P['counts_event'].record(self._cur) if self._cur is not None else P['counts_event'].record()
        else:
            if 'counts' in P:
                send_counts = [int(c) for c in P['counts']]
            else:
                send_counts = P['counts_dev'].cpu().tolist()
            if self.collectives:
                s = torch.tensor(send_counts, dtype=torch.int64)
                r = torch.empty_like(s)
                dist.all_to_all_single(r, s, group=self.group)
                recv_counts = r.tolist()
            else:
                recv_counts = list(send_counts)
            self._split_counts(P, send_counts, recv_counts)
        return P

    def exchange_keys(self, P, solo=True):
        """Every owner learns which of its rows each rank wants (4 B per distinct row): one all-to-all(v) per chunk.
        solo: the batch is its step's only micro-batch (then — self.phases — the library issues the exchanges: drx_shard_phase_*)."""
        if 'send_counts' not in P:
            if 'counts_host' not in P:
                self.exchange_counts(P)
            if 'counts_host' in P:
                t0 = time.perf_counter()
                P['counts_event'].synchronize()         # the tiny count exchange, issued at least two steps earlier
                self.wait_s += time.perf_counter() - t0
                if self.phases and solo:
                    # the library reads the mailbox itself; here only the buffer sizes.  The key exchange proper happens in
                    # index_owner (drx_shard_phase_keys: exchange + owner table, chunk by chunk)
                    P['x'], P['sizes'] = self.ops.x_new(P, P['counts_host'])
                    P['req'] = torch.empty(P['sizes'][2], dtype=torch.int32, device=P['uniq'].device)
                    if P.get('consumer') is not None:
                        P['req'].record_stream(P['consumer'])
                    return P
                self._split_counts(P, P['counts_host'][0].tolist(), P['counts_host'][1].tolist())
        if 'send_counts' not in P:
            self.exchange_counts(P)
        P['req'], k0 = [], 0
        for c in range(self.chunks):
            sc, rc = P['send_counts'][c], P['recv_counts'][c]
            req, _ = self._a2a(P['uniq'][k0:], sc, rc)
            req = req[:int(sum(rc))]
            if req.is_cuda and P.get('consumer') is not None:
                req.record_stream(P['consumer'])
            P['req'].append(req)
            k0 += int(sum(sc))
        return P

    def index_owner(self, Ps, slot=0):
        """The owner side of a step's key exchange(s): per chunk one run of segments (micro-batch-major, then source), one table."""
        head = Ps[0]
        if 'x' in head:
            if len(Ps) != 1:
                raise ValueError('phases=True runs one micro-batch per step')
            head['table'] = self.ops.x_keys(self.xfer.comm, head['x'], head['req'], slot)
            if head.get('consumer') is not None:
                head['event'] = torch.cuda.Event()
                head['event'].record(self._cur) if self._cur is not None else head['event'].record()
            return head
        head['req_all'], head['counts_all'] = [], []
        for c in range(self.chunks):
            req = Ps[0]['req'][c] if len(Ps) == 1 else torch.cat([P['req'][c] for P in Ps])
            cnt = [int(n) for P in Ps for n in P['recv_counts'][c]]
            head['req_all'].append(req)
            head['counts_all'].append(cnt)
            head['table'] = self.ops.owner_index(req, cnt, slot, c)
            if req.is_cuda and head.get('consumer') is not None and len(Ps) > 1:
                req.record_stream(head['consumer'])
        if head['req_all'][0].is_cuda and head.get('consumer') is not None:
            head['event'] = torch.cuda.Event()
            head['event'].record(self._cur) if self._cur is not None else head['event'].record()
        return head

    # ---- the two halves of a step's exchanges ------------------------------------------------------------------
    def _ready(self, bts, Ps):
        """prepared + keys exchanged + owner table built for a step's micro-batches (inline, when nothing ran ahead)"""
        for m, b in enumerate(bts):
            P = Ps[m] = Ps[m] if Ps[m] is not None else self.prepare(b)
            if 'req' not in P:
                self.exchange_keys(P, solo=len(bts) == 1)
        if 'table' not in Ps[0]:
            self.index_owner(Ps)
        return Ps

    def _main_waits(self, Ps):
        if Ps[0]['event'] is not None:                 # the run-ahead stages of this step, all queued on one side stream
            (self._main or torch.cuda.current_stream()).wait_event(Ps[0]['event'])

    def _fetch_chunk(self, Ps, c, overlap):
        """owners answer chunk c of every micro-batch's request from the tables as they are NOW; the rows travel"""
        ops = self.ops
        for P in Ps:
            if 'cache' not in P:
                n = sum(sum(ops.xsplits(sc)) for sc in P['send_counts'])
                P['cache'] = torch.empty(max(32, n), dtype=torch.float32, device=P['uniq'].device) if torch.is_tensor(P['uniq']) and P['uniq'].is_cuda \
                    else torch.zeros(max(32, n), dtype=torch.float64)
                P['cache_off'] = np.concatenate([[0], np.cumsum([sum(ops.xsplits(sc)) for sc in P['send_counts']])]).astype(np.int64)
                P['rx'] = []
            rows = ops.gather_rows(P['req'][c], P['recv_counts'][c])
            if P['cache'].dtype != rows.dtype:
                P['cache'] = P['cache'].to(rows.dtype)
            o = int(P['cache_off'][c])
            P['rx'].append(self._a2a(rows, ops.xsplits(P['recv_counts'][c]), ops.xsplits(P['send_counts'][c]), out=P['cache'][o:],
                                     overlap=overlap)[1])

    def fetch_rows(self, Ps):
        """head of a step: gather + row exchange of all chunks (when the previous step's tail did not already do it)"""
        if 'x' in Ps[0]:
            self._x_row_buffers(Ps[0])
            for c in range(self.chunks):
                self.ops.x_rows(self.xfer.comm, Ps[0]['x'], c)
            Ps[0]['rx'] = True
            return
        ov = len(Ps) > 1 or self.chunks > 1
        for c in range(self.chunks):
            self._fetch_chunk(Ps, c, ov)

    def _x_row_buffers(self, P):
        """the two row buffers of a prepared batch (allocated with the TRAINING stream current: both are read and written only by it and
        by exchanges it has waited for when the batch's dict is dropped)"""
        if 'rows_cache' not in P:
            dev = P['uniq'].device
            P['rows_cache'] = torch.empty(P['sizes'][0], dtype=torch.float32, device=dev)
            P['rows_send'] = torch.empty(P['sizes'][1], dtype=torch.float32, device=dev)
            P['x'].rows_cache, P['x'].rows_send = P['rows_cache'].data_ptr(), P['rows_send'].data_ptr()

    def _step_phases(self, step, bt, P, events, want_loss, after_row_requests, after_apply, next_prepared):
        """step() with the exchanges issued by the library: four calls (rows of the head only when nothing ran ahead)"""
        ops, comm = self.ops, self.xfer.comm
        b_norm = bt.B * self.world
        opt = ops.optim(step)
        rec = (lambda i: events[i].record()) if events is not None else (lambda i: None)
        rec(0)
        Ps = self._ready([bt], [P])
        P = Ps[0]
        self._main_waits(Ps)
        if 'rx' not in P:
            self.fetch_rows(Ps)
        dev = P['uniq'].device
        P['grad_send'] = torch.empty(P['sizes'][0], dtype=torch.float32, device=dev)
        P['grad_recv'] = torch.empty(P['sizes'][1], dtype=torch.float32, device=dev)
        X = P['x']
        X.grad_send, X.grad_recv = P['grad_send'].data_ptr(), P['grad_recv'].data_ptr()
        ops.x_local(comm, X, bt, P, b_norm, self.loss_kind, opt, events=events[1:5] if events is not None else None)
        if after_row_requests is not None:
            after_row_requests()                       # (the run-ahead exchanges: in front of this step's gradient exchange)
        nxt = next_prepared() if callable(next_prepared) else next_prepared
        if nxt is not None:
            nxt = list(nxt) if isinstance(nxt, (list, tuple)) else [nxt]
            if len(nxt) != 1 or 'table' not in nxt[0] or 'x' not in nxt[0]:
                nxt = None
            else:
                self._main_waits(nxt)
                self._x_row_buffers(nxt[0])
        loss = ops.x_tail(comm, X, nxt[0]['x'] if nxt is not None else None, b_norm, opt, want_loss=want_loss)
        if nxt is not None:
            nxt[0]['rx'] = True
        rec(5)
        if after_apply is not None:
            after_apply()
        if want_loss:
            self.last_loss = float(loss[0].item())
            return self.last_loss
        return None

    # ---- one step ------------------------------------------------------------------------------------------
    def step(self, step, bt, events=None, want_loss=False, prepared=None, after_row_requests=None, after_apply=None, next_prepared=None):
        """One training step of this rank.  `bt` is a DrxBatch or a list of up to DRX_MAX_MICRO micro-batches with pairwise
        DISJOINT users (then `prepared` is the matching list): all micro-batches read the pre-step parameters and every
        owned row is updated once with the sum of their gradients — the step still equals the single-process step on the
        concatenated batch — but the row exchange of micro-batch m+1 and the gradient exchange of micro-batch m travel while
        the other one computes.
        next_prepared: the prepared (keys exchanged, owner table built) micro-batches of the NEXT step, or a callable that returns them.  Their rows are then gathered
        and sent chunk by chunk right behind this step's owner apply of the same chunk (the module docstring's pipeline), and the
        next call finds them in flight instead of fetching them at its head.
        Event slots (bench; one micro-batch): [0,1) waiting for the rows (fetched here when the previous step did not), [1,2) forward /
        backward kernel, [2,3) reduction, [3,4) span launch (+ the rank's bias row), [4,5) gradient exchange + owner apply (+ bias
        update) + — pipelined — the next step's gather and row exchange; slots 1..4 are recorded by the library between its launches
        (drx_shard_step_local).  With several micro-batches: [1,4) = all of them.
        after_row_requests: called once the kernels of this step's first micro-batch are queued — ShardedPipeline queues the
        run-ahead count / key exchanges of later batches there (they travel while the step computes).
        after_apply: called once the whole step is queued — the pipeline's collective-free preparation of a later batch goes there
        (its host time would otherwise delay the issue of this step's owner apply)."""
        ops = self.ops
        bts = list(bt) if isinstance(bt, (list, tuple)) else [bt]
        Ps = (list(prepared) if isinstance(prepared, (list, tuple)) else [prepared]) if prepared is not None else [None] * len(bts)
        if self.phases and len(bts) == 1 and (Ps[0] is None or 'send_counts' not in Ps[0]):
            return self._step_phases(step, bts[0], Ps[0], events, want_loss, after_row_requests, after_apply, next_prepared)
        b_norm = sum(b.B for b in bts) * self.world
        opt = ops.optim(step)
        rec = (lambda i: events[i].record()) if events is not None else (lambda i: None)
        Cn = self.chunks
        rec(0)
        self._ready(bts, Ps)
        self._main_waits(Ps)
        ov = len(Ps) > 1 or Cn > 1
        if 'rx' not in Ps[0]:
            self.fetch_rows(Ps)
        for h in Ps[0]['rx']:
            h.wait()
        inner = events[1:5] if (events is not None and len(Ps) == 1) else None
        if inner is None:
            rec(1)
        # ---- forward / backward + local reduction of every micro-batch; gradient rows leave chunk by chunk
        pushed = [[None] * len(Ps) for _ in range(Cn)]
        gbufs = []
        n_recv = [[sum(ops.xsplits(P['recv_counts'][c])) for P in Ps] for c in range(Cn)]
        grecv = [None] * Cn
        for m, (b, P) in enumerate(zip(bts, Ps)):
            for h in P['rx']:
                h.wait()
            gsend = ops.local_step(b, P, P['cache'], b_norm, self.loss_kind, opt, events=inner)
            gbufs.append(gsend)
            if m == 0 and after_row_requests is not None:
                # the run-ahead stages are ISSUED here — while the training stream has the forward / backward and the reduction of
                # this batch queued: the host time they take is hidden behind those kernels; on the communicator they sit in front
                # of this step's gradient exchange
                after_row_requests()
            for c in range(Cn):
                if grecv[c] is None:                       # one receive buffer per chunk for all micro-batches: segments stay adjacent
                    grecv[c] = torch.empty(max(32, sum(n_recv[c])), dtype=gsend.dtype, device=gsend.device)
                off = sum(n_recv[c][:m])
                so = int(P['cache_off'][c])                # (the gradient buffer has the cache's geometry: chunk c's units at the same place)
                pushed[c][m] = self._a2a(gsend[so:], ops.xsplits(P['send_counts'][c]), ops.xsplits(P['recv_counts'][c]),
                                         out=grecv[c][off:off + max(n_recv[c][m], 1)], overlap=ov)[1]
        if inner is None:
            rec(2); rec(3); rec(4)
        # ---- owner side, chunk by chunk: apply what has arrived, then answer the next step's requests for the same key range
        head = Ps[0]
        nxt = None
        if callable(next_prepared):                    # (the pipeline: the next step's keys are exchanged INSIDE this call — looked up now)
            next_prepared = next_prepared()
        if next_prepared is not None:
            nxt = list(next_prepared) if isinstance(next_prepared, (list, tuple)) else [next_prepared]
            if not nxt or 'table' not in nxt[0]:       # (not ready: the next call fetches its rows itself)
                nxt = None
            else:
                self._main_waits(nxt)
        loss = None
        for c in range(Cn):
            for h in pushed[c]:
                h.wait()
            own = None
            if getattr(ops, 'self_bypass', False):
                # the rank's own pieces: behind all the units that travel, in chunk order
                own = []
                for g, P in zip(gbufs, Ps):
                    base = int(P['cache_off'][Cn])
                    own.append((g, base + sum(chunk_floats([P['send_counts'][cc][self.rank]], self.ld)[0] for cc in range(c))))
            out = ops.apply(head['req_all'][c], grecv[c], head['counts_all'][c], head['table'], b_norm, opt,
                            want_loss=want_loss and c == Cn - 1, own=own, chunk=c)
            if c == Cn - 1:
                loss = out
            if nxt is not None:
                self._fetch_chunk(nxt, c, True)
        rec(5)
        for P in Ps:                                    # (buffers of a finished step: the pipeline reuses the dict's prepared buffer)
            for k_ in ('cache', 'rx', 'cache_off'):
                P.pop(k_, None)
        if after_apply is not None:
            after_apply()
        if want_loss:
            self.last_loss = float(loss[0].item()) if torch.is_tensor(loss) else float(loss)
            return self.last_loss
        return None


class ShardedPipeline:
    """Drives ShardedCdae so that no step waits on the host or on a parameter-independent exchange.

    Iteration s issues, in this program order (identical on every rank — one communicator):
        step(batch s) up to its forward / backward + reduction  ·  exchange_keys + index_owner(batch s+1)  ·
        exchange_counts(batch s+3)  ·  gradient exchange, owner apply and — chunk by chunk — gather + row exchange of batch s+1  ·
        prepare(batch s+4)
    The run-ahead exchanges travel while step s computes; the count exchange of batch s+1 was issued two iterations before iteration s
    reads its host-side result: the host never waits for it.  On a GPU the run-ahead stages use a side stream; on CPU (gloo tests)
    everything runs inline in the same order.  `batch_of(s)` must return the DrxBatch (or the list of micro-batches) of step s and be
    callable LOOKAHEAD steps ahead."""

    LOOKAHEAD = 4

    def __init__(self, model, batch_of, n_steps, use_side_stream=None):
        self.m, self.batch_of, self.n = model, batch_of, n_steps
        eng = model.engine
        self.cuda = eng is not None and torch.device(eng.device).type == 'cuda'
        if use_side_stream is None:
            use_side_stream = self.cuda
        self.main = torch.cuda.current_stream(eng.device) if self.cuda else None
        model._main = self.main
        # High priority: ROCm multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) and a stream that
        # lands on the queue of the training or the RCCL stream inherits their barriers (measured: the count exchange then
        # completes only when the GPU drains, +0.2 ms/step); priority streams get queues of their own.
        from .engine import run_ahead_stream              # (the process-wide pool: streams probed for a hardware queue of their own)
        self.side = run_ahead_stream(eng.device, 0) if (self.cuda and use_side_stream) else None
        self.P = {}
        self.next = 0
        self.host_s = [0.0, 0.0, 0.0]
        # prepared buffers are reused LOOKAHEAD + 2 steps later: step s's buffer is rewritten by the preparation of step
        # s + RING, issued during iteration s + RING - LOOKAHEAD >= s + 2, behind the event recorded after step s
        self.RING = self.LOOKAHEAD + 2
        self.bufs = {}
        self.done = {}
        if self.side is not None:
            self.side.wait_stream(self.main)            # histories / batches set up on the training stream
        # Staggered run-ahead: a collective queued on the communicator must only depend on work that finished long ago —
        # it sits in FRONT of the current step's gradient exchange there, and would hold it up while, say, a sort that was
        # queued a moment ago is still running.  So batch s+4 is prepared, batch s+3 exchanges counts, batch s+1 keys.
        with self._on_side():
            for s in range(min(self.LOOKAHEAD, n_steps)):
                self._prepare(s)
            for s in range(min(self.LOOKAHEAD - 1, n_steps)):
                self._counts(s)
            self._keys(0)

    def _on_side(self):
        return _SideStream(self.m, self.side) if self.side is not None else _Null()

    def _micro(self, s):
        bt = self.batch_of(s)
        return list(bt) if isinstance(bt, (list, tuple)) else [bt]

    # (the three stages below run with the side stream current: _on_side())
    def _prepare(self, s):
        k = s % self.RING
        if self.side is not None and k in self.done:
            self.side.wait_event(self.done[k])             # the step that last read this slot's buffers has finished
        out = []
        for i, bt in enumerate(self._micro(s)):
            buf = self.bufs.get((k, i)) if self.cuda else None
            P = self.m.prepare(bt, consumer_stream=self.main if self.side is not None else None, out=buf)
            if self.cuda:
                self.bufs[(k, i)] = P['buf']
            out.append(P)
        self.P[s] = out

    def _counts(self, s):
        for P in self.P[s]:
            self.m.exchange_counts(P)

    def _keys(self, s):
        for P in self.P[s]:
            self.m.exchange_keys(P, solo=len(self.P[s]) == 1)
        self.m.index_owner(self.P[s], slot=s % self.RING)

    def run_step(self, events=None, want_loss=False):
        s = self.next
        assert s < self.n
        t0 = time.perf_counter()
        ahead_s = [0.0]

        def run_ahead():                       # queued while this step's forward / reduction are on the training stream
            ta = time.perf_counter()
            with self._on_side():
                if s + 1 < self.n:
                    self._keys(s + 1)
                if s + self.LOOKAHEAD - 1 < self.n:
                    self._counts(s + self.LOOKAHEAD - 1)
            ahead_s[0] += time.perf_counter() - ta

        def prepare_ahead():                   # no collective in it: issued when the whole step is queued
            ta = time.perf_counter()
            if s + self.LOOKAHEAD < self.n:
                with self._on_side():
                    self._prepare(s + self.LOOKAHEAD)
            ahead_s[0] += time.perf_counter() - ta
        # (step s + 1's keys are exchanged inside this call, by run_ahead: its prepared micro-batches are looked up when the tail starts)
        out = self.m.step(s, self._micro(s), events=events, want_loss=want_loss, prepared=self.P.pop(s),
                          after_row_requests=run_ahead, after_apply=prepare_ahead,
                          next_prepared=(lambda: self.P.get(s + 1)) if s + 1 < self.n else None)
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.done[s % self.RING] = ev
        t3 = time.perf_counter()
        self.host_s[1] += ahead_s[0]; self.host_s[2] += t3 - t0 - ahead_s[0]                 # host time issuing each part
        if hasattr(self.batch_of, 'release'):
            self.batch_of.release(s)                   # e.g. engine.DeviceBatchSource: the batch's buffers may be reused
        self.next = s + 1
        return out


class _SideStream:
    """Makes `side` torch's current stream for the duration of the block and tells the model which stream is current (event records
    without the current-stream lookup).  torch.cuda.set_stream on the way in and out: the `torch.cuda.stream(...)` context manager
    resolves devices and the previous stream through several Python layers — 25 - 30 us per block, three blocks per step (r05:
    scripts/rows_host_profile.py)."""

    def __init__(self, model, side):
        self.model, self.side = model, side

    def __enter__(self):
        self.prev = torch.cuda.current_stream(self.side.device)      # (whatever the caller had current: engine._on_stream does the same)
        torch.cuda.set_stream(self.side)
        self.model._cur = self.side
        return self

    def __exit__(self, *a):
        self.model._cur = None
        torch.cuda.set_stream(self.prev)
        return False


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class ColumnShardedCdae:
    """One rank of the COLUMN-sharded ("K-sharded") sampled-mode CDAE: every rank holds all rows but only the columns
    [k_lo, k_hi) of W, W2T, V and b (b2 is replicated) and every rank trains on the SAME global batch (identical sampler and
    mask seeds).  A step is the single-GPU step on K/N columns with ONE exchange: the all-reduce of the per-triple partial
    dot products h_b . W2T[i_b] (B floats).  Nothing else travels, so unlike the row-sharded layout it does not depend on
    the point-to-point xGMI bandwidth between a PAIR of GPUs (one link), which bounds row-sharding at 2 and 4 GPUs.
    The result equals the single-process step on the global batch with all K columns (the dot product is summed in rank
    order); histories (the positives CSR of ALL users) are replicated."""

    TURNS_AHEAD = 3       # prepare='turns': a list is built three steps before its step and broadcast one step before it

    def __init__(self, n_users, n_items, k, rank, world, device, hist_indptr, hist_indices, seed=10, lr=0.05, reg=1e-3,
                 optimizer='adagrad', group=None, loss='bce', q=0.2, cpu_staging=False, force_collectives=False, engine=None,
                 prepare='local'):
        self.rank, self.world, self.group = rank, world, group
        # who builds the sorted touch list of a step (the same list on every rank):
        #   'local'  every rank builds all of it;   'turns'  rank s % N builds the list of step s and broadcasts it;
        #   'parts'  every rank builds 1/N of every list, one all-gather
        assert prepare in ('local', 'turns', 'parts')
        self.prepare_mode = prepare
        self._all, self._oflow, self._oflow_ev = [None, None], [None, None], [None, None]
        self._oflow_host = None
        self._prep_group = None
        self.cpu_staging = cpu_staging
        self.collectives = world > 1 or force_collectives
        self.k = k
        kpr = -(-k // world)
        self.k_lo, self.k_hi = min(k, rank * kpr), min(k, (rank + 1) * kpr)
        assert self.k_hi > self.k_lo, 'more ranks than columns'
        self.loss, self.q = loss, q
        if engine is not None:            # tests: a NumPy statement of the two per-rank halves (tests/test_dist_gloo.py)
            self.engine = engine
            return
        from .engine import CdaeEngine
        self.engine = e = CdaeEngine(n_users, n_items, self.k_hi - self.k_lo, device=device)
        e.share_users = False          # (the forward half here is a kernel of its own: plain lists)
        e.set_history(hist_indptr, hist_indices)
        e.init_optimizer(optimizer, lr, reg)
        # GlorotUniform of the GLOBAL shapes; this rank's columns from its own stream, the replicated b2 from a shared one
        gen = torch.Generator(device=e.device); gen.manual_seed(int(seed) * 1000003 + rank)
        kl = self.k_hi - self.k_lo
        for t, lim in ((e.W, math.sqrt(6.0 / (n_items + k))), (e.W2T, math.sqrt(6.0 / (k + n_items))), (e.V, math.sqrt(6.0 / (n_users + k)))):
            t.zero_()
            t[:, :kl] = (torch.rand(t.shape[0], kl, generator=gen, device=e.device) * 2 - 1) * lim
        e.b.zero_(); e.b[:kl] = (torch.rand(kl, generator=gen, device=e.device) * 2 - 1) * math.sqrt(3.0 / k)
        gb = torch.Generator(device=e.device); gb.manual_seed(int(seed))
        e.b2.copy_((torch.rand(n_items, generator=gb, device=e.device) * 2 - 1) * math.sqrt(3.0 / n_items))

    def set_params_global(self, W, W_, V, b, b_):
        """Slices global (reference-orientation) weights into this rank's columns."""
        lo, hi = self.k_lo, self.k_hi
        self.engine.set_params(W=np.asarray(W)[:, lo:hi], W_=np.asarray(W_)[lo:hi, :], V=np.asarray(V)[:, lo:hi], b=np.asarray(b)[lo:hi],
                               b_=np.asarray(b_))

    def get_params(self):
        return self.engine.get_params()

    def gather_params_global(self):
        """All K columns on every rank (reference orientation: W [N,K], W_ [K,N], V [U,K], b [K], b_ [N]) — for evaluation or
        export; an infrequent host-side gather, not part of a step."""
        mine = self.get_params()
        if not self.collectives:
            return mine
        if not all(isinstance(mine[k], np.ndarray) for k in ('W', 'W_', 'V', 'b')) or not hasattr(self, 'k_hi'):
            parts = [None] * self.world
            dist.all_gather_object(parts, mine, group=self.group)
            return {'W': np.concatenate([p['W'] for p in parts], axis=1), 'W_': np.concatenate([p['W_'] for p in parts], axis=0),
                    'V': np.concatenate([p['V'] for p in parts], axis=1), 'b': np.concatenate([p['b'] for p in parts]), 'b_': parts[0]['b_']}
        # tensors, not pickled objects (a rank's column slice of V is 640 MB at BASELINE configuration 4): the column axis first, padded
        # to the widest slice (the last rank's may be narrower)
        Wd = self.world
        kpr = -(-self.k // Wd)
        cols = [min(self.k, (r + 1) * kpr) - min(self.k, r * kpr) for r in range(Wd)]
        dev = torch.device(self.engine.device) if dist.get_backend(self.group) == 'nccl' else torch.device('cpu')
        out = {'b_': mine['b_']}
        for name in ('W', 'W_', 'V', 'b'):
            a = np.ascontiguousarray(mine[name].T if name in ('W', 'V') else mine[name])        # [columns, ...]
            t = torch.zeros((max(cols),) + a.shape[1:], dtype=torch.as_tensor(a[:0]).dtype, device=dev)
            t[:a.shape[0]].copy_(torch.as_tensor(a))
            parts = [torch.empty_like(t) for _ in range(Wd)]
            dist.all_gather(parts, t, group=self.group)
            full = torch.cat([p[:cols[r]] for r, p in enumerate(parts)], dim=0).cpu().numpy()
            out[name] = np.ascontiguousarray(full.T) if name in ('W', 'V') else full
        return out

    def step(self, step, bt, prepared=None, events=None, want_loss=False):
        e = self.engine
        h, dot = e.kshard_forward(bt, prepared) if hasattr(e, '_params') else e.kshard_forward(bt)
        if self.collectives:
            if self.cpu_staging:
                d = dot.cpu()
                dist.all_reduce(d, group=self.group)
                dot = d.to(dot.device)
            else:
                dist.all_reduce(dot, group=self.group)
        out = e.step_sparse(step, bt, self.loss, want_loss=want_loss, events=events, prepared=prepared, kshard=(h, dot))
        return float(out[0]) if want_loss else None

    def _prep_comm(self):
        """A communicator of their own for the exchanges of prepared lists: they are long and run ahead, and must never queue in
        front of a step's all-reduce."""
        if self._prep_group is None:
            self._prep_group = dist.new_group(ranks=dist.get_process_group_ranks(self.group) if self.group is not None else None)
        return self._prep_group

    def build_in_turns(self, s, bt, out=None):
        """First half of prepare='turns': rank s % world builds the sorted touch list of step s (the others only make sure
        they have a buffer for it).  A rank thus sorts one list in `world` steps instead of one per step."""
        e = self.engine
        return e.prepare_sparse(bt, out) if s % self.world == self.rank else e.prep_buffer(bt, out)

    def deliver_in_turns(self, s, bt, out):
        """Second half: the list of step s travels from its builder to everybody — the leading drx_cdae_prep_result_bytes of
        the prepared buffer, broadcast as they are.  Issued a step before the list is used and long after it was built, so
        that the collective never waits for a sort."""
        if not self.collectives:
            return
        e = self.engine
        owner = s % self.world
        res = out[:e.prep_result_bytes(bt)]
        src = dist.get_global_rank(self.group, owner) if self.group is not None else owner
        if self.cpu_staging:
            # (debugging aid: ranks sharing one GPU, exchange staged through the host.  The list was built on the side stream and this
            # stream waited for it through a drx event — an agent-scope release, enough for the kernels of an RCCL broadcast but not
            # for a copy engine reading the buffer from memory: a device-wide synchronise makes the bytes visible to it.  Without it
            # this path handed both ranks a torn list about once in six runs of tests/test_gpu_kshard.py — r03n.)
            torch.cuda.synchronize(e.device)
            h = res.cpu()
            dist.broadcast(h, src=src, group=self._prep_comm())
            if owner != self.rank:
                res.copy_(h)
        else:
            dist.broadcast(res, src=src, group=self._prep_comm())

    def prepare_in_turns(self, s, bt, out=None):
        """Both halves at once, on the CURRENT stream (inline stepping; pipeline() issues them separately)."""
        out = self.build_in_turns(s, bt, out)
        self.deliver_in_turns(s, bt, out)
        return out

    def prepare(self, s, bt, out=None):
        """The sorted touch list of the (global) batch `bt`, built in PARTS: this rank sorts the touches of the rows it owns
        (row id % world == rank), the parts travel in one all-gather on a communicator of their own (so that this long
        exchange never queues in front of a step's all-reduce) and every rank assembles the same list.  Runs on the CURRENT
        stream (the pipeline's side stream).  A part that does not fit its fixed capacity is reported two calls later."""
        e = self.engine
        k = s % 2
        if self._oflow_host is None:
            self._oflow_host = [torch.zeros(1, dtype=torch.int32, pin_memory=True) for _ in range(2)]
        self._check_overflow(k)
        part = e.prepare_part(bt, self.rank, self.world, slot=k)
        if self.collectives:
            need = part.numel() * self.world
            if self._all[k] is None or self._all[k].numel() < need:
                if self._all[k] is not None:
                    e._retire(self._all[k])
                self._all[k] = None
                self._all[k] = torch.empty(int(need * 1.05) + 4096, dtype=torch.uint8, device=part.device)
            allp = self._all[k][:need]
            if self.cpu_staging:
                hp = part.cpu()
                ha = torch.empty(part.numel() * self.world, dtype=torch.uint8)
                dist.all_gather_into_tensor(ha, hp, group=self._prep_comm())
                allp.copy_(ha)
            else:
                dist.all_gather_into_tensor(allp, part, group=self._prep_comm())
        else:
            allp = part
        out, self._oflow[k] = e.prepare_assemble(bt, allp, self.world, out, self._oflow[k])
        self._oflow_host[k].copy_(self._oflow[k], non_blocking=True)
        self._oflow_ev[k] = torch.cuda.Event()
        self._oflow_ev[k].record(torch.cuda.current_stream(e.device))
        return out

    def _check_overflow(self, k):
        if self._oflow_ev[k] is not None:
            self._oflow_ev[k].synchronize()                  # recorded two preparations ago
            if int(self._oflow_host[k][0]):
                raise RuntimeError('a part of the touch list exceeded its capacity (row popularity too uneven for parts of 1.25x the '
                                   "even share): train with prepare='local' or 'turns'")

    def pipeline(self, batch_size, neg_ratio, sample_seed_of, mask_seed_of):
        """SampledPipeline over this rank: the SAME seeds on every rank give every rank the same global batch."""
        from .engine import SampledPipeline
        return SampledPipeline(self.engine, batch_size, neg_ratio, self.q, sample_seed_of, mask_seed_of, n_items=self.engine.n_items,
                               loss=self.loss, step_fn=lambda s, bt, prep, events, want_loss: self.step(s, bt, prepared=prep, events=events,
                                                                                                         want_loss=want_loss),
                               prepare_fn={'local': None, 'turns': self.build_in_turns, 'parts': self.prepare}[self.prepare_mode],
                               deliver_fn=self.deliver_in_turns if self.prepare_mode == 'turns' else None,
                               prep_ahead=self.TURNS_AHEAD if self.prepare_mode == 'turns' else 1,
                               # a rank's streams share four hardware queues (engine._run_ahead_slot): training, torch.distributed's own,
                               # and two more — two run-ahead streams, or one and the deliveries' stream
                               side_streams=1 if self.prepare_mode == 'turns' else 2)

