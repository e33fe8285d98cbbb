"""Row-sharded CDAE across the GPUs of one node: one process per GPU, `torch.distributed` over
RCCL/xGMI (backend "nccl"); the same orchestration runs over gloo on CPU tensors in the world-size-2 tests, with the
per-rank device work swapped for a NumPy statement (tests/dist_ops_numpy.py).

Sharding
  * users (V rows + optimizer state, their positives CSR, and the triples sampled for them): contiguous uid ranges —
    the 5 GB table of the 10M-user configuration never leaves its GPU and needs no exchange;
  * item rows (W, W2T, b2 + state): contiguous item ranges of `ipr = ceil(N / world)` rows.
  * hidden bias b (K floats): replicated, its gradient all-reduced.

One step (every rank, its own B triples; losses/L2 are normalised by the GLOBAL batch so the step equals a single-GPU
step on the concatenated batch):
  1. touches of the local batch -> stable sort -> DISTINCT row keys (owner-major key space: a rank's distinct keys are
     contiguous per owner) -> per-owner counts                                       [drx_shard_touches / _index]
  2. all-to-all counts, all-to-all(v) of the requested keys                          [xGMI, 4 B per distinct row]
  3. owners gather the requested rows; all-to-all(v) of rows (+ output biases)       [xGMI, 4K B per distinct row]
  4. forward/backward against the row cache                                          [drx_shard_fwd_bwd]
  5. local segmented reduction: ONE gradient row per distinct item row (hot Zipf rows are merged before they travel);
     V rows are updated in place                                                     [drx_shard_reduce]
  6. all-to-all(v) of the gradient rows back to the owners                           [xGMI, 4K B per distinct row]
  7. owners sum duplicates across ranks in rank order and apply sparse Adagrad/Adam  [drx_shard_apply]
  8. all-reduce of the K-float bias gradient (+ loss), dense update of b.
Direct all-to-all drives all 7 xGMI links of a GPU at once (ring all-reduce would be bound by one link), and only
distinct rows travel.
"""
import ctypes as C
import math
import time

import numpy as np
import torch
import torch.distributed as dist


def items_per_rank(n_items, world):
    return (n_items + world - 1) // world


def item_key(n, ipr, is_out):
    """owner-major row key of item n (numpy / python ints)"""
    o = n // ipr
    return o * 2 * ipr + (ipr if is_out else 0) + (n - o * ipr)


class HipShardOps:
    """Per-rank device work through the C ABI (include/drx.h, drx_shard_*)."""

    def __init__(self, n_users_local, n_items, k, rank, world, device, optimizer, lr, reg):
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
        self.shard = _lib.Shard(world, rank, n_items, self.ipr, n_users_local)
        self._scratch = {}

    # -- helpers
    def _sc(self, n, stage='step'):
        """Scratch of one stage: `index` may run ahead on a side stream while reduce/apply of an earlier batch run on the
        training stream, so the two never share a buffer."""
        need = self.L.drx_shard_scratch_bytes(C.byref(self.engine._params), C.byref(self.shard), int(n))
        cur = self._scratch.get(stage)
        if cur is None or cur.numel() < need:
            cur = self._scratch[stage] = torch.empty(int(need * 1.3) + 4096, dtype=torch.uint8, device=self.device)
        return cur

    def _e(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def _stream(self):
        return self._lib.stream_ptr(self.device)

    # -- ops
    def touches(self, bt):
        T = bt.n_touch_slots + 2 * bt.B
        keys, vals, bpos = (self._e(T, dtype=torch.int32) for _ in range(3))
        p = self._lib.ptr
        self._lib.check(self.L.drx_shard_touches(C.byref(self.shard), C.byref(self.engine._hist), C.byref(bt), p(keys), p(vals),
                                                 p(bpos), self._stream()), 'drx_shard_touches')
        return keys, vals, bpos

    def index(self, keys, vals):
        T = keys.numel()
        ks, vs, ss, sp, uk = (self._e(T, dtype=torch.int32) for _ in range(5))
        bounds = self._e(self.world + 2, dtype=torch.int32)
        sc = self._sc(T, 'prepare')
        p = self._lib.ptr
        self._lib.check(self.L.drx_shard_index(C.byref(self.engine._params), C.byref(self.shard), p(keys), p(vals), T, p(ks),
                                               p(vs), p(ss), p(sp), p(uk), p(bounds), p(sc), sc.numel(), self._stream()),
                        'drx_shard_index')
        host = torch.empty(self.world + 2, dtype=torch.int32, pin_memory=True)
        host.copy_(bounds, non_blocking=True)          # read by bounds_of() after the stream (or its event) is reached
        return {'keys_s': ks, 'vals_s': vs, 'slot_sorted': ss, 'slot_of_pos': sp, 'uniq_keys': uk, 'bounds_dev': bounds,
                'bounds_host': host}

    @staticmethod
    def bounds_of(idx, event=None):
        if 'bounds' not in idx:
            (event.synchronize() if event is not None else torch.cuda.current_stream().synchronize())
            idx['bounds'] = idx['bounds_host'].tolist()
        return idx['bounds']

    def gather_rows(self, req):
        n = req.numel()
        rows, b2v = self._e(max(n, 1), self.ld), self._e(max(n, 1))
        p = self._lib.ptr
        self._lib.check(self.L.drx_shard_gather_rows(C.byref(self.engine._params), C.byref(self.shard), p(req), n, p(rows),
                                                     p(b2v), self._stream()), 'drx_shard_gather_rows')
        return rows[:n], b2v[:n]

    def fwd_bwd(self, bt, slot_of_pos, rows_cache, b2_cache, b_norm, loss_kind):
        """Returns the per-sample gradient buffers of this (micro-)batch: the context reduce() and bias_grad() consume."""
        B = bt.B
        ctx = {'dz1': self._e(B, self.ld), 'g2': self._e(B, self.ld), 'dz2': self._e(B), 'lossb': self._e(B)}
        if rows_cache.numel() == 0:
            rows_cache, b2_cache = self._e(1, self.ld), self._e(1)
        p = self._lib.ptr
        self._lib.check(self.L.drx_shard_fwd_bwd(C.byref(self.engine._params), C.byref(self.engine._hist), C.byref(bt),
                                                 p(slot_of_pos), p(rows_cache), p(b2_cache), b_norm, loss_kind, p(ctx['dz1']),
                                                 p(ctx['g2']), p(ctx['dz2']), p(ctx['lossb']), self._stream()), 'drx_shard_fwd_bwd')
        return ctx

    def reduce(self, idx, bpos, q_item, b_norm, q, opt, ctx):
        T = idx['keys_s'].numel()
        gc, gb2c = self._e(max(q_item, 1), self.ld), self._e(max(q_item, 1))
        sc = self._sc(T)
        p = self._lib.ptr
        self._lib.check(self.L.drx_shard_reduce(C.byref(self.engine._params), C.byref(opt), C.byref(self.shard), b_norm,
                                                float(q), p(idx['keys_s']), p(idx['vals_s']), p(idx['slot_sorted']), p(bpos),
                                                T, p(ctx['dz1']), p(ctx['g2']), p(ctx['dz2']), p(gc), p(gb2c), p(sc), sc.numel(),
                                                self._stream()), 'drx_shard_reduce')
        return gc[:q_item], gb2c[:q_item]

    def apply(self, recv_keys, recv_rows, recv_b2, b_norm, opt, recv_counts):
        n = recv_keys.numel()
        if n == 0:
            return
        sc = self._sc(n)
        p = self._lib.ptr
        counts = (C.c_int32 * len(recv_counts))(*[int(c) for c in recv_counts])      # world x micro-batches segments
        self._lib.check(self.L.drx_shard_apply(C.byref(self.engine._params), C.byref(opt), C.byref(self.shard), b_norm,
                                               p(recv_keys), p(recv_rows.contiguous()), p(recv_b2.contiguous()), n, counts,
                                               len(recv_counts), p(sc), sc.numel(), self._stream()), 'drx_shard_apply')

    def bias_grad(self, B, ctx):
        out = self._e(self.ld + 1)
        sc = self._sc(1)
        p = self._lib.ptr
        self._lib.check(self.L.drx_shard_bias_grad(C.byref(self.engine._params), p(ctx['dz1']), p(ctx['lossb']), B, p(out), p(sc),
                                                   sc.numel(), self._stream()), 'drx_shard_bias_grad')
        return out

    def bias_apply(self, grad, b_norm, opt):
        g = grad[:self.ld].contiguous()
        self._lib.check(self.L.drx_shard_bias_apply(C.byref(self.engine._params), C.byref(opt), b_norm, self._lib.ptr(g),
                                                    self._stream()), 'drx_shard_bias_apply')

    def optim(self, step):
        a = self.engine.adam_alpha(self.engine.lr, step + 1, self.engine.beta1, self.engine.beta2)
        return self.engine._optim([a] * 5)

    def make_batch(self, *a, **k):
        return self.engine.make_batch(*a, **k)

    def set_params(self, W, W_, V, b, b_):
        self.engine.set_params(W=W, W_=W_, V=V, b=b, b_=b_)

    def get_params(self):
        return self.engine.get_params()


class ShardedCdae:
    """One rank of the row-sharded sampled-mode CDAE.  `ops` = per-rank compute backend (HipShardOps on a GPU)."""

    def __init__(self, n_users_total, n_items, k, rank, world, device, hist_indptr, hist_indices, seed=10, lr=0.05, reg=1e-3,
                 optimizer='adagrad', ops=None, group=None, loss='bce', q=0.2, cpu_staging=False, force_collectives=False):
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
        self.ops = ops if ops is not None else HipShardOps(n_local, n_items, k, rank, world, device, optimizer, lr, reg)
        self.engine = getattr(self.ops, 'engine', None)
        self.loss_kind = 0 if loss == 'bce' else 1
        self.q = q
        if self.engine is not None:
            self.engine.set_history(hist_indptr, hist_indices)
            self._init_random(seed)
        self.last_loss = None
        self.wait_s = 0.0                     # host time spent waiting for count exchanges (should stay ~0 when pipelined)

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

    # ---- exchanges ---------------------------------------------------------------------------------------
    def _a2a(self, send, send_counts, recv_counts):
        shape = (int(sum(recv_counts)),) + tuple(send.shape[1:])
        out = torch.empty(shape, dtype=send.dtype, device=send.device)
        if not self.collectives:
            out.copy_(send[:shape[0]])
            return out
        if self.cpu_staging and send.is_cuda:
            o = torch.empty(shape, dtype=send.dtype)
            dist.all_to_all_single(o, send.contiguous().cpu(), output_split_sizes=list(recv_counts),
                                   input_split_sizes=list(send_counts), group=self.group)
            return o.to(send.device)
        dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(recv_counts),
                               input_split_sizes=list(send_counts), group=self.group)
        return out

    def _a2a_start(self, send, send_counts, recv_counts, out=None, overlap=False):
        """all-to-all(v) that the caller waits for later: returns (received tensor, wait()-able or None).  The training stream
        keeps running kernels of another micro-batch while the rows travel."""
        n = int(sum(recv_counts))
        if out is None:
            out = torch.empty((n,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if not self.collectives:
            out.copy_(send[:n])
            return out, None
        if (self.cpu_staging and send.is_cuda) or not send.is_cuda:
            out.copy_(self._a2a(send, send_counts, recv_counts))
            return out, None
        # async_op only when there is another micro-batch to compute meanwhile: on one micro-batch the asynchronous form was
        # measured slower (1-rank RCCL: 1.15 vs 1.05 ms/step)
        work = dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(recv_counts),
                                      input_split_sizes=list(send_counts), group=self.group, async_op=overlap)
        return out, (work if overlap else None)

    def _counts(self, send_counts, device):
        if not self.collectives:
            return list(send_counts)
        s = torch.tensor(send_counts, dtype=torch.int64, device='cpu' if self.cpu_staging else device)
        r = torch.empty_like(s)
        dist.all_to_all_single(r, s, group=self.group)
        return r.cpu().tolist()

    # ---- one step ------------------------------------------------------------------------------------------
    # A step has three parameter-independent stages that may run ahead of the training stream (ShardedPipeline below):
    #   prepare(bt)            local: touches, stable sort, distinct keys, slots, per-owner bounds      [no collective]
    #   exchange_counts(P)     all-to-all of the per-owner distinct-key counts; the result goes to pinned host memory
    #   exchange_keys(P)       all-to-all(v) of the distinct keys each owner is asked for (needs the counts on the host)
    # and the parameter-dependent rest in step().  Every rank must call the stages in the same program order: they all
    # run on one communicator.
    def prepare(self, bt, consumer_stream=None):
        """Parameter-independent, collective-free part of a step: touches of the local batch, stable sort, distinct
        keys, slots, per-owner bounds.  May run on a side stream for a later batch while the current one trains;
        `consumer_stream` is the stream that will later read the result (allocator bookkeeping)."""
        keys, vals, bpos = self.ops.touches(bt)
        idx = self.ops.index(keys, vals)
        ev = None
        if torch.is_tensor(keys) and keys.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            if consumer_stream is not None:
                for t in [bpos] + [v for v in idx.values() if torch.is_tensor(v) and v.is_cuda]:
                    t.record_stream(consumer_stream)
        return {'idx': idx, 'bpos': bpos, 'event': ev, 'keys': keys, 'vals': vals, 'consumer': consumer_stream}

    def exchange_counts(self, P):
        """Stage 2: how many distinct keys every rank asks of every owner.  On the device path nothing here waits on the
        host: the send counts are differences of the device-side bounds, the received counts land in pinned memory."""
        idx, W = P['idx'], self.world
        if 'bounds_dev' in idx and not self.cpu_staging:
            bd = idx['bounds_dev']
            send = (bd[1:W + 1] - bd[0:W]).to(torch.int64)
            recv = torch.empty_like(send)
            if self.collectives:
                dist.all_to_all_single(recv, send, group=self.group)
            else:
                recv.copy_(send)
            host = torch.empty(2, W, dtype=torch.int64, pin_memory=True)
            host[0].copy_(send, non_blocking=True)
            host[1].copy_(recv, non_blocking=True)
            P['counts_host'] = host
            P['event'] = torch.cuda.Event()
            P['event'].record()
            if P.get('consumer') is not None:
                send.record_stream(P['consumer']); recv.record_stream(P['consumer'])
        else:
            bounds = self.ops.bounds_of(idx, P['event']) if hasattr(self.ops, 'bounds_of') else idx['bounds']
            send_counts = [bounds[o + 1] - bounds[o] for o in range(W)]
            P['send_counts'] = send_counts
            P['recv_counts'] = self._counts(send_counts, idx['uniq_keys'].device)
        return P

    def exchange_keys(self, P):
        """Stage 3: every owner learns which of its rows each rank wants (4 B per distinct row)."""
        if 'send_counts' not in P:
            if 'counts_host' not in P:
                self.exchange_counts(P)
            if 'counts_host' in P:
                t0 = time.perf_counter()
                P['event'].synchronize()                # the tiny count exchange, issued at least one step earlier
                self.wait_s += time.perf_counter() - t0
                P['send_counts'] = P['counts_host'][0].tolist()
                P['recv_counts'] = P['counts_host'][1].tolist()
        q_item = int(sum(P['send_counts']))
        P['q_item'] = q_item
        P['req'] = self._a2a(P['idx']['uniq_keys'][:q_item], P['send_counts'], P['recv_counts'])
        if torch.is_tensor(P['req']) and P['req'].is_cuda:
            P['event'] = torch.cuda.Event()
            P['event'].record()
            if P.get('consumer') is not None:
                P['req'].record_stream(P['consumer'])
        return P

    def step(self, step, bt, events=None, want_loss=False, prepared=None, after_row_requests=None):
        """One training step of this rank.  `bt` is a DrxBatch or a list of up to DRX_MAX_MICRO micro-batches with pairwise
        DISJOINT users (then `prepared` is the matching list): all micro-batches read the pre-step parameters and every
        owned row is updated once with the sum of their gradients — the step still equals the single-process step on the
        concatenated batch — but the row exchange of micro-batch m+1 and the gradient exchange of micro-batch m travel while
        the other one computes.
        Event slots (bench): [0,1) row gather + the first row exchange, [1,2) forward/backward + local reduce of all
        micro-batches (later row / earlier gradient exchanges hidden behind them), [2,3) rest of the gradient exchange,
        [3,4) owner apply, [4,5) bias.
        after_row_requests: called once the row exchanges of this step are queued — ShardedPipeline queues the run-ahead
        count / key exchanges of later batches there, so that on the communicator they sit behind this step's row exchange
        (and travel while it computes) instead of in front of it."""
        ops = self.ops
        bts = list(bt) if isinstance(bt, (list, tuple)) else [bt]
        Ps = (list(prepared) if isinstance(prepared, (list, tuple)) else [prepared]) if prepared is not None else [None] * len(bts)
        b_norm = sum(b.B for b in bts) * self.world
        opt = ops.optim(step)
        rec = (lambda i: events[i].record()) if events is not None else (lambda i: None)
        wait = lambda w: w.wait() if w is not None else None
        rec(0)
        for m, b in enumerate(bts):
            P = Ps[m] = Ps[m] if Ps[m] is not None else self.prepare(b)
            if 'req' not in P:
                self.exchange_keys(P)
            elif P['event'] is not None:
                torch.cuda.current_stream().wait_event(P['event'])
        fetched = []
        for P in Ps:                                   # owners answer every micro-batch's request from the pre-step tables
            rows, b2v = ops.gather_rows(P['req'])
            fetched.append((self._a2a_start(rows, P['recv_counts'], P['send_counts'], overlap=len(Ps) > 1),
                            self._a2a_start(b2v, P['recv_counts'], P['send_counts'], overlap=len(Ps) > 1)))
        if after_row_requests is not None:
            after_row_requests()
        wait(fetched[0][0][1]); wait(fetched[0][1][1])
        rec(1)
        n_recv = [int(sum(P['recv_counts'])) for P in Ps]
        rg = rb2 = None
        pushed, ctxs, off = [], [], 0
        for m, (b, P) in enumerate(zip(bts, Ps)):
            (rows_cache, w1), (b2_cache, w2) = fetched[m]
            wait(w1); wait(w2)
            ctx = ops.fwd_bwd(b, P['idx']['slot_of_pos'], rows_cache, b2_cache, b_norm, self.loss_kind)
            gc, gb2c = ops.reduce(P['idx'], P['bpos'], P['q_item'], b_norm, b.q, opt, ctx)
            if rg is None:                             # one receive buffer for all micro-batches: segments stay adjacent
                rg = torch.empty((sum(n_recv),) + tuple(gc.shape[1:]), dtype=gc.dtype, device=gc.device)
                rb2 = torch.empty(sum(n_recv), dtype=gb2c.dtype, device=gb2c.device)
            ov = len(Ps) > 1
            pushed.append((self._a2a_start(gc, P['send_counts'], P['recv_counts'], out=rg[off:off + n_recv[m]], overlap=ov)[1],
                           self._a2a_start(gb2c, P['send_counts'], P['recv_counts'], out=rb2[off:off + n_recv[m]], overlap=ov)[1], gc, gb2c))
            ctxs.append(ctx)
            off += n_recv[m]
        rec(2)
        for w1, w2, _, _ in pushed:
            wait(w1); wait(w2)
        rec(3)
        req = Ps[0]['req'] if len(Ps) == 1 else torch.cat([P['req'] for P in Ps])
        ops.apply(req, rg, rb2, b_norm, opt, [c for P in Ps for c in P['recv_counts']])
        rec(4)
        gb = ops.bias_grad(bts[0].B, ctxs[0])
        for b, ctx in zip(bts[1:], ctxs[1:]):
            gb = gb + ops.bias_grad(b.B, ctx)
        if self.collectives:
            if self.cpu_staging and gb.is_cuda:
                g = gb.cpu()
                dist.all_reduce(g, group=self.group)
                gb = g.to(gb.device)
            else:
                dist.all_reduce(gb, group=self.group)
        ops.bias_apply(gb, b_norm, opt)
        rec(5)
        if want_loss:
            self.last_loss = float(gb[-1].item()) / b_norm
            return self.last_loss
        return None


class ShardedPipeline:
    """Drives ShardedCdae so that no step waits on the host or on a parameter-independent exchange.

    Iteration s issues, in this program order (identical on every rank — one communicator):
        step(batch s) up to its row exchange  ·  exchange_keys(batch s+1)  ·  exchange_counts(batch s+2)  ·
        prepare(batch s+3)  ·  rest of step(batch s)
    The run-ahead exchanges sit behind step s's row exchange on the communicator and travel while step s computes; the
    count exchange of batch s+2 has long finished when iteration s+1 reads its host-side result.  On a GPU the run-ahead stages use a side stream; on CPU (gloo tests) everything runs inline in the
    same order.  `batch_of(s)` must return the DrxBatch (or the list of micro-batches) of step s and be callable three
    steps ahead."""

    LOOKAHEAD = 3

    def __init__(self, model, batch_of, n_steps, use_side_stream=None):
        self.m, self.batch_of, self.n = model, batch_of, n_steps
        eng = model.engine
        self.cuda = eng is not None and torch.device(eng.device).type == 'cuda'
        if use_side_stream is None:
            use_side_stream = self.cuda
        self.main = torch.cuda.current_stream(eng.device) if self.cuda else None
        # High priority: ROCm multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) and a stream that
        # lands on the queue of the training or the RCCL stream inherits their barriers (measured: the count exchange then
        # completes only when the GPU drains, +0.2 ms/step); priority streams get queues of their own.
        self.side = torch.cuda.Stream(eng.device, priority=-1) if (self.cuda and use_side_stream) else None
        self.P = {}
        self.next = 0
        self.host_s = [0.0, 0.0, 0.0]
        if self.side is not None:
            self.side.wait_stream(self.main)            # histories / batches set up on the training stream
        # Staggered run-ahead: a collective queued on the communicator must only depend on work that finished long ago —
        # it sits in FRONT of the current step's gradient exchange there, and would hold it up while, say, a sort that was
        # queued a moment ago is still running.  So batch s+3 is indexed, batch s+2 exchanges counts, batch s+1 keys.
        for s in range(min(3, n_steps)):
            self._prepare(s)
        for s in range(min(2, n_steps)):
            self._counts(s)
        self._keys(0)

    def _on_side(self):
        return torch.cuda.stream(self.side) if self.side is not None else _Null()

    def _micro(self, s):
        bt = self.batch_of(s)
        return list(bt) if isinstance(bt, (list, tuple)) else [bt]

    def _prepare(self, s):
        with self._on_side():
            self.P[s] = [self.m.prepare(bt, consumer_stream=self.main if self.side is not None else None) for bt in self._micro(s)]

    def _counts(self, s):
        with self._on_side():
            for P in self.P[s]:
                self.m.exchange_counts(P)

    def _keys(self, s):
        with self._on_side():
            for P in self.P[s]:
                self.m.exchange_keys(P)

    def run_step(self, events=None, want_loss=False):
        s = self.next
        assert s < self.n
        t0 = time.perf_counter()
        ahead_s = [0.0]

        def run_ahead():                       # queued behind this step's row exchange on the communicator
            ta = time.perf_counter()
            if s + 1 < self.n:
                self._keys(s + 1)
            if s + 2 < self.n:
                self._counts(s + 2)
            if s + 3 < self.n:
                self._prepare(s + 3)
            ahead_s[0] = time.perf_counter() - ta
        out = self.m.step(s, self._micro(s), events=events, want_loss=want_loss, prepared=self.P.pop(s),
                          after_row_requests=run_ahead)
        t3 = time.perf_counter()
        self.host_s[1] += ahead_s[0]; self.host_s[2] += t3 - t0 - ahead_s[0]                 # host time issuing each part
        if hasattr(self.batch_of, 'release'):
            self.batch_of.release(s)                   # e.g. engine.DeviceBatchSource: the batch's buffers may be reused
        self.next = s + 1
        return out


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
        parts = [None] * self.world
        dist.all_gather_object(parts, mine, group=self.group)
        return {'W': np.concatenate([p['W'] for p in parts], axis=1), 'W_': np.concatenate([p['W_'] for p in parts], axis=0),
                'V': np.concatenate([p['V'] for p in parts], axis=1), 'b': np.concatenate([p['b'] for p in parts]), 'b_': parts[0]['b_']}

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
                               prep_ahead=self.TURNS_AHEAD if self.prepare_mode == 'turns' else 1)

