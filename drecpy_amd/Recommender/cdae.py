"""CDAE (Collaborative Denoising Auto-Encoder) on the MI355X engine — constructor, hooks and semantics of the reference
model (DRecPy/Recommender/cdae.py), arithmetic in hand-written HIP kernels behind the C ABI of include/drx.h.

mode='reference' (default) reproduces the reference step exactly: every sampled triple contributes its USER only
(cdae.py:52), the loss runs over ALL output units against the batch-mean target ((B,B,N) Keras broadcast,
cdae.py:78-79), L2/B over the full W, W_, V (cdae.py:81-82), dense Keras Adam with one apply per variable
(recommender_abc.py:328-334), corruption drawn from MT19937 seeded like RecommenderABC._rng, N draws per row
(cdae.py:63) — the stream is generated in C++ (drx_rng_corruption_keep) and is bit-identical to `random.Random(seed)`.

mode='sampled' is the engine's throughput mode (CDAE paper's negative sampling, which the reference skips, cdae.py:5):
the sampled (uid, iid, value) triple selects ONE output unit with target 1[value >= threshold]; sparse Adagrad (or lazy
Adam) touches only the rows involved; corruption comes from a counter-based mask evaluated on the device.
"""
import numpy as np

from .recommender_abc import RecommenderABC
from ..Sampler import PointSampler


class CDAE(RecommenderABC):
    fused_fit = True      # reference mode: fit() runs its quiet loop inside the library (_run_steps); False = always step by step

    def __init__(self, hidden_factors=50, corruption_level=0.2, loss='bce', mode='reference', loss_targets='reference',
                 sparse_optimizer='adagrad', device_sampler=False, device='cuda:0', layout='rows', exchange_chunks=2, exchange_transport='rccl', **kwds):
        super().__init__(**kwds)
        if layout not in ('rows', 'columns'):
            raise Exception(f'Unknown multi-GPU layout "{layout}" (supported: "rows", "columns").')
        # how a sampled, device-sampled fit() under a torch.distributed process group shards the model (drecpy_amd/dist.py):
        #   'rows'    north_star / BASELINE configuration 4 — V rows, histories and samples by user range, item rows by item range, the
        #             rows a batch needs and their merged gradients exchanged by all-to-all(v) in `exchange_chunks` pipelined chunks;
        #             every rank draws `batch_size` triples of ITS users per step (global batch = world x batch_size);
        #   'columns' every rank all rows x K / world columns, the same global batch on every rank, one all-reduce of B floats.
        # exchange_transport (rows): 'rccl' = the library's own RCCL communicator, the exchanges of a step issued from C
        #             (include/drx.h drx_comm_*, drx_shard_phase_*) — under an "nccl" process group, where librccl opens on every rank;
        #             'torch' = torch.distributed.all_to_all_single call by call (always under gloo).
        if exchange_transport not in ('rccl', 'torch'):
            raise Exception(f'Unknown exchange transport "{exchange_transport}" (supported: "rccl", "torch").')
        self.layout, self.exchange_chunks, self.exchange_transport = layout, int(exchange_chunks), exchange_transport
        self.hidden_factors = hidden_factors
        self.corruption_level = corruption_level
        if loss not in ('mse', 'bce'):
            raise Exception(f'Loss function "{loss}" is not supported. Supported losses: "mse", "bce".')
        if mode not in ('reference', 'sampled'):
            raise Exception(f'Unknown mode "{mode}" (supported: "reference", "sampled").')
        self._loss_name = loss
        self.mode = mode
        self.loss_targets = loss_targets
        self.sparse_optimizer = sparse_optimizer
        self.device_sampler = device_sampler      # sampled mode: draw the triples on the GPU (counter-based stream)
        self.device = device

    # ---- multi-GPU: the same fit() under an initialised torch.distributed process group ------------------------------------
    def _world(self):
        """(rank, world) when this fit() is one process of a multi-GPU job, else None.  Sampled mode with the device sampler
        trains sharded (drecpy_amd/dist.py: ShardedCdae for layout='rows', ColumnShardedCdae for 'columns'): every process is given
        the SAME dataset and seed; afterwards every process assembles the whole model and predicts / ranks on its own like a
        single-GPU model."""
        import torch.distributed as dist
        if not (self.mode == 'sampled' and self.device_sampler and dist.is_available() and dist.is_initialized()):
            return None
        return (dist.get_rank(), dist.get_world_size()) if dist.get_world_size() > 1 else None

    def fit(self, interaction_dataset, epochs=50, batch_size=32, learning_rate=0.001, neg_ratio=5, reg_rate=0.001,
            copy_dataset=False, **kwds):
        self._fit_epochs = int(epochs)         # (the row layout's pipeline issues exchanges for the steps to come: it must know the last one)
        out = super().fit(interaction_dataset, epochs=epochs, batch_size=batch_size, learning_rate=learning_rate, neg_ratio=neg_ratio,
                          reg_rate=reg_rate, copy_dataset=copy_dataset, **kwds)
        if getattr(self, '_dist_model', None) is not None:      # training is over: every rank keeps the whole model
            from ..engine import CdaeEngine
            full = self._dist_model.gather_params_global()
            self._dist_model = self._pipeline = None
            self._engine = CdaeEngine(self.n_users, self.n_items, self.hidden_factors, device=self.device)
            self._engine.set_params(**full)
            self._engine.set_history(self._hist_indptr, self._hist_indices)
        return out

    def _pre_fit_rows(self, rank, world, learning_rate, neg_ratio, reg_rate, **kwds):
        """layout='rows' (recommender_abc.py:97-98 is the surface; north_star's row-wise shard behind it): this rank keeps the V rows,
        histories and recorded pairs of users [lo, hi) and the item rows of its item range; dist.ShardedCdae + ShardedPipeline train."""
        import torch.distributed as dist
        from ..dist import ShardedCdae
        ds = self.interaction_dataset
        self._hist_indptr, self._hist_indices = ds.positives_csr(self.interaction_threshold)
        seed = self.seed if self.seed is not None else 0
        U = self.n_users
        lo, hi = U * rank // world, U * (rank + 1) // world
        ip = np.asarray(self._hist_indptr, np.int64)
        m = ShardedCdae(U, self.n_items, self.hidden_factors, rank, world, self.device, ip[lo:hi + 1] - ip[lo],
                        self._hist_indices[ip[lo]:ip[hi]], seed=seed, lr=learning_rate, reg=reg_rate, optimizer=self.sparse_optimizer,
                        loss=self._loss_name, q=self.corruption_level, cpu_staging=(dist.get_backend() == 'gloo'),
                        chunks=self.exchange_chunks,
                        transport='rccl' if (self.exchange_transport == 'rccl' and dist.get_backend() == 'nccl') else None)
        weights = kwds.get('initial_weights')
        n_params = (2 * self.n_items + U) * self.hidden_factors
        if weights is None and n_params <= (1 << 26):                # the single-GPU initialisation, sliced (larger: drawn per shard on the device)
            weights = self._glorot_like_single_gpu(seed)
        if weights is not None:
            m.set_params_global(**weights)
        self._dist_model, self._engine = m, m.engine
        self._row_range = (lo, hi)
        all_ip, all_cols, _ = ds.interaction_csr()                  # (negatives avoid every recorded pair: see _pre_fit)
        if len(all_cols) != len(self._hist_indices):
            all_ip = np.asarray(all_ip, np.int64)
            m.engine.set_recorded_pairs(all_ip[lo:hi + 1] - all_ip[lo], all_cols[all_ip[lo]:all_ip[hi]])
        else:
            m.engine.set_recorded_pairs(None, None)
        self._pipeline = self._pending = None
        self._register_tables()
        self._sampler = PointSampler(ds, neg_ratio, self.interaction_threshold, self.seed)
        self._mask_seed = int(self.seed if self.seed is not None else 0)
        self._mask_rng = None

    def _glorot_like_single_gpu(self, seed):
        rng = np.random.default_rng(seed)

        def glorot(shape):
            fi, fo = (shape[0], shape[0]) if len(shape) == 1 else (shape[0], shape[1])
            lim = np.sqrt(6.0 / (fi + fo))
            return rng.uniform(-lim, lim, size=shape).astype(np.float32)
        k = self.hidden_factors
        return dict(W=glorot((self.n_items, k)), W_=glorot((k, self.n_items)), V=glorot((self.n_users, k)), b=glorot((k,)),
                    b_=glorot((self.n_items,)))

    def _pre_fit_distributed(self, rank, world, learning_rate, neg_ratio, reg_rate, **kwds):
        import torch.distributed as dist
        from ..dist import ColumnShardedCdae
        if kwds.get('epoch_callback_fn') is not None or kwds.get('early_stopping_rule') is not None:
            raise Exception('epoch callbacks / early stopping need the whole model at every call: not available while it is sharded')
        if self.layout == 'rows':
            return self._pre_fit_rows(rank, world, learning_rate, neg_ratio, reg_rate, **kwds)
        ds = self.interaction_dataset
        self._hist_indptr, self._hist_indices = ds.positives_csr(self.interaction_threshold)
        seed = self.seed if self.seed is not None else 0
        m = ColumnShardedCdae(self.n_users, self.n_items, self.hidden_factors, rank, world, self.device, self._hist_indptr,
                              self._hist_indices, seed=seed, lr=learning_rate, reg=reg_rate, optimizer=self.sparse_optimizer,
                              loss=self._loss_name, q=self.corruption_level, cpu_staging=(dist.get_backend() == 'gloo'),
                              prepare=kwds.get('prepare', 'turns' if world >= 4 else 'local'))     # who sorts the touch lists: dist.py
        weights = kwds.get('initial_weights')
        n_params = (2 * self.n_items + self.n_users) * self.hidden_factors
        if weights is None and n_params <= (1 << 26):                # the single-GPU initialisation, sliced
            weights = self._glorot_like_single_gpu(seed)
        if weights is not None:
            m.set_params_global(**weights)
        self._dist_model, self._engine = m, m.engine
        all_ip, all_cols, _ = ds.interaction_csr()                  # (negatives avoid every recorded pair: see _pre_fit)
        m.engine.set_recorded_pairs(*((all_ip, all_cols) if len(all_cols) != len(self._hist_indices) else (None, None)))
        self._pipeline = self._pending = None
        self._register_tables()
        self._sampler = PointSampler(ds, neg_ratio, self.interaction_threshold, self.seed)
        self._mask_seed = int(self.seed if self.seed is not None else 0)
        self._mask_rng = None

    def _register_tables(self):
        """cdae.py:35-43: the five variables, registered in the order W, W_, V, b, b_ (their Adam counters: t = 5*step + j + 1).
        The handles are views over the engine's tables (W_ is stored transposed, one output unit per row)."""
        from .trainables import Variable
        e = self._engine
        self.W, self.W_, self.V, self.b, self.b_ = (Variable.over(t, n) for t, n in zip(e.tables(), ('W', 'W_', 'V', 'b', 'b_')))
        self._register_trainables([self.W, self.W_, self.V, self.b, self.b_])

    def _fused_trainables(self):
        return [self.W, self.W_, self.V, self.b, self.b_]

    def _configure_optimizer(self):
        """Reference mode trains with Keras Adam (the default registered by fit(), or an optimizers.Adam passed as `optimizer=`);
        sampled mode with the constructor's `sparse_optimizer` unless `optimizer=` forces another kind."""
        o, e = self.optimizer, self._engine
        if self.mode == 'reference':
            if getattr(o, 'kind', None) != 'adam':
                raise Exception(f'CDAE mode="reference" is the reference step: Keras Adam only (got {o!r}); Adagrad variants exist in mode="sampled"')
            kind = 'adam'
        else:
            if not self._optimizer_forced:
                return
            kind = o.kind
        if getattr(self, '_dist_model', None) is not None:
            if self._optimizer_forced:
                raise Exception('optimizer= cannot be forced on a column-sharded fit(): choose it with CDAE(sparse_optimizer=...)')
            return
        if kind == 'adam':
            same = e.s2 is not None and (e.lr, e.beta1, e.beta2, e.opt_eps) == (o.learning_rate, o.beta_1, o.beta_2, o.epsilon)
            if not same:                                   # (the slots _pre_fit allocated are kept when nothing differs)
                e.init_optimizer('adam', o.learning_rate, e.reg_rate, o.beta_1, o.beta_2, o.epsilon)
        else:
            e.init_optimizer(kind, o.learning_rate, e.reg_rate, eps=o.epsilon, initial_accumulator=o.initial_accumulator_value)

    # ---- cdae.py:34-45 ---------------------------------------------------------------------------
    def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
        from .. import _lib
        from ..engine import CdaeEngine
        ds = self.interaction_dataset
        self._dist_model = None
        world = self._world()
        if world is not None:
            return self._pre_fit_distributed(world[0], world[1], learning_rate, neg_ratio, reg_rate, **kwds)
        self._engine = CdaeEngine(self.n_users, self.n_items, self.hidden_factors, device=self.device)
        self._pipeline = None
        self._close_drawahead()                                      # (draws an early-stopped fit() left in flight included)
        self._pending = None
        weights = kwds.get('initial_weights')
        if weights is not None:                      # injected weights (TF's GlorotUniform stream is not reproducible)
            self._engine.set_params(**weights)
        else:
            seed = self.seed if self.seed is not None else np.random.SeedSequence().entropy % (2 ** 32)
            n_params = (2 * self.n_items + self.n_users) * self.hidden_factors
            # large tables are drawn on the device (same distribution, torch's generator instead of numpy's)
            (self._engine.init_glorot_device if n_params > (1 << 26) else self._engine.init_glorot)(seed)
        self._hist_indptr, self._hist_indices = ds.positives_csr(self.interaction_threshold)
        self._max_degree = int(np.diff(self._hist_indptr).max()) if len(self._hist_indptr) > 1 else 0
        self._engine.set_history(self._hist_indptr, self._hist_indices)
        # the device PointSampler draws negatives among the pairs ABSENT from the frame (point_sampler.py:56): where the frame records
        # pairs below the threshold too, it needs the CSR of every recorded pair
        all_ip, all_cols, _ = ds.interaction_csr()
        self._engine.set_recorded_pairs(*((all_ip, all_cols) if len(all_cols) != len(self._hist_indices) else (None, None)))
        if self.mode == 'reference':
            self._engine.init_optimizer('adam', learning_rate, reg_rate)
        else:
            self._engine.init_optimizer(self.sparse_optimizer, learning_rate, reg_rate)
        self._register_tables()
        self._sampler = PointSampler(ds, neg_ratio, self.interaction_threshold, self.seed)
        L = _lib.lib()
        seed = self.seed if self.seed is not None else self._rng.getrandbits(62)
        self._mask_rng = L.drx_rng_create(int(seed))    # same MT19937 stream as self._rng = random.Random(seed)
        # a second generator on the same stream: inside fit() two worker threads draw batches t+1 and t+2 at once, each from
        # its own generator advanced to where its batch begins (a batch always consumes 2·N·B words: cdae.py:63 draws for
        # every item of every row), so the stream is still consumed strictly batch by batch
        self._mask_rngs = [self._mask_rng, L.drx_rng_create(int(seed))]
        self._mask_at = [0, 0]                          # words each generator has consumed
        self._mask_pos = 0                              # where the next batch begins
        self._draw_ticket = 0
        self._L, self._q_float = L, float(self.corruption_level)
        # (reference mode only: the draw-ahead workers need the native host sampler, which a device-sampled fit never builds)
        self._drawahead = None
        if self.mode == 'reference':
            self._drawahead = L.drx_drawahead_create(self._sampler._host._h, self._mask_rngs[0], self._mask_rngs[1],
                                                     self._hist_indptr.ctypes.data, self._hist_indices.ctypes.data, self.n_items)
        self._mask_seed = int(seed)

    def _close_drawahead(self):
        if getattr(self, '_drawahead', None):
            from .. import _lib
            self._drain_draws()
            _lib.lib().drx_drawahead_destroy(self._drawahead)
        self._drawahead = None
        self._pending = None

    def __del__(self):
        try:
            self._close_drawahead()
        except Exception:
            pass

    def _restore_engine(self, params):
        from ..engine import CdaeEngine
        self._engine = CdaeEngine(self.n_users, self.n_items, self.hidden_factors, device=self.device)
        self._engine.set_params(**params)
        ip, idx = self.interaction_dataset.positives_csr(self.interaction_threshold)
        self._hist_indptr, self._hist_indices = ip, idx
        self._engine.set_history(ip, idx)

    def _sample_batch(self, batch_size, **kwds):       # cdae.py:47
        if self.mode == 'sampled' and self.device_sampler:
            # the device sampler runs two batches ahead of the training stream inside SampledPipeline; this hook only
            # makes sure the pipeline exists for this batch size and hands _do_batch a token
            pipe = getattr(self, '_pipeline', None)
            if pipe is None or pipe.B != batch_size:
                from ..engine import SampledPipeline
                first = 0 if pipe is None else pipe.next
                ms = self._mask_seed
                seeds = (lambda s: ms * 7919 + first + s + 1, lambda s: ms + 0x9E3779B9 * (first + s + 1))
                dm = getattr(self, '_dist_model', None)
                if dm is not None and self.layout == 'rows':
                    # every rank draws ITS users' triples: the seeds differ by rank (row_seeds below; tests replay them)
                    from ..dist import ShardedPipeline
                    from ..engine import DeviceBatchSource
                    rs = self.row_seeds(ms, dm.rank, dm.world, first)
                    src = DeviceBatchSource(dm.engine, batch_size, self._sampler.neg_ratio, self.corruption_level, rs[0], rs[1],
                                            n_items=self.n_items, stream_slot=0)      # (the pipeline's own run-ahead stream: engine.DeviceBatchSource)
                    pipe = self._pipeline = ShardedPipeline(dm, src, max(1, self._fit_epochs - first))
                    pipe.B = batch_size
                elif dm is not None:                                     # the same seeds on every rank: the same batches
                    pipe = self._pipeline = dm.pipeline(batch_size, self._sampler.neg_ratio, *seeds)
                else:
                    pipe = self._pipeline = SampledPipeline(self._engine, batch_size, self._sampler.neg_ratio, self.corruption_level,
                                                            *seeds, loss=self._loss_name)
            return ('device-batch', pipe.next)
        if self.mode == 'sampled' and kwds.get('as_arrays', True):
            return self._sampler.sample_arrays(batch_size)      # (uid, iid, value, is_negative) numpy arrays
        if self.mode == 'reference':
            return self._reference_batch(batch_size, kwds.get('batches_after', 1 if kwds.get('more_to_come', False) else 0))
        return self._sampler.sample(batch_size)                  # list of (uid, iid, value) like the reference

    @staticmethod
    def row_seeds(mask_seed, rank, world, first=0):
        """(sample seed, corruption-mask seed) of step s on rank `rank` of a row-sharded fit"""
        return (lambda s: (mask_seed * 7919 + first + s + 1) * world + rank,
                lambda s: (mask_seed + 0x9E3779B9 * (first + s + 1)) * world + rank)

    class _Batch:
        """What _sample_batch hands _do_batch in reference mode: the reference's list of (uid, iid, value) triples — built only
        if somebody looks at it (len / iteration / indexing) — carrying what the fused step needs: the users and the
        corruption stream's keep flags, already in a pinned staging slot of the engine."""

        def __init__(self, arrays, val_type):
            self._arrays, self._val_type, self._list = arrays, val_type, None
            self.uid = self.keep_off = self.keep = self.slot = None

        def _triples(self):
            if self._list is None:
                u, i, v, neg = self._arrays
                vt = self._val_type
                self._list = [(a, b, 0) if ng else (a, b, vt(c)) for a, b, c, ng in zip(u.tolist(), i.tolist(), v.tolist(), neg.tolist())]
            return self._list

        def __len__(self):
            return len(self._arrays[0])

        def __iter__(self):
            return iter(self._triples())

        def __getitem__(self, k):
            return self._triples()[k]

    def _submit_draw(self, batch_size):
        """Claims the next batch's place in both streams (sampler ticket, corruption words) and its buffers — a pinned staging
        slot of the engine for users / offsets / keep flags — and hands the job to the native draw-ahead worker of its
        generator (drx_drawahead_submit).  Returns what _finish_draw needs."""
        from .. import _lib
        B = int(batch_size)
        ticket = self._draw_ticket                      # (the counters move only once the job is really queued: a ticket that
        gen = ticket % 2                                #  was handed out but never submitted would stall both workers)
        at = self._mask_pos
        discard = at - self._mask_at[gen]
        cap = max(B * self._max_degree, 1)
        if self._engine.device.type == 'cuda':
            stage = self._engine.stage_acquire(B, cap)
            extra = self._engine.stage_extra(stage[0])
            ptrs = self._engine.stage_pointers(stage[0])
        else:
            stage = (None, np.empty(B, np.int32), np.empty(B + 1, np.int32), np.empty(cap, np.uint8))
            extra = (np.empty(B, np.int32), np.empty(B, np.float64), np.empty(B, np.uint8))
            ptrs = (stage[1].ctypes.data, extra[0].ctypes.data, extra[1].ctypes.data, extra[2].ctypes.data, stage[2].ctypes.data,
                    stage[3].ctypes.data, len(stage[3]))
        job = self._L.drx_drawahead_submit(self._drawahead, gen, ticket, discard, B, self._q_float, *ptrs)
        if job < 0:
            _lib.check(int(job), 'drx_drawahead_submit')
        self._draw_ticket = ticket + 1
        self._mask_pos = self._mask_at[gen] = at + 2 * self.n_items * B
        return B, gen, int(job), stage, extra

    def _finish_draw(self, entry):
        from .. import _lib
        B, gen, job, stage, extra = entry
        rc = self._L.drx_drawahead_wait(self._drawahead, gen, job)
        if rc:
            _lib.check(rc, 'drx_cdae_reference_draw')
        slot, uid_v, ko_v, kp_v = stage
        batch = CDAE._Batch((uid_v,) + extra, self._sampler._val_type)
        batch.slot, batch.uid, batch.keep_off = slot, uid_v, ko_v
        batch.keep = kp_v[:max(int(ko_v[B]), 1)]
        return batch

    def _drain_draws(self):
        """Waits for (does not consume) the draws in flight: whoever touches the sampler or a generator directly comes after."""
        from .. import _lib
        for B, gen, job, _, _ in (getattr(self, '_pending', None) or []):
            _lib.lib().drx_drawahead_wait(self._drawahead, gen, job)

    _DRAW_AHEAD = 4        # batches in flight: two per worker

    def _reference_batch(self, batch_size, batches_after):
        """Inside fit() the host work of the next batches (sampler triples, N uniform draws per row) runs on the two native
        draw-ahead threads of libdrx while batch t trains; the streams are still consumed strictly batch by batch, and nothing
        is drawn beyond the last epoch."""
        pending = getattr(self, '_pending', None) or []
        self._pending = None
        if pending and pending[0][0] == batch_size:
            batch = self._finish_draw(pending.pop(0))
        else:
            for entry in pending:                                  # (batches of another size were drawn: consumed, like draws)
                self._finish_draw(entry)
            pending = []
            batch = self._finish_draw(self._submit_draw(batch_size))
        while len(pending) < min(self._DRAW_AHEAD, batches_after):
            pending.append(self._submit_draw(batch_size))
        self._pending = pending or None
        return batch

    def _run_steps(self, first_step, n_steps, batch_size, **kwds):
        """fit()'s quiet path in reference mode: the whole sample -> step loop of the remaining epochs in one library call
        (drx_cdae_fit_dense), when these very hooks would have run — a subclass or an instance that replaces _sample_batch or
        _do_batch gets the per-step loop.  Same streams, same arithmetic, same launches as that loop; returns the steps done."""
        if (self.mode != 'reference' or getattr(self, '_dist_model', None) is not None or self._engine.device.type != 'cuda'
                or not getattr(self, 'fused_fit', True) or n_steps < 1 or getattr(self, '_pending', None)):
            return 0
        for hook in ('_sample_batch', '_do_batch', '_reference_batch', '_submit_draw', '_finish_draw'):
            if hook in self.__dict__ or getattr(type(self), hook) is not getattr(CDAE, hook):
                return 0
        B = int(batch_size)
        from ..engine import CdaeEngine
        if max(B * self._max_degree, 1) * CdaeEngine._FIT_SLOTS > (1 << 30):      # (the loop's pinned staging ring would exceed 1 GB)
            return 0
        cursor = np.array([self._draw_ticket, self._mask_pos, self._mask_at[0], self._mask_at[1]], dtype=np.int64)
        try:
            with self._device_lock:
                self._engine.fit_dense(self._drawahead, cursor, B, self.corruption_level, max(B * self._max_degree, 1), first_step,
                                       n_steps, self._loss_name, self.loss_targets)
        finally:                                        # (what the call consumed of the streams, also when it raised)
            self._draw_ticket, self._mask_pos = int(cursor[0]), int(cursor[1])
            self._mask_at = [int(cursor[2]), int(cursor[3])]
        return n_steps

    # ---- fused training step (replaces recommender_abc.py:190-204 for this model) ------------------------------------
    def _batch_arrays(self, batch_samples):
        if isinstance(batch_samples, tuple) and len(batch_samples) == 4 and hasattr(batch_samples[0], 'dtype'):
            return batch_samples[0], batch_samples[1], batch_samples[2]
        u = np.array([s[0] for s in batch_samples], dtype=np.int32)
        i = np.array([s[1] for s in batch_samples], dtype=np.int32)
        v = np.array([float(s[2]) for s in batch_samples], dtype=np.float64)
        return u, i, v

    def _corruption_keep(self, uid, gen=0, at=None, out=None):
        """MT19937 corruption stream of cdae.py:63 for the batch rows (C++ host, N draws per row, batch order).  gen / at: which
        generator draws and the word of the stream where this batch begins (default: the next unclaimed one)."""
        from .. import _lib
        B = len(uid)
        if at is None:                                   # a draw outside fit()'s run-ahead: after whatever is in flight
            self._drain_draws()
            at, self._mask_pos = self._mask_pos, self._mask_pos + 2 * self.n_items * B
        rng = self._mask_rngs[gen]
        assert at >= self._mask_at[gen], 'corruption stream claimed out of order'
        if at > self._mask_at[gen]:
            _lib.lib().drx_rng_discard(rng, at - self._mask_at[gen])
        self._mask_at[gen] = at + 2 * self.n_items * B
        uid32 = np.ascontiguousarray(uid, dtype=np.int32)
        if out is not None:                               # caller's buffers (a pinned staging slot): keep holds B * max degree
            keep_off, keep = out
        else:
            u64 = uid32.astype(np.int64)
            keep_off = np.zeros(B + 1, dtype=np.int32)
            keep = np.zeros(max(int((self._hist_indptr[u64 + 1] - self._hist_indptr[u64]).sum()), 1), dtype=np.uint8)
        _lib.check(_lib.lib().drx_rng_corruption_keep(
            rng, self._hist_indptr.ctypes.data, self._hist_indices.ctypes.data, self.n_items,
            uid32.ctypes.data, B, float(self.corruption_level), keep_off.ctypes.data, keep.ctypes.data, len(keep)),
            'drx_rng_corruption_keep')
        return keep_off, (keep if out is None else keep[:max(int(keep_off[B]), 1)])

    def _do_batch(self, batch_samples, step=0, want_loss=False, **kwds):
        eng = self._engine
        if self.mode == 'sampled' and self.device_sampler:       # batch drawn and indexed ahead of time on the device
            loss = self._pipeline.run_step(want_loss=want_loss)
            if not want_loss:
                return None
            return float(loss) if isinstance(loss, float) else float(loss[0].item())
        if self.mode == 'reference' and getattr(batch_samples, 'slot', None) is not None:     # already in pinned memory
            bt = eng.batch_in_slot(batch_samples.slot, len(batch_samples.uid), int(batch_samples.keep_off[-1]), self.corruption_level)
            loss = eng.step_dense(step, bt, self._loss_name, self.loss_targets, want_loss=want_loss)
            eng.stage_release(batch_samples.slot)
            return float(loss.sum().item()) if want_loss else None
        if self.mode == 'reference' and getattr(batch_samples, 'keep', None) is not None:
            uid, keep_off, keep = batch_samples.uid, batch_samples.keep_off, batch_samples.keep      # prepared by _sample_batch
            bt, alive = eng.make_batch(uid, keep_off=keep_off, keep=keep, q=self.corruption_level, n_touch_slots=int(keep_off[-1]))
            loss = eng.step_dense(step, bt, self._loss_name, self.loss_targets, want_loss=want_loss)
            return float(loss.sum().item()) if want_loss else None
        uid, iid, val = self._batch_arrays(batch_samples)
        if self.mode == 'reference':
            keep_off, keep = self._corruption_keep(uid)
            bt, alive = eng.make_batch(uid, keep_off=keep_off, keep=keep, q=self.corruption_level,
                                       n_touch_slots=int(keep_off[-1]))
            loss = eng.step_dense(step, bt, self._loss_name, self.loss_targets, want_loss=want_loss)
            return float(loss.sum().item()) if want_loss else None
        y = (val >= self.interaction_threshold).astype(np.float32)
        bt, alive = eng.make_batch(uid, iid, y, q=self.corruption_level, mask_seed=self._mask_seed + 0x9E3779B9 * (step + 1))
        loss = eng.step_sparse(step, bt, self._loss_name, want_loss=want_loss)
        return float(loss[0].item()) if want_loss else None

    # ---- hooks kept for API compatibility (cdae.py:50-82) ---------------------------------------------------
    def _predict_batch(self, batch_samples, **kwds):
        """Training-mode reconstructions of the batch users: (predictions [B,N] device tensor, desired [B,N] uint8)."""
        uid, _, _ = self._batch_arrays(batch_samples)
        keep_off, keep = self._corruption_keep(uid)
        with self._device_lock:
            _, pred = self._engine.forward(uid, keep_off=keep_off, keep=keep, q=self.corruption_level)
        desired = np.zeros((len(uid), self.n_items), dtype=np.uint8)
        for b, u in enumerate(uid):
            desired[b, self._hist_indices[self._hist_indptr[u]:self._hist_indptr[u + 1]]] = 1
        return pred, desired

    def _compute_batch_loss(self, predictions, desired_values, **kwds):
        """Keras BCE / MSE of the reference on the (B,B,N) broadcast == against the batch-mean target (cdae.py:78-79)."""
        # (an API-compatibility hook — the fused step computes the training loss itself: host numpy, no device arithmetic in torch)
        p = np.asarray(predictions.cpu().numpy() if hasattr(predictions, 'cpu') else predictions, dtype=np.float64)
        t = np.asarray(desired_values, dtype=np.float64)
        tbar = t.mean(axis=0, keepdims=True)
        if self._loss_name == 'bce':
            eps = 1e-7
            pc = np.clip(p, eps, 1 - eps)
            return float((-(tbar * np.log(pc + eps) + (1 - tbar) * np.log(1 - pc + eps))).mean())
        return float((((p - tbar) ** 2) + tbar * (1 - tbar)).mean())

    def _compute_reg_loss(self, reg_rate, batch_size, trainable_models=None, trainable_layers=None, trainable_weights=None, **kwds):
        e = self._engine                                   # cdae.py:81-82
        from .. import _lib
        return _lib.sumsq([e.W, e.W2T, e.V]) * 0.5 * reg_rate / batch_size

    # ---- inference (cdae.py:67-71, 84-103) -----------------------------------------------------------------
    def _predict(self, uid, iid=None, **kwds):
        if uid is None:
            return None
        with self._device_lock:
            _, pred = self._engine.forward(np.array([uid], dtype=np.int32))
            row = pred[0].cpu().numpy()
        return row if iid is None else row[iid]

    def _rank(self, uid, iids, n, novelty):
        """Top-n of the candidates by (prediction, iid) — heapq.nlargest order (cdae.py:90-103) — on the device."""
        import torch
        from ..engine import pack_mask_bits
        cand = np.zeros(self.n_items, dtype=bool)
        cand[np.fromiter((int(i) for i in iids), dtype=np.int64)] = True
        if novelty:
            cand[self._all_user_items(uid)] = False        # every (uid, iid) row of the frame, whatever its value
        n_cand = int(cand.sum())
        k = min(int(n), n_cand)
        if k <= 0:
            return []
        with self._device_lock:
            eng = self._engine
            _, pred = eng.forward(np.array([uid], dtype=np.int32))
            mask = torch.as_tensor(pack_mask_bits(cand).view(np.int32)).to(eng.device)
            idx, val = eng.topk(pred, k, mask)
            idx, val = idx[0].cpu().numpy(), val[0].cpu().numpy()
        return [(float(v), int(i)) for v, i in zip(val, idx) if i >= 0]

    def _all_user_items(self, uid):
        if not hasattr(self, '_user_items'):
            ds = self.interaction_dataset
            order = np.argsort(ds._cols['uid'], kind='stable')
            self._ui_sorted = ds._cols['iid'][order].astype(np.int64)
            self._ui_ptr = np.searchsorted(ds._cols['uid'][order], np.arange(self.n_users + 1))
            self._user_items = True
        return self._ui_sorted[self._ui_ptr[uid]:self._ui_ptr[uid + 1]]
