"""DMF (Deep Matrix Factorization) on the MI355X engine — constructor, hooks and semantics of
DRecPy/Recommender/dmf.py; arithmetic in drx_dmf.hip / drx_generic.hip behind include/drx.h.

Reference behaviour kept: PointSampler triples, inputs = RAW interaction row / column l2-normalised (dmf.py:75-86),
targets standardised to [0,1] when use_nce with min_interaction forced to 0 when the data minimum is 1
(recommender_abc.py:141,463-465), cosine clipped at 1e-6, Keras BCE, l2(reg_rate) on the Dense kernels, dense Keras Adam
with one apply per tower per step, `_predict` rescaled to the interaction range (dmf.py:101-106).
Additions: `_rank` scores all candidates of a user in ONE launch instead of one `_predict` per item
(recommender_abc.py:460), and `score_matrix(user_ids)` scores users against all items on the matrix cores (bf16).
"""
from heapq import nlargest

import numpy as np

from .recommender_abc import RecommenderABC
from ..Sampler import PointSampler


class DMF(RecommenderABC):
    _host_prefetch = True      # fit() draws batch t+1 on a worker thread while batch t trains (sampler-only, engine-free hook)
    _prefetch_from = 2048      # ... for batches of at least this many samples (smaller ones: drawn inline, recommender_abc.fit)
    # from this batch size on, the host-side preparation of a batch (distinct ids, groupings) runs in _do_batch on the main thread instead
    # of behind the draw on the worker: at B = 4096 the reference-exact draw (0.29 ms) plus the preparation (0.09 ms) made the worker the
    # slowest stage of fit()
    _PREP_ON_MAIN_FROM = int(__import__('os').environ.get('DRX_DMF_PREP_ON_MAIN_FROM', 2048))

    def __init__(self, user_factors=None, item_factors=None, use_nce=True, l2_norm_vectors=True, device='cuda:0', **kwds):
        super().__init__(**kwds)
        self.user_factors = [64, 32] if user_factors is None else user_factors
        assert type(self.user_factors) is list, 'The "user_factors" argument must be of type list (ex: [64, 32]).'
        assert len(self.user_factors) > 0, 'The "user_factors" argument must have at least 1 element.'
        self.item_factors = [64, 32] if item_factors is None else item_factors
        assert type(self.item_factors) is list, 'The "item_factors" argument must be of type list (ex: [64, 32]).'
        assert len(self.item_factors) > 0, 'The "item_factors" argument must have at least 1 element.'
        assert self.user_factors[-1] == self.item_factors[-1], \
            f'The last user and item factors dimension must be equal ({self.user_factors[-1]} != {self.item_factors[-1]})'
        # the reference accepts any factor lists (dmf.py:22-44); here the hidden units of a layer are the lanes of one wavefront, two
        # units per lane at most, and a tower is unrolled over at most 4 layers (examples/consistency_eval/dmf.py:20 builds [128, 64]):
        # say so at construction, not at the first step
        for name, f in (('user_factors', self.user_factors), ('item_factors', self.item_factors)):
            if len(f) > 4 or any((not isinstance(x, (int, np.integer))) or x < 1 or x > 128 for x in f):
                raise Exception(f'drecpy_amd.DMF supports towers of 1..4 layers of width 1..128 (given: {name}={f}).')
        self.use_nce = use_nce
        self.l2_norm_vectors = l2_norm_vectors
        self.device = device

    def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):          # dmf.py:46-62
        from ..engine_dmf import DmfEngine
        ds = self.interaction_dataset
        self._engine = DmfEngine(self.n_users, self.n_items, self.user_factors, self.item_factors, self.l2_norm_vectors,
                                 device=self.device)
        self._engine.lr, self._engine.reg = float(learning_rate), float(reg_rate)
        self._engine.set_interactions(ds.interaction_csr(), ds.interaction_csr(transpose=True))
        weights = kwds.get('initial_weights')
        if weights is None:
            weights = self._keras_init(np.random.default_rng(self.seed))
        self._engine.set_params(weights)
        self.user_nn, self.item_nn = self._engine.user_nn, self._engine.item_nn
        self._register_trainables([self.user_nn, self.item_nn])               # dmf.py:60
        self._sampler = PointSampler(ds, neg_ratio, self.interaction_threshold, self.seed)
        # fit(..., device_sampler=True): THROUGHPUT MODE, a named deviation like CDAE's and Caser's — the triples are drawn on the
        # device by a counter-based generator with the reference sampler's distribution (include/drx.h drx_point_sample_valued:
        # negatives among the pairs absent from the frame, positives user-uniform, carrying the pair's interaction value) and the
        # batch's distinct ids are numbered there too (drx_dmf_batch_distinct_device), one step ahead on a side stream, instead of
        # the reference's three MT19937 streams on one host thread (0.07 us per triple: at B = 4096 that alone is 1.5 device steps).
        # Duplicate (user, item) rows of the frame count as one pair with their values summed (the interaction matrix's view).
        # The default stays the reference-exact stream.
        self._device_sampler = bool(kwds.get('device_sampler', False))
        if self._device_sampler:
            self._setup_device_sampler(ds, neg_ratio)
        else:
            self.__dict__.pop('_host_prefetch', None)

    def _restore_engine(self, params):
        """RecommenderABC.load (recommender_abc.py:517-524): the engine rebuilt from the saved weights and the saved dataset.  A subclass
        that bound a prediction scale in its `_pre_fit` (ModifiedDMF) binds it again in its own override before calling this."""
        from ..engine_dmf import DmfEngine
        ds = self.interaction_dataset
        self._engine = DmfEngine(self.n_users, self.n_items, self.user_factors, self.item_factors, self.l2_norm_vectors, device=self.device)
        self._engine.set_interactions(ds.interaction_csr(), ds.interaction_csr(transpose=True))
        self._engine.set_params({k: v for k, v in params.items() if k != 'extra_w'})
        self.user_nn, self.item_nn = self._engine.user_nn, self._engine.item_nn

    def _setup_device_sampler(self, ds, neg_ratio):
        import torch
        indptr, cols, vals = ds.interaction_csr()
        thr = self.interaction_threshold
        keep = np.ones(len(vals), bool) if thr is None else (np.asarray(vals) >= thr)
        if keep.all():
            positives, recorded = (indptr, cols, vals), None
        else:
            rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))[keep]
            ip = np.zeros(len(indptr), dtype=np.int64)
            ip[1:] = np.cumsum(np.bincount(rows, minlength=len(indptr) - 1))
            positives, recorded = (ip, np.asarray(cols)[keep], np.asarray(vals)[keep]), (indptr, cols)
        vmin, vrange = (self.min_interaction, self.max_interaction - self.min_interaction) if self.use_nce else (0.0, 0.0)
        self._engine.set_sampler_frame(positives, recorded, vmin, vrange)
        self._host_prefetch = False                  # nothing to draw on a host thread
        self._neg_ratio = int(neg_ratio)
        self._dev_draws, self._dev_next = 0, None
        from ..engine import run_ahead_stream
        self._dev_side = run_ahead_stream(self._engine.device, 0)          # (the process-wide pool: a stream with a hardware queue of its own)
        self._dev_side.wait_stream(torch.cuda.current_stream(self._engine.device))       # the frame uploaded a moment ago
        self._dev_done = {}                          # ring slot -> event recorded behind the step that read it
        self._sampler_kind = 'device PointSampler (drx_point_sample_valued, counter-based; throughput mode)'

    def _draw_device(self, batch_size):
        """The next batch, drawn and prepared on the side stream (it waits only for the step that last read the ring slot it writes)."""
        import torch
        eng = self._engine
        k = eng._dev_i % 3
        if k in self._dev_done:
            self._dev_side.wait_event(self._dev_done[k])
        self._dev_draws += 1
        seed = (int(self.seed) if self.seed is not None else 0) * 1000003 + self._dev_draws
        from ..engine import _on_stream                 # (`with torch.cuda.stream(...)` without most of its Python layers)
        with _on_stream(self._dev_side):
            prep = eng.prepare_batch_device(batch_size, self._neg_ratio, seed)
            prep['ready'] = torch.cuda.Event()
            prep['ready'].record(self._dev_side)
        prep['slot'] = k
        return prep

    def _fused_trainables(self):
        e = self._engine
        return [e.user_nn, e.item_nn] + ([e.scale_var] if e.scale_var is not None else [])

    def _configure_optimizer(self):
        o = self.optimizer
        if getattr(o, 'kind', None) != 'adam':
            raise Exception(f'DMF trains with Keras Adam only (dense update of both towers); got {o!r}')
        e = self._engine
        e.lr, e.beta1, e.beta2, e.eps = o.learning_rate, o.beta_1, o.beta_2, o.epsilon

    def _keras_init(self, rng):
        p = {}
        for tower, n_in, factors in (('u', self.n_items, self.user_factors), ('i', self.n_users, self.item_factors)):
            prev = n_in
            for l, f in enumerate(factors):
                lim = np.sqrt(6.0 / (prev + f))
                p[f'{tower}{l}_k'] = rng.uniform(-lim, lim, size=(prev, f)).astype(np.float32)
                p[f'{tower}{l}_b'] = np.zeros(f, np.float32)
                prev = f
        return p

    def _sample_batch(self, batch_size, **kwds):                          # dmf.py:64-73
        if getattr(self, '_device_sampler', False):
            nxt = self._dev_next
            cur = nxt if (nxt is not None and nxt['B'] == batch_size) else self._draw_device(batch_size)
            # the batch after this one is drawn NOW, while this one trains (never beyond the last epoch)
            self._dev_next = self._draw_device(batch_size) if kwds.get('more_to_come', False) else None
            return ('device-batch', cur)
        u, i, v, _ = self._sampler.sample_arrays(batch_size)
        y = np.asarray(self._standardize_value(v) if self.use_nce else v, dtype=np.float32)
        # (u, i, y) plus the engine's host-side preparation of the batch (distinct users / items): this hook runs on fit()'s
        # sampler thread while the previous batch trains
        if len(u) >= self._PREP_ON_MAIN_FROM:        # large batches: the draw alone fills the worker (0.07 us per triple on one thread)
            return u, i, y
        return u, i, y, self._engine.prepare_batch(u, i, y)

    def _do_batch(self, batch_samples, step=0, want_loss=False, **kwds):
        e = self._engine
        if isinstance(batch_samples, tuple) and len(batch_samples) == 2 and batch_samples[0] == 'device-batch':
            import torch
            prep = batch_samples[1]
            main = torch.cuda.current_stream(e.device)
            main.wait_event(prep['ready'])
            applies = (len(self._apply_order()), self._apply_position(e.user_nn), self._apply_position(e.item_nn),
                       self._apply_position(e.scale_var) if e.scale_var is not None else None)
            out = e.step(step, prep, want_loss=want_loss, applies=applies)
            ev = torch.cuda.Event()
            ev.record(main)
            self._dev_done[prep['slot']] = ev
            return out
        prep = batch_samples[3] if len(batch_samples) > 3 else e.prepare_batch(*batch_samples[:3])
        # one apply_gradients per registered item, in registration-list order (recommender_abc.py:194-196,328-334)
        applies = (len(self._apply_order()), self._apply_position(e.user_nn), self._apply_position(e.item_nn),
                   self._apply_position(e.scale_var) if e.scale_var is not None else None)
        return e.step(step, prep, want_loss=want_loss, applies=applies)

    def _predict_batch(self, batch_samples, **kwds):
        """max(1e-6, cosine) of the batch pairs (dmf.py:88-96) as a device array [B] — WITHOUT any bound prediction scale: a
        subclass that registered one multiplies here itself, like the reference's ModifiedDMF._predict_batch."""
        u, i, y = batch_samples[:3]
        with self._device_lock:
            return self._engine.predict(u, i, scaled=False), y

    def _compute_batch_loss(self, predictions, desired_values, **kwds):
        import torch
        p = predictions.double()
        t = torch.as_tensor(np.asarray(desired_values), dtype=torch.float64, device=p.device)
        eps = 1e-7
        pc = p.clamp(eps, 1 - eps)
        return float((-(t * torch.log(pc + eps) + (1 - t) * torch.log(1 - pc + eps))).mean().item())

    @staticmethod
    def _as_array(preds):
        """Predictions of _predict_batch -> numpy [B]: a device array, or (subclass overrides) a list of scalars / (1,)-arrays."""
        import torch
        if isinstance(preds, (list, tuple)):
            preds = torch.stack([torch.as_tensor(p).reshape(-1)[0] for p in preds]) if len(preds) else torch.zeros(0)
        return torch.as_tensor(preds).detach().reshape(-1).cpu().numpy()

    def _predict(self, uid, iid, **kwds):                                 # dmf.py:101-106: through _predict_batch, like the reference
        preds, _ = self._predict_batch((np.array([uid]), np.array([iid]), None))
        return self._rescale_value(float(self._as_array(preds)[0]))

    def _rank(self, uid, iids, n, novelty):
        if novelty:
            rated = self.interaction_dataset.select(f'uid == {uid}').values_list('iid', to_list=True)
            iids = set(iids).difference(set(rated))
        iids = sorted(set(int(i) for i in iids))
        if not iids:
            return []
        preds = self._as_array(self._predict_batch((np.full(len(iids), uid), np.asarray(iids), None))[0])
        return nlargest(n, [(self._rescale_value(float(p)), i) for p, i in zip(preds, iids)])

    def score_matrix(self, user_ids):
        """[len(user_ids), n_items] clipped cosine scores via the bf16 MFMA scorer (raw user ids in)."""
        uids = [self.interaction_dataset.user_to_uid(u) for u in user_ids]
        with self._device_lock:
            return self._engine.score_matrix_bf16(np.asarray(uids))
