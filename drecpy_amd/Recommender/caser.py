"""Caser (Convolutional Sequence Embedding Recommendation) on the MI355X engine — constructor, hooks and semantics of
DRecPy/Recommender/caser.py; arithmetic in drx_caser.hip / drx_generic.hip behind include/drx.h.

Reference behaviour kept: ListSampler configured as caser.py:72-75 (window of L inputs + T targets + T*neg_ratio
negative ids, sorted by `sort_column`), vertical conv that sums over the embedding dimension (caser.py:53 — not the
paper's per-dimension conv), max over time of the horizontal convs, dense Keras Adam with one apply per registered layer
(6 + L per step), l2(reg_rate) on embeddings and kernels.  TF's dropout RNG cannot be reproduced: the keep mask comes
from a counter-based hash of (model seed, step, sample, unit) evaluated in the kernel (or is injected with `dropout_mask_fn` for tests); act_h / act_mlp: relu (the
reference's default for both), tanh, sigmoid, linear.
`_rank(novelty=False)` ignores the candidate list and ranks ALL items, exactly like caser.py:128-146 (only the novelty branch
filters by `iids` there; ranking_evaluation defaults to novelty=False, so HR/NDCG@k are comparable with the reference's);
`Caser(reference_rank=False)` restricts that branch to the candidates instead.
"""
import numpy as np

from .recommender_abc import RecommenderABC
from ..Sampler import ListSampler


class Caser(RecommenderABC):
    _host_prefetch = True      # fit() draws batch t+1 on a worker thread while batch t trains (sampler-only, engine-free hook)
    _prefetch_from = 2048      # ... for batches of at least this many samples (smaller ones: drawn inline, recommender_abc.fit)

    def __init__(self, L=5, T=3, d=50, n_v=4, n_h=16, act_h='relu', act_mlp='relu', dropout_rate=0.5,
                 sort_column='timestamp', device='cuda:0', reference_rank=True, **kwds):
        super().__init__(**kwds)
        self.reference_rank = reference_rank
        # the reference hands act_h / act_mlp to Keras (caser.py:57,63: any activation name or callable); the kernels implement
        # relu, tanh, sigmoid and linear, and lanes = embedding channels of one wavefront bound d; other values are rejected HERE
        supported = ('relu', 'tanh', 'sigmoid', 'linear', None)
        if act_h not in supported or act_mlp not in supported:
            raise Exception(f'drecpy_amd.Caser supports the activations {supported[:4]} (given: act_h={act_h!r}, act_mlp={act_mlp!r}).')
        # The HIP kernels (drx_caser_tile.hpp: tiles of 16 samples on the matrix cores, the tile's item rows and the convolution weights in
        # LDS; drx_caser.hip: inference) take 1 <= L <= 8 and 1 <= d <= 64 — BASELINE configuration 5 and examples/caser.py (L = 5, d = 50).
        # Anything else is REJECTED here, as DMF rejects towers it has no kernel for: there is no second backend (through r05 a
        # torch-autodiff engine took those shapes; it is now tests/caser_torch_checker.py, a checker).
        if not (1 <= L <= 8 and 1 <= d <= 64):
            raise Exception(f'drecpy_amd.Caser supports 1 <= L <= 8 and 1 <= d <= 64 (given: L={L}, d={d}).')
        # caser.py:108: tf.squeeze(tf.nn.max_pool1d(conv, n_h, n_h, 'SAME'), 1) is a max over ALL positions only while the longest
        # convolution output (L positions) fits one pooling window; with L > n_h the pooled axis has ceil(L / n_h) > 1 entries and the
        # reference's squeeze raises — a model the reference cannot produce is not trained here either
        if L > n_h:
            raise Exception(f'drecpy_amd.Caser needs L <= n_h (given: L={L}, n_h={n_h}): the reference pools windows of n_h positions and '
                            f'squeezes the pooled axis (caser.py:108), which fails for L > n_h.')
        if n_v < 1 or n_h < 1 or T < 1:
            raise Exception(f'drecpy_amd.Caser needs n_v, n_h, T >= 1 (given: n_v={n_v}, n_h={n_h}, T={T}).')
        self.act_h, self.act_mlp = act_h, act_mlp
        self.L, self.T, self.d, self.n_v, self.n_h = L, T, d, n_v, n_h
        self.dropout_rate = dropout_rate
        self.sort_column = sort_column
        self.device = device

    def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):          # caser.py:45-75
        from ..engine_caser import CaserEngine
        self.neg_ratio = neg_ratio
        # (raises DrxError when the configuration's tiles do not fit the LDS — very many filters; no other backend takes it)
        self._engine = CaserEngine(self.n_users, self.n_items, self.L, self.T, neg_ratio, self.d, self.n_v, self.n_h, device=self.device,
                                   act_h=self.act_h, act_mlp=self.act_mlp)
        self._engine.lr, self._engine.reg = float(learning_rate), float(reg_rate)
        weights = kwds.get('initial_weights')
        if weights is None:
            weights = self._keras_init(np.random.default_rng(self.seed))
        self._engine.set_params(weights)
        self._layers = self._engine.layers
        self._register_trainables(self._layers)                              # caser.py:47-70, one Keras layer each
        self._sampler = ListSampler(self.interaction_dataset, ['uid'], neg_ratio=neg_ratio, n_targets=self.T,
                                    interaction_threshold=self.interaction_threshold, negative_ids_col='iid',
                                    min_positive_records=self.L, max_positive_records=self.L,
                                    sort_column=self.sort_column, seed=self.seed)
        # fit(..., device_sampler=True): THROUGHPUT MODE, a named deviation like CDAE's — the windows are drawn on the device by a
        # counter-based generator with the reference sampler's distribution (include/drx.h drx_list_sample_device) instead of the
        # reference's one MT19937 stream, whose 0.5 us per window on one host thread is ten times the device step at B = 4096.
        # The default stays the reference-exact stream.
        self._device_sampler = bool(kwds.get('device_sampler', False))
        if self._device_sampler:
            self._sampler.device_twin(self.device)
            self._host_prefetch = False
            self._draws = 0
            self._ahead = None                            # (a batch drawn ahead by an earlier fit() belongs to its sampler)
            self._sampler_kind = 'device list sampler (drx_list_sample_device, counter-based; throughput mode)'
        else:
            self.__dict__.pop('_host_prefetch', None)
            self._sampler_kind = 'reference-exact ListSampler stream (C++)'
        self._drop_seed = int(self.seed) if self.seed is not None else int(np.random.SeedSequence().entropy % (2 ** 62))
        self._dropout_mask_fn = kwds.get('dropout_mask_fn')

    def _restore_engine(self, params):
        """RecommenderABC.load (recommender_abc.py:517-524): the engine rebuilt from the saved weights"""
        from ..engine_caser import CaserEngine
        self._engine = CaserEngine(self.n_users, self.n_items, self.L, self.T, self.neg_ratio, self.d, self.n_v, self.n_h, device=self.device,
                                   act_h=self.act_h, act_mlp=self.act_mlp)
        self._engine.set_params(params)
        self._layers = self._engine.layers

    def _fused_trainables(self):
        return self._engine.layers

    def _configure_optimizer(self):
        o = self.optimizer
        if getattr(o, 'kind', None) != 'adam':
            raise Exception(f'Caser trains with Keras Adam only (dense update of every registered layer); got {o!r}')
        e = self._engine
        e.lr, e.beta1, e.beta2, e.eps = o.learning_rate, o.beta_1, o.beta_2, o.epsilon

    def _keras_init(self, rng):
        """Keras defaults: Embedding uniform(-0.05, 0.05); Conv1D / Dense glorot_uniform kernels, zero biases."""
        f32 = np.float32
        u = lambda *s: rng.uniform(-0.05, 0.05, size=s).astype(f32)

        def glorot(shape, fan_in, fan_out):
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            return rng.uniform(-lim, lim, size=shape).astype(f32)
        L, d, n_v, n_h = self.L, self.d, self.n_v, self.n_h
        p = {'user_emb': u(self.n_users, d), 'item_emb': u(self.n_items, d),
             'conv_v_k': glorot((L, d, n_v), L * d, L * n_v), 'conv_v_b': np.zeros(n_v, f32)}
        for i in range(L):
            p[f'conv_h{i}_k'] = glorot((i + 1, d, n_h), (i + 1) * d, (i + 1) * n_h)
            p[f'conv_h{i}_b'] = np.zeros(n_h, f32)
        nx = n_v + L * n_h
        p['dense0_k'] = glorot((nx, d), nx, d)
        p['dense0_b'] = np.zeros(d, f32)
        p['W1'] = u(self.n_items, 2 * d)
        p['b1'] = u(self.n_items, 1)
        return p

    def _sample_batch(self, batch_size, **kwds):                          # caser.py:77-84
        if getattr(self, '_device_sampler', False):
            return self._draw_ahead(batch_size)
        if getattr(self._sampler, '_native', None) is not None:           # array form of the same draws (C++ loop)
            ds = self.interaction_dataset
            grp, in_off, in_rows, tg_off, tg_rows, ng_off, negs = self._sampler.sample_group_arrays(batch_size)
            iid = ds._cols['iid']
            L, T, Tn = self.L, self.T, self.T * self.neg_ratio
            assert in_off[-1] == batch_size * L and tg_off[-1] == batch_size * T and ng_off[-1] == batch_size * Tn
            before = np.asarray(iid[in_rows], dtype=np.int64).reshape(batch_size, L)
            after = np.concatenate([np.asarray(iid[tg_rows], dtype=np.int64).reshape(batch_size, T),
                                    negs.astype(np.int64).reshape(batch_size, Tn)], axis=1)
            # (the engine's host-side preparation of the batch — lookups grouped by table row — stays with _do_batch on the main
            # thread: the ListSampler stream on this worker thread is what bounds Caser.fit())
            return np.asarray(grp, dtype=np.int64), before, after
        uids, before, after = [], [], []
        for pos, targets, negs in self._sampler.sample_group_records(batch_size):
            uids.append(int(pos[0]['uid']))
            before.append([int(r['iid']) for r in pos])
            after.append([int(r['iid']) for r in targets] + [int(x) for x in negs])
        return uids, before, after

    def _draw_ahead(self, batch_size):
        """device_sampler=True: the windows of batch s + 1 are drawn — and their lookups grouped by table row (engine.prepare_device_batch)
        — on a side stream while step s trains: neither depends on the parameters.  The draws are the ones the inline loop made (seed =
        f(model seed, number of the draw)); one batch beyond the last step is drawn and dropped."""
        import torch
        eng = self._engine
        st = self.__dict__.get('_ahead')
        if st is None or st['B'] != batch_size:
            from ..engine import run_ahead_stream
            st = self._ahead = {'B': batch_size, 'side': run_ahead_stream(eng.device, 0), 'next': None}
        main = torch.cuda.current_stream(eng.device)

        from ..engine import _on_stream                 # (`with torch.cuda.stream(...)` without most of its Python layers)

        def draw():
            self._draws += 1
            with _on_stream(st['side']):
                # a ring slot of the engine: the id tensors, the grouped lookups and every argument struct built once per slot
                sl = eng.device_slot(batch_size)
                self._sampler.sample_device(batch_size, self._drop_seed * 1000003 + self._draws, out=(sl['uid'], sl['before'], sl['after']))
                prep = eng.group_slot(sl)
                ev = torch.cuda.Event()
                ev.record(st['side'])
            return prep, ev
        if st['next'] is None:
            st['side'].wait_stream(main)                  # (tables / sampler arrays set up on the training stream)
            st['next'] = draw()
        prep, ev = st['next']
        main.wait_event(ev)
        st['next'] = draw()
        return prep['uid'], prep['before'], prep['after'], prep

    def _do_batch(self, batch_samples, step=0, want_loss=False, **kwds):
        uids, before, after = batch_samples[:3]
        prep = batch_samples[3] if len(batch_samples) > 3 else None
        B = len(uids)
        keep, rate = None, 0.0
        if self.dropout_rate and self.dropout_rate > 0:
            rate = float(self.dropout_rate)
            nx = self.n_v + self.L * self.n_h
            # an injected mask (tests), else the kernel's counter-based mask keyed by (model seed, step): TF's dropout stream
            # cannot be reproduced, and 344 K host random numbers per batch of 4096 cost more than the device step
            keep = self._dropout_mask_fn(step, B, nx) if self._dropout_mask_fn is not None else None
        if prep is not None:
            return self._engine.step(step, prep, keep=keep, rate=rate, want_loss=want_loss,
                                     mask_seed=self._drop_seed * 0x9E3779B97F4A7C15 + step + 1)
        if hasattr(uids, 'data_ptr'):                 # a batch drawn on the device (device_sampler=True)
            return self._engine.step(step, uids, before, after, keep, rate, want_loss=want_loss,
                                     mask_seed=self._drop_seed * 0x9E3779B97F4A7C15 + step + 1)
        return self._engine.step(step, np.asarray(uids), np.asarray(before), np.asarray(after), keep, rate,
                                 want_loss=want_loss, mask_seed=self._drop_seed * 0x9E3779B97F4A7C15 + step + 1)

    def _predict_batch(self, batch_samples, **kwds):
        """Sigmoid scores of the batch's targets (caser.py:86-95, evaluation mode: no dropout)."""
        import torch
        uids, before, after = batch_samples[:3]
        with self._device_lock:
            sc = self._engine.scores_all(np.asarray(uids), np.asarray(before))
        # (an API-compatibility hook, never on the training path: the gather + sigmoid of caser.py:94 on the host)
        logits = np.take_along_axis(sc.cpu().numpy().astype(np.float64), np.asarray(after, dtype=np.int64), axis=1)
        preds = torch.as_tensor(1.0 / (1.0 + np.exp(-logits)), dtype=torch.float32)
        desired = np.tile(np.array([1.] * self.T + [0.] * (self.T * self.neg_ratio), dtype=np.float32), (len(uids), 1))
        return preds, desired

    def _compute_batch_loss(self, predictions, desired_values, **kwds):
        p = np.asarray(predictions.cpu().numpy() if hasattr(predictions, 'cpu') else predictions, dtype=np.float64)
        t = np.asarray(desired_values, dtype=np.float64)
        eps = 1e-7
        pc = np.clip(p, eps, 1 - eps)
        return float((-(t * np.log(pc + eps) + (1 - t) * np.log(1 - pc + eps))).mean())

    def _predict(self, uid, iid, **kwds):
        raise NotImplementedError('This model does not support point-based predictions.')

    def _user_sequence(self, uid):
        if not hasattr(self, '_seq_ptr'):
            ds = self.interaction_dataset
            u = ds._cols['uid'].astype(np.int64)
            key = ds._cols[self.sort_column] if self.sort_column in ds.columns else np.arange(len(u))
            order = np.lexsort((np.arange(len(u)), key, u))           # by user, then sort column (stable)
            self._seq_items = ds._cols['iid'][order].astype(np.int64)
            self._seq_ptr = np.searchsorted(u[order], np.arange(self.n_users + 1))
        return self._seq_items[self._seq_ptr[uid]:self._seq_ptr[uid + 1]]

    def _rank(self, uid, iids, n, novelty):                               # caser.py:128-146
        import torch
        from ..engine import CdaeEngine, pack_mask_bits
        seq = self._user_sequence(uid)
        cand = np.zeros(self.n_items, dtype=bool)
        cand[np.fromiter((int(i) for i in iids), dtype=np.int64)] = True
        if novelty:
            cand[seq] = False
        elif getattr(self, 'reference_rank', True):
            cand[:] = True                                                # caser.py:146: nlargest over every item
        k = min(int(n), int(cand.sum()))
        if k <= 0:
            return []
        with self._device_lock:
            sc = self._engine.scores_all(np.array([uid]), seq[-self.L:][None, :])
            if not hasattr(self, '_topk_helper'):
                self._topk_helper = CdaeEngine.__new__(CdaeEngine)
                self._topk_helper.device = self._engine.device
            mask = torch.as_tensor(pack_mask_bits(cand).view(np.int32)).to(sc.device)
            idx, val = CdaeEngine.topk(self._topk_helper, sc, k, mask)
            idx, val = idx[0].cpu().numpy(), val[0].cpu().numpy()
        return [(float(v), int(i)) for v, i in zip(val, idx) if i >= 0]
