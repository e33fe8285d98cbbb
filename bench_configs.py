"""The `configs` block of bench.py's JSON line: BASELINE.json configurations 2, 3 and 5 measured in the driver's own run (VERDICT r02
item 3), beside the headline (configuration 4 at one GPU) and `hr_at_10` (configuration 1).  Budget: about a minute.

  cfg2  CDAE hidden_factors=128 at the ml-1m shape, sampled-output sparse-Adagrad step (the same kernels as the headline, MovieLens
        geometry: 6040 x 3706, 165-item histories) — triples/s, per-kernel dedup-aware roofline fractions, its own cpu_baseline
  cfg3  DMF and ModifiedDMF (examples/extending_recommender_dmf.py:5-18) at the ml-1m shape, B = 256 and 4096: device step and the
        public fit() rate; the MFMA bf16 all-pairs scorer (2048 users x 3706 items): time and achieved write GB/s
  cfg5  Caser (examples/caser.py:13-14) at the ml-1m shape, B = 4096: device step and fit() rate
No MovieLens files exist offline: the sets are the seeded synthetic stand-ins of drecpy_amd.synth (same shapes)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench_cpu          # noqa: E402  (the CPU legs: the only importer of oracle/ outside tests and smoke)


def frame_of(shape, seed=0):
    from drecpy_amd import synth
    U, N, md, mn, a = synth.SHAPES[shape]
    ip, idx = synth.synth_history(U, N, md, mn, a, seed=seed)
    ip, idx = ip.numpy(), idx.numpy()
    rng = np.random.RandomState(seed)
    user = np.repeat(np.arange(U), np.diff(ip)) + 1
    item = idx.astype(np.int64) + 1
    perm = rng.permutation(len(user))                                   # shuffled row order, like a ratings file
    return {'user': user[perm], 'item': item[perm], 'interaction': rng.randint(1, 6, size=len(user))[perm],
            'timestamp': rng.randint(0, 10 ** 9, size=len(user))[perm]}


def _timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def _events_ms(fn, n):
    """Mean device time of fn() over n calls, HIP events on the current stream."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def _fit_steady(model, fit, every, n_windows=6):
    """(seconds per step of the whole fit() incl. set-up, seconds per step of the STEADY STATE, the windows' spread): ONE long fit() of
    (n_windows + 1) * every one-batch epochs; behind every `every`-th step the device is synchronised and the host clock stamped, and
    the steady state is the MEDIAN of the n_windows fenced windows behind the first stamp (the first `every` epochs carry the set-up
    and are not a window).  r01 - r04 took the difference of a long and a short fit: at 50 - 170 us per step the two runs' set-up
    jitter was larger than what it measured (a "steady" rate below the device step).
    The stamps ride on the model's own `_do_batch` (wrapped for the duration of the fit): fit()'s `epoch_callback_fn` only runs for
    verbose fits and fits with an early-stopping rule (recommender_abc.py:207-214), both of which read the loss back every epoch."""
    stamps, count = [], [0]
    inner = model._do_batch

    def stamped(batch_samples, **kw):
        r = inner(batch_samples, **kw)
        count[0] += 1
        if count[0] % every == 0:
            torch.cuda.synchronize()
            stamps.append(time.perf_counter())
        return r
    n = (n_windows + 1) * every
    model._do_batch = stamped
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit(n)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    finally:
        del model._do_batch                          # (the instance attribute: the class's method is back)
    w = np.diff(np.asarray(stamps[:n_windows + 1])) / every
    if len(w) == 0:                                   # (a model that ran its loop natively: the whole run, set-up included)
        return total / n, total / n, {'windows': 0}
    return total / n, float(np.median(w)), {'windows': int(len(w)), 'every': every, 'window_ms_per_step_min': float(w.min() * 1e3),
                                           'window_ms_per_step_max': float(w.max() * 1e3)}


HBM_PEAK_GBS, L2_GATHER_GBS = 8000.0, 17000.0       # MI355X_MICROARCH.md: HBM3E peak; indexed rows served by the XCDs' L2 (16.8 - 18.8 TB/s)


def _roofline(step_ms, hbm_bytes, requested_bytes, what):
    """The step against its two bounds: `frac` = necessary HBM bytes / step time / 8 TB/s (the contract's field; tiny
    where the tables are L2-resident) and `cache_level` = the rows the kernels request from L2 / step time / the guide's L2 gather rate."""
    t = step_ms * 1e-3
    return {'bound': 'hbm', 'kernel': 'whole step (all launches)', 'achieved': hbm_bytes / t / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': hbm_bytes / t / 1e9 / HBM_PEAK_GBS, 'traffic': None, 'avg_launch_ms': step_ms, 'bytes_per_launch': hbm_bytes,
            'cache_level': {'requested_bytes_per_step': requested_bytes, 'requested_GBs': requested_bytes / t / 1e9, 'bound_GBs': L2_GATHER_GBS,
                            'frac_of_bound': requested_bytes / t / 1e9 / L2_GATHER_GBS,
                            'time_at_bound_ms': 1e3 * (requested_bytes / (L2_GATHER_GBS * 1e9) + hbm_bytes / (HBM_PEAK_GBS * 1e9))},
            'definition': what}


def _dmf_bytes(ds, u, i, f0u, f0i, n_rest_params):
    """Algorithmic bytes of one DMF step (dmf.py:76-96 under recommender_abc.py:190-204): the first layers are embedding bags — sample b
    sums r_un * K0u[n] over the items of its user's row and r_ui * K0i[u] over the users of its item's column (rows of f0 floats), and the
    backward pass visits the same (row, sample) pairs; Keras Adam is DENSE: every parameter's p, m, v read and written every step."""
    ip, cols, _ = ds.interaction_csr()
    deg_u = np.diff(np.asarray(ip))
    deg_i = np.bincount(np.asarray(cols), minlength=int(np.max(cols)) + 1)
    row_uses = float(deg_u[np.asarray(u)].sum()) * f0u + float(deg_i[np.asarray(i)].sum()) * f0i          # floats
    n_k0 = float(len(deg_i)) * f0u + float(len(deg_u)) * f0i
    nnz = float(len(cols))
    requested = 4.0 * 2.0 * row_uses                                   # forward bags + the same pairs in the backward pass
    hbm = 24.0 * (n_k0 + n_rest_params) + 2.0 * 8.0 * nnz + 12.0 * len(u)       # dense Adam sweep (p, m, v in and out), one walk over both
    return hbm, requested, {'bag_row_uses_floats': row_uses, 'first_layer_params': n_k0, 'nnz': nnz}            # orientations' (index, value), ids


def dmf_block(ds, dev, with_cpu=True):
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    from drecpy_amd.Recommender import DMF
    out = {}
    classes = [('DMF', DMF)]
    try:
        from extending_recommender_dmf import ModifiedDMF
        classes.append(('ModifiedDMF', ModifiedDMF))
    except Exception as e:                                      # noqa: BLE001
        out['ModifiedDMF'] = {'error': repr(e)}
    m = None
    for name, cls in classes:
        for B in (256, 4096):
            m = cls(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=False, device=str(dev))
            m.fit(ds, epochs=2, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5)
            batch = m._sample_batch(B)
            if len(batch) == 3:                                     # (large batches are prepared by _do_batch itself: here once, up front)
                batch = tuple(batch) + (m._engine.prepare_batch(*batch[:3]),)
            state = {'s': 2}

            def step():
                m._do_batch(batch, step=state['s'])
                state['s'] += 1
            dev_s = _timed(step, 40)
            ev_ms = _events_ms(step, 40)
            if name == 'DMF':
                e_ = m._engine
                f0u, f0i = int(m.user_factors[0]), int(m.item_factors[0])
                rest = sum(a_ * b_ + b_ for fs in (m.user_factors, m.item_factors) for a_, b_ in zip(fs[:-1], fs[1:])) + f0u + f0i
                hbm, req, counts = _dmf_bytes(ds, batch[0], batch[1], f0u, f0i, rest)
                rl = _roofline(ev_ms, hbm, req, 'DMF step, HIP events over all its launches; HBM bytes = dense Keras-Adam sweep of every parameter '
                               '(24 B per float) + one walk over both orientations of the interaction matrix (8 B per non-zero) + the batch ids; '
                               'requested = the first layers\' embedding-bag rows, forward and backward (4 * f0 B per (row, sample) pair) — the two '
                               'first-layer kernels are 0.9 / 1.5 MB: L2-resident, so the cache-level bound is the binding one')
                rl['counts'] = counts
                out.setdefault('roofline', {})[f'B{B}'] = rl
            # the public call: set-up included, and the steady state between two lengths
            every = 400 if B <= 256 else 150
            e2e, steady, spread = _fit_steady(m, lambda n: m.fit(ds, epochs=n, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5), every)
            out[f'{name}_B{B}'] = {'step_ms': dev_s * 1e3, 'step_samples_per_s': B / dev_s, 'fit_ms_per_step_incl_setup': e2e * 1e3,
                                   'fit_steady_ms_per_step': steady * 1e3, 'fit_samples_per_s': B / steady, 'fit_windows': spread}
            # throughput mode: triples drawn and prepared on the device, one step ahead (a named deviation, like CDAE's and Caser's)
            md = cls(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=False, device=str(dev))
            md.fit(ds, epochs=3, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5, device_sampler=True)
            e2d, steady_d, spread_d = _fit_steady(md, lambda n: md.fit(ds, epochs=n, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5,
                                                                       device_sampler=True), every)
            out[f'{name}_B{B}_device_sampler'] = {'fit_ms_per_step_incl_setup': e2d * 1e3, 'fit_steady_ms_per_step': steady_d * 1e3,
                                                  'fit_samples_per_s': B / steady_d, 'fit_windows': spread_d,
                                                  'sampler': getattr(md, '_sampler_kind', None)}
    # the one MFMA kernel: all-pairs cosine scores of a block of users against every item (k_score_pairs_bf16)
    from drecpy_amd import _lib
    L = _lib.lib()
    n_u, n_i = 2048, m.n_items
    ru = torch.nn.functional.normalize(torch.randn(n_u, 64, device=dev), dim=1)
    ri = torch.nn.functional.normalize(torch.randn(n_i, 64, device=dev), dim=1)
    pitch = (n_i + 31) // 32 * 32
    sc = torch.empty(n_u, pitch, device=dev)

    def score():
        _lib.check(L.drx_score_pairs_bf16(_lib.ptr(ru), n_u, _lib.ptr(ri), n_i, 64, 32, None, _lib.ptr(sc), pitch, _lib.stream_ptr(dev)), 'score')
    ms = _events_ms(score, 200)
    wbytes = n_u * pitch * 4.0
    flops = 2.0 * n_u * n_i * 32
    out['mfma_scorer'] = {'kernel': 'k_score_pairs_bf16 (v_mfma_f32_32x32x16_bf16)', 'users': n_u, 'items': n_i, 'factors': 32, 'ms': ms,
                          'write_bytes': wbytes, 'achieved_write_GBs': wbytes / (ms * 1e-3) / 1e9, 'frac_of_hbm_peak': wbytes / (ms * 1e-3) / 1e9 / 8000.0,
                          'tflops': flops / (ms * 1e-3) / 1e12, 'bound': 'hbm (16 FLOP per output byte)'}
    ue = torch.arange(0, n_u, device=dev)
    out['score_matrix_incl_towers_ms'] = _timed(lambda: m._engine.score_matrix_bf16(ue), 10) * 1e3
    out['cpu_baseline'] = bench_cpu.dmf_cpu_baseline(ds) if with_cpu else None
    return out


def caser_block(ds, dev, with_cpu=True):
    from drecpy_amd.Recommender import Caser
    out = {}
    for B in (4096,):
        m = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, seed=10, verbose=False, device=str(dev))
        m.fit(ds, epochs=2, batch_size=B, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3)
        batch = m._sample_batch(B)
        state = {'s': 2}

        def step():
            m._do_batch(batch, step=state['s'])
            state['s'] += 1
        dev_s = _timed(step, 30)
        ev_ms = _events_ms(step, 30)
        # algorithmic bytes of one Caser step (caser.py:97-120): per window L item rows + 1 user row of d floats and T' rows of dense_1
        # (2 d floats + bias) gathered; Keras Adam is DENSE over the three lookup tables (item and user embeddings, dense_1_W / _b) and
        # the small weights: p, m, v read and written every step; the gradient rows of the lookups: one per lookup, written and read once
        L_, T_, d_, nv_, nh_, neg_ = 5, 3, 50, 4, 16, 3
        Tp = T_ + T_ * neg_
        n_tab = (m.n_items + m.n_users) * d_ + m.n_items * (2 * d_ + 1)
        n_small = L_ * d_ * nv_ + nv_ + sum((i_ + 1) * d_ * nh_ + nh_ for i_ in range(L_)) + (nv_ + L_ * nh_) * d_ + d_
        gathered = 4.0 * B * ((L_ + 1) * d_ + Tp * (2 * d_ + 1))
        hbm = 24.0 * (n_tab + n_small) + 2.0 * gathered + 4.0 * B * (1 + L_ + Tp)
        out['roofline'] = {f'B{B}': _roofline(ev_ms, hbm, gathered, 'Caser step, HIP events over its three launches; HBM bytes = dense Keras-Adam sweep of '
                                              'the three lookup tables and the small weights (24 B per float) + the lookups\' gradient rows written and '
                                              'read once + the window ids; requested = the gathered lookup rows (tables of 0.7 - 1.5 MB: L2-resident)')}
        def fit_times(model, every, **kw):
            """seconds per step of one long fit() (set-up included) and of its steady state (median of fenced windows: _fit_steady)"""
            return _fit_steady(model, lambda n: model.fit(ds, epochs=n, batch_size=B, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3, **kw), every)
        e2e, steady, spread = fit_times(m, 60)
        out[f'Caser_B{B}'] = {'step_ms': dev_s * 1e3, 'step_windows_per_s': B / dev_s, 'fit_ms_per_step_incl_setup': e2e * 1e3,
                              'fit_steady_ms_per_step': steady * 1e3, 'fit_windows_per_s': B / steady, 'fit_windows': spread,
                              'sampler': getattr(m, '_sampler_kind', 'reference-exact ListSampler stream (C++)')}
        # throughput mode: windows drawn on the device (a named deviation, like CDAE's device PointSampler)
        m2 = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, seed=10, verbose=False, device=str(dev))
        m2.fit(ds, epochs=3, batch_size=B, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3, device_sampler=True)
        e2d, steady_d, spread_d = fit_times(m2, 120, device_sampler=True)
        out[f'Caser_B{B}_device_sampler'] = {'fit_ms_per_step_incl_setup': e2d * 1e3, 'fit_steady_ms_per_step': steady_d * 1e3,
                                             'fit_windows_per_s': B / steady_d, 'fit_windows': spread_d,
                                             'sampler': getattr(m2, '_sampler_kind', None)}
    out['cpu_baseline'] = bench_cpu.caser_cpu_baseline(m.n_users, m.n_items) if with_cpu else None
    return out


def configs_block(dev, run_direct, base_args, with_cpu=True):
    """run_direct: bench.py's single-GPU sampled-step measurement (called again at the ml-1m shape)."""
    import copy
    out = {}
    t_all = time.perf_counter()
    # ---- cfg2 -----------------------------------------------------------------------------------------------------------------
    try:
        a = copy.copy(base_args)
        a.workload, a.steps, a.warmup, a.windows, a.users = 'ml-1m', 30, 5, 3, 0
        a.no_hr, a.no_configs = True, True
        a.cpu_budget_s, a.cpu_triples, a.no_all_cores = 5.0, 128, True
        a.no_cpu_baseline = not with_cpu
        r = run_direct(a, 0, 1, dev, None)
        out['cfg2_cdae_ml1m_sampled'] = {
            'value': r['value'], 'unit': 'samples/s', 'ms_per_step': r['ms_per_step'], 'batch': r['config']['batch_per_gpu'],
            'workload': r['config']['workload'], 'phases_ms': r['phases_ms'],
            'roofline': {k_: r['roofline'].get(k_) for k_ in ('kernel', 'frac', 'achieved', 'whole_step_frac', 'whole_step_per_occurrence_frac', 'kernels', 'cache_bytes_k_seg_reduce',
                                                                        'cache_bytes_k_sampled_fwd_bwd', 'cache_resident', 'cache_level', 'traffic', 'traffic_source', 'row_counts',
                                                                        'whole_step_traffic', 'whole_step_traffic_frac')}
            if r.get('roofline') else None,
            'cpu_baseline': r.get('cpu_baseline')}
    except Exception as e:                                      # noqa: BLE001
        out['cfg2_cdae_ml1m_sampled'] = {'error': repr(e)}
    # ---- cfg3, cfg5 -----------------------------------------------------------------------------------------------------------
    ds = None
    try:
        from drecpy_amd.Dataset import InteractionDataset
        ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
    except Exception as e:                                      # noqa: BLE001
        out['dataset_error'] = repr(e)
    if ds is not None:
        for name, fn in (('cfg3_dmf_ml1m', dmf_block), ('cfg5_caser_ml1m', caser_block)):
            try:
                out[name] = fn(ds, dev, with_cpu)
            except Exception as e:                              # noqa: BLE001
                out[name] = {'error': repr(e)}
    out['seconds'] = round(time.perf_counter() - t_all, 1)
    out['data'] = 'synthetic ml-1m-shaped set (6040 users x 3706 items, ~1 M ratings), drecpy_amd.synth seed 0'
    return out
