"""CPU legs of bench.py and bench_configs.py (`cpu_baseline*`): the oracle timed on the GPU box's host cores.  The ONLY place outside tests/ and
__graft_entry__.smoke() that touches oracle/ — as the thing timed beside the product, never as part of it."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))


def _compact_problem(eng, hist_indptr, hist_indices, batch, seed, n_cpu, q, q_threshold):
    """The first n_cpu triples of one bench batch with the tables compacted to the rows they touch (same arithmetic per sample; the
    CPU sees a cache-friendlier table than the GPU does)."""
    from oracle import cdae_oracle as co
    n_cpu = min(n_cpu, batch[0].numel())
    uid, iid, y = [t[:n_cpu].cpu().numpy() for t in batch[:3]]
    ip = hist_indptr.cpu().numpy() if hist_indptr.numel() < 50_000_000 else None
    thr = q_threshold(q)
    kept, users, items = [], {}, {}
    for b in range(n_cpu):
        u = int(uid[b])
        s, e = (int(hist_indptr[u].item()), int(hist_indptr[u + 1].item())) if ip is None else (int(ip[u]), int(ip[u + 1]))
        row = hist_indices[s:e].cpu().numpy()
        kf = co.drx_hash_u32(seed, np.full(e - s, b), np.arange(e - s)) >= thr
        users.setdefault(u, len(users))
        for n in row[kf].tolist() + [int(iid[b])]:
            items.setdefault(n, len(items))
        kept.append([items[n] for n in row[kf].tolist()])
    ul = torch.tensor(list(users.keys()), device=eng.device)
    il = torch.tensor(list(items.keys()), device=eng.device)
    k = eng.k
    p = {'W': eng.W[il, :k].cpu().numpy().copy(), 'W_': eng.W2T[il, :k].t().cpu().numpy().copy(),
         'V': eng.V[ul, :k].cpu().numpy().copy(), 'b': eng.b[:k].cpu().numpy().copy(), 'b_': eng.b2[il].cpu().numpy().copy()}
    cu = np.array([users[int(u)] for u in uid])
    ci = np.array([items[int(i)] for i in iid])
    return n_cpu, p, cu, ci, y, kept


def cpu_baseline(eng, hist_indptr, hist_indices, batch, seed, q, lr, reg, q_threshold, budget_s=12.0, n_cpu=1024, optimizer='adagrad',
                 all_cores=True, all_cores_budget_s=8.0):
    """(one-core dict, all-cores dict or None).  One core: oracle/cdae_oracle.py:sparse_step (the NumPy restatement, 'port') on the first
    n_cpu triples of one bench batch, in this process.  All cores: the same loop in one worker process per host core
    (oracle/cpu_baseline_worker.py, each on a private copy of the compacted tables), rates added up."""
    from oracle import cdae_oracle as co
    n_cpu, p, cu, ci, y, kept = _compact_problem(eng, hist_indptr, hist_indices, batch, seed, n_cpu, q, q_threshold)
    lr_ = 1e-3 if optimizer == 'adam' else lr
    many = None
    if all_cores:
        many = _all_cores(p, cu, ci, y, kept, q, lr_, reg, optimizer, n_cpu, all_cores_budget_s)
    st = co.sparse_state(p, optimizer)
    t0 = time.perf_counter()
    n_done = 0
    while time.perf_counter() - t0 < budget_s:
        co.sparse_step(p, st, n_done, cu, ci, y, kept, float(np.float32(q)), lr_, reg, 'bce', optimizer)
        n_done += 1
    dt = time.perf_counter() - t0
    one = {'value': n_cpu * n_done / dt, 'unit': 'samples/s', 'cores': 1, 'kind': 'port',
           'sample': f'{n_done} steps of the first {n_cpu} triples of one bench batch (same tables, compacted to touched '
                     f'rows), NumPy restatement oracle/cdae_oracle.py:sparse_step, {dt:.1f} s, host has {os.cpu_count()} cpus'}
    return one, many


def _all_cores(p, cu, ci, y, kept, q, lr, reg, optimizer, n_cpu, budget_s):
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    off = np.zeros(len(kept) + 1, np.int64)
    off[1:] = np.cumsum([len(k_) for k_ in kept])
    flat = np.array([x for k_ in kept for x in k_], dtype=np.int64)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'problem.npz')
        np.savez(path, **{'p_' + k_: v for k_, v in p.items()}, cu=cu, ci=ci, y=y, kept_off=off, kept_flat=flat,
                 q=np.float64(np.float32(q)), lr=np.float64(lr), reg=np.float64(reg), optimizer=np.array(optimizer))
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
        worker = os.path.join(ROOT, 'oracle', 'cpu_baseline_worker.py')
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, worker, path, str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
                 for _ in range(cores)]
        rate, ok = 0.0, 0
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=budget_s * 6 + 120)
                n, dt = out.decode().split()[:2]
                rate += n_cpu * int(n) / float(dt)
                ok += 1
            except Exception:
                pr.kill()
        wall = time.perf_counter() - t0
    if not ok:
        return None
    return {'value': rate, 'unit': 'samples/s', 'cores': ok, 'kind': 'port',
            'sample': f'{ok} single-threaded worker processes (one per host core available to this process), each looping '
                      f'oracle/cdae_oracle.py:sparse_step over the same {n_cpu} triples on a private copy of the compacted tables for {budget_s:.0f} s; '
                      f'rates added up ({wall:.1f} s wall incl. start-up)'}


def cpu_reference_fit(tr, budget_s=10.0, B=64, k=50, q=0.2, seed=10):
    """BASELINE config 1 on the host: the reference's CDAE fit() loop (examples/cdae.py: K = 50, batch 64, lr 1e-3, reg 1e-3, neg_ratio 5)
    as restated by the oracle — PointSampler draw (oracle/data_oracle.py), N uniform corruption draws per row from random.Random(seed)
    (cdae.py:63), dense_step in fp32 (what TF computes in) with the (B,B,N) mean-target loss, L2/B on the full tables and 5 Keras-Adam
    applies — on the SAME training set the GPU fit() above trains on, for `budget_s` seconds, numpy limited to one thread."""
    import random
    from oracle import cdae_oracle as co
    from oracle import data_oracle as do
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    c = tr._cols
    uid, iid, val = c['uid'].astype(np.int64), c['iid'].astype(np.int64), c['interaction']
    U, N = int(uid.max()) + 1, int(iid.max()) + 1
    smp = do.PointSamplerOracle(uid, iid, val, 5, 1e-3, seed)
    rng = random.Random(seed)
    pos = np.zeros((U, N), dtype=bool)
    pos[uid[val >= 1e-3], iid[val >= 1e-3]] = True
    p = co.init_params(np.random.default_rng(seed), U, N, k, np.float32)
    st = co.adam_state(p)
    scale = np.float32(1.0 / (1.0 - float(np.float32(q))))

    def loop():
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < budget_s:
            users = np.array([t[0] for t in smp.sample(B)])
            t = pos[users]
            mask = np.array([[rng.uniform(0, 1) >= q for _ in range(N)] for _ in range(B)])      # cdae.py:63: a draw for EVERY item
            x = np.where(t & mask, scale, np.float32(0)).astype(np.float32)
            co.dense_step(p, st, n, users, x, t, 1e-3, 1e-3, 'bce', 'reference')
            n += 1
        return n, time.perf_counter() - t0
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            n, dt = loop()
    else:
        n, dt = loop()
    return {'value': n * B / dt, 'unit': 'samples/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n} one-batch epochs of {B} in {dt:.1f} s: oracle PointSampler + per-item corruption draws + oracle/cdae_oracle.py:dense_step '
                      f'(fp32) on the {U} x {N} ml-100k-shaped training set of the GPU fit; numpy on 1 thread; host has {os.cpu_count()} cpus'}


def dmf_cpu_baseline(ds, B=256, budget_s=6.0):
    """oracle/dmf_oracle.step (the reference's dense-row formulation: X_u [B, N], X_i [B, U] densified per batch, dmf.py:76-77) on the
    host, one numpy thread, for `budget_s` seconds."""
    from oracle import dmf_oracle as dm
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    import scipy.sparse as sp
    ip, cols, vals = ds.interaction_csr()
    U, N = len(ip) - 1, int(np.max(cols)) + 1
    R = sp.csr_matrix((np.asarray(vals, np.float32), np.asarray(cols), np.asarray(ip)), shape=(U, N))
    Rt = R.T.tocsr()
    rng = np.random.default_rng(0)
    p = dm.init_params(rng, U, N)
    st = dm.adam_state(p)

    def loop():
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s:
            u, i = rng.integers(0, U, size=B), rng.integers(0, N, size=B)
            xu, xi = np.asarray(R[u].todense(), np.float32), np.asarray(Rt[i].todense(), np.float32)
            y = np.asarray(R[u, i]).ravel().astype(np.float32) / 5.0
            dm.step(p, st, n, xu, xi, y, 1e-3, 1e-4, 2, 2)
            n += 1
        return n, time.perf_counter() - t0
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            n, dt = loop()
    else:
        n, dt = loop()
    return {'value': n * B / dt, 'unit': 'samples/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n} steps of {B} uniform (user, item) pairs in {dt:.1f} s: rows densified per batch + oracle/dmf_oracle.py:step (fp32) on the '
                      f'{U} x {N} ml-1m-shaped set; numpy on 1 thread; host has {os.cpu_count()} cpus'}


def caser_cpu_baseline(n_users, n_items, B=512, budget_s=6.0):
    """oracle/caser_oracle.step (caser.py:97-120 under the tape; Keras Adam per layer) on the host, one numpy thread."""
    from oracle import caser_oracle as ca
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    rng = np.random.default_rng(0)
    L, T, d, n_v, n_h, neg = 5, 3, 50, 4, 16, 3
    p = ca.init_params(rng, n_users, n_items, L, d, n_v, n_h, np.float32)
    st = ca.adam_state(p)
    nx = n_v + L * n_h

    def loop():
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s:
            uids = rng.integers(0, n_users, size=B)
            before, after = rng.integers(0, n_items, size=(B, L)), rng.integers(0, n_items, size=(B, T + T * neg))
            ca.step(p, st, n, uids, before, after, T, 5e-3, 1e-6, rng.random((B, nx)) >= 0.5, 0.5)
            n += 1
        return n, time.perf_counter() - t0
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            n, dt = loop()
    else:
        n, dt = loop()
    return {'value': n * B / dt, 'unit': 'windows/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n} steps of {B} random windows in {dt:.1f} s: oracle/caser_oracle.py:step (fp32, dense Keras Adam over the '
                      f'{n_users} x {n_items} tables); numpy on 1 thread; host has {os.cpu_count()} cpus'}
