"""bench.py — training samples/s of the CDAE hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload synth-10m|ml-1m|ml-100k] [--batch B]

A "step" is one pass of the hot path over one batch of B (u, i, y) triples per GPU: gather + hidden layer + sampled
output unit + BCE + backward + sparse-Adagrad update (drx_cdae_step_sparse_prepared).  At 1 GPU every step trains on a
FRESH batch drawn by the device PointSampler (drx_point_sample) five steps ahead on a side stream, and the batch's sorted
touch list (drx_cdae_sparse_prepare) is built three steps ahead on the same side stream — both depend only on the data,
never on the parameters — so the timed region is the whole training loop including sampling (`--presampled` cycles
through batches sampled at setup instead).  All inputs live in HBM; nothing crosses PCIe in the timed region except the
8-byte touch count the sampler posts to a pinned mailbox per step.  One process per GPU; for N > 1 launch with
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RCCL over xGMI).  The N-GPU layout is chosen with
`--layout`: `columns` (default: every rank holds all rows x K/N columns of every table and trains on the same global batch of
N x B triples — one all-reduce of B floats per step) or `rows` (the partitioning BASELINE.json describes: users and item rows
sharded by range, rows and gradient rows travel by all-to-all(v)); `config.sharding` in the output says which one ran.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the definitions of roofline / cpu_baseline / hr_at_10).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
K = 128
Q = 0.2
NEG_RATIO = 5
LR, REG = 0.05, 1e-3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='synth-10m', choices=['synth-10m', 'ml-1m', 'ml-100k'])
    ap.add_argument('--batch', type=int, default=65536, help='triples per GPU per step')
    ap.add_argument('--n-batches', type=int, default=8, help='batches sampled at setup (R estimate, CPU baseline; cycled with --presampled)')
    ap.add_argument('--presampled', action='store_true', help='cycle through the setup batches instead of sampling a fresh batch every step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-hr', action='store_true', help='skip the HR@10 sanity run (ml-100k-shaped set, reference mode)')
    ap.add_argument('--no-overlap', action='store_true', help='build the touch list inline instead of one batch ahead on a side stream')
    ap.add_argument('--users', type=int, default=0, help='override the number of users (debug)')
    ap.add_argument('--force-sharded', action='store_true', help='run the row-sharded step even at 1 GPU (measures its overhead)')
    ap.add_argument('--optimizer', default='adagrad', choices=['adagrad', 'adam', 'rowwise_adagrad'], help='sparse optimizer of the sampled mode (S = 1 / 2 slots per parameter; single-GPU path)')
    ap.add_argument('--k', type=int, default=0, help='hidden factors (default 128); with --force-columns --k 128/N --batch 65536*N one GPU runs the '
                                                      'shape of ONE rank of an N-GPU column-sharded job')
    ap.add_argument('--layout', default='columns', choices=['columns', 'rows'],
                    help='multi-GPU layout: columns = every rank all rows x K/N columns, same global batch, one all-reduce of B scalars per step; '
                         'rows = users and item rows sharded by range, rows and gradient rows travel by all-to-all')
    ap.add_argument('--force-columns', action='store_true', help='run the column-sharded code path even at 1 GPU')
    ap.add_argument('--prepare', default='auto', choices=['auto', 'local', 'turns', 'parts'],
                    help='column layout, who sorts the touch list of a step: every rank all of it (local), rank s %% N for all (turns), every rank '
                         '1/N of it (parts); auto = turns from 4 GPUs on (at 2 it saves nothing), else local')
    ap.add_argument('--micro', type=int, default=1, help='micro-batches per sharded step (exchanges of one overlap the compute of the other); default 1')
    return ap.parse_args()


def hbm_copy_gbs(dev, gib=2, reps=8):
    """Achievable HBM rate on this box: a device-to-device copy of `gib` GiB (read + write counted), GB/s."""
    n = gib * (1 << 30) // 4
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        dst.copy_(src)
    ev[1].record()
    torch.cuda.synchronize()
    return 2.0 * n * 4 * reps / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9


def hash_u32_torch(seed, a, b):
    """drx_hash_u32 restated with wrapping int64 torch ops (to count surviving inputs of a batch exactly)."""
    def c(v):
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    def srl(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)
    x = a * c(0x9E3779B97F4A7C15) + b * c(0xD1B54A32D192ED03) + c(seed)
    x = (x ^ srl(x, 30)) * c(0xBF58476D1CE4E5B9)
    x = (x ^ srl(x, 27)) * c(0x94D049BB133111EB)
    x = x ^ srl(x, 31)
    return srl(x, 32)


def q_threshold(q):
    t = float(np.float32(q)) * 4294967296.0
    return 0 if t <= 0 else min(int(t), 0xFFFFFFFF)


def kept_count(keep_off, seed, q):
    B = keep_off.numel() - 1
    deg = (keep_off[1:] - keep_off[:-1]).long()
    row = torch.repeat_interleave(torch.arange(B, device=keep_off.device), deg)
    j = torch.arange(int(keep_off[-1].item()), device=keep_off.device) - keep_off[:-1].long()[row]
    return int((hash_u32_torch(seed, row, j) >= q_threshold(q)).sum().item())


def batch_row_stats(indptr, indices, uid, iid, keep_off, seed, q):
    """Exact row statistics of ONE batch, counted on the device: per key class (W = input rows of the kept history items,
    V = user rows, O = W2T output rows) the number of touches (occurrences) and of DISTINCT rows, and the rows a single triple
    touches (the forward kernel updates those in place).  These are what the dedup-aware byte model of the roofline needs."""
    B = uid.numel()
    dev = uid.device
    deg = (keep_off[1:] - keep_off[:-1]).long()
    row = torch.repeat_interleave(torch.arange(B, device=dev), deg)
    j = torch.arange(int(keep_off[-1].item()), device=dev) - keep_off[:-1].long()[row]
    keep = hash_u32_torch(seed, row, j) >= q_threshold(q)
    pos = (indptr[uid.long()][row] + j)[keep]
    items = indices[pos]
    st = {'B': B, 'history_items': int(deg.sum().item()), 'occ_W': int(keep.sum().item()), 'dist_W': int(torch.unique(items).numel())}
    for name, col in (('V', uid), ('O', iid)):
        _, cnt = torch.unique(col.long(), return_counts=True)
        st['dist_' + name], st['solo_' + name] = int(cnt.numel()), int((cnt == 1).sum().item())
    return st


def byte_model(st, k, S, fused_solo):
    """Dedup-aware algorithmic HBM bytes of one sparse step, rows of 4k bytes, S optimizer slots per parameter:
      forward kernel : gathers one row per OCCURRENCE (kept W rows + V row + W2T row), writes dz1[b] for every triple and g2[b] for the
                       triples whose W2T row is shared; the V / W2T rows only this triple touches are updated in place from registers
                       (read S slot rows, write parameter + S slot rows) when the touch list was prepared ahead (fused_solo);
                       + the CSR indices of the batch users (4 B per history item) and 40 B of ids / offsets per triple
      reduction      : one read-modify-write of parameter + S slot rows per DISTINCT remaining row: (2 + 2S) rows, + 8 B per touch
                       (sorted key, sample).  The gradient rows it sums (dz1 / g2, one read per occurrence) were written by the forward
                       kernel just before — 2B rows = 67 MB at B = 65 536 — and are ASSUMED served by L2 / the 256 MiB Infinity Cache:
                       they are reported as cache_bytes, not as HBM bytes.
    Returns bytes per launch of each kernel."""
    row = 4.0 * k
    B = st['B']
    solo_V, solo_O = (st['solo_V'], st['solo_O']) if fused_solo else (0, 0)
    fwd = row * (st['occ_W'] + 2 * B) + row * (B + (B - solo_O)) + row * (solo_V + solo_O) * (1 + 2 * S) + 4.0 * st['history_items'] + 40.0 * B
    n_touch = st['occ_W'] + (B - solo_V) + (B - solo_O)
    red = row * (st['dist_W'] + st['dist_V'] - solo_V + st['dist_O'] - solo_O) * (2 + 2 * S) + 8.0 * n_touch
    return {'k_sampled_fwd_bwd': fwd, 'k_seg_reduce': red, 'cache_bytes_k_seg_reduce': row * n_touch}


def kernel_source_hash():
    """sha256 over the sources libdrx.so is built from: a PMC profile is only quoted for the code it was taken on."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'drecpy_amd', 'csrc')
    for name in sorted(os.listdir(src)) + ['../../include/drx.h']:
        if name.endswith(('.hip', '.hpp', '.cpp', '.h')):
            with open(os.path.join(src, name), 'rb') as f:
                h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def cpu_baseline(eng, hist_indptr, hist_indices, batch, seed, budget_s=12.0, n_cpu=1024, optimizer='adagrad'):
    """Times the CPU oracle (oracle/cdae_oracle.py sparse_step: the NumPy restatement, 'port') on the first n_cpu triples
    of one bench batch.  The tables are compacted to the rows that sample touches (same arithmetic per sample; the
    CPU sees a cache-friendlier table than the GPU does)."""
    from oracle import cdae_oracle as co
    n_cpu = min(n_cpu, batch[0].numel())
    uid, iid, y = [t[:n_cpu].cpu().numpy() for t in batch[:3]]
    ip = hist_indptr.cpu().numpy() if hist_indptr.numel() < 50_000_000 else None
    thr = q_threshold(Q)
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
    st = co.sparse_state(p, optimizer)
    cu = np.array([users[int(u)] for u in uid])
    ci = np.array([items[int(i)] for i in iid])
    t0 = time.perf_counter()
    n_done = 0
    while time.perf_counter() - t0 < budget_s:
        co.sparse_step(p, st, n_done, cu, ci, y, kept, float(np.float32(Q)), 1e-3 if optimizer == 'adam' else LR, REG, 'bce', optimizer)
        n_done += 1
    dt = time.perf_counter() - t0
    return {'value': n_cpu * n_done / dt, 'unit': 'samples/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n_done} steps of the first {n_cpu} triples of one bench batch (same tables, compacted to touched '
                      f'rows), NumPy restatement oracle/cdae_oracle.py:sparse_step, {dt:.1f} s, host has {os.cpu_count()} cpus'}


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


def hr_at_10(dev, with_cpu=True):
    """The HR@10 half of BASELINE.json's metric: CDAE in REFERENCE mode with the README configuration (K=50, q=0.2, BCE,
    100 one-batch epochs of 64, lr 1e-3, reg 1e-3, neg_ratio 5, seed 10) on the ml-100k-shaped synthetic set, leave-10-out,
    evaluated with the protocol of examples/cdae.py:15-17.  No MovieLens files exist offline: not comparable with
    README.md:140-141 (0.5536 on the real ml-100k); an untrained model scores ~10/101."""
    from drecpy_amd import synth
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Evaluation import leave_k_out, ranking_evaluation
    from drecpy_amd.Recommender import CDAE
    U, N, md, mn, a = synth.SHAPES['ml-100k']
    ip, idx = synth.synth_history(U, N, md + 12, mn, a, seed=0)
    ip, idx = ip.numpy(), idx.numpy()
    rng = np.random.RandomState(0)
    user = np.repeat(np.arange(U), np.diff(ip)) + 1
    perm = rng.permutation(len(user))
    ds = InteractionDataset.read_df({'user': user[perm], 'item': (idx.astype(np.int64) + 1)[perm],
                                     'interaction': rng.randint(1, 6, size=len(user))[perm]}, verbose=False)
    tr, te = leave_k_out(ds, k=10, min_user_interactions=10, seed=10, verbose=False)
    m = CDAE(hidden_factors=50, corruption_level=0.2, loss='bce', seed=10, verbose=False, device=str(dev))
    m.fit(tr, epochs=2, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)       # loads the dense-mode kernels once
    t0 = time.perf_counter()
    m.fit(tr, epochs=100, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)     # a fresh fit: new tables, new sampler
    torch.cuda.synchronize()
    fit_s = time.perf_counter() - t0
    res = ranking_evaluation(m, te, k=[1, 5, 10], novelty=True, n_test_users=100, n_pos_interactions=1, n_neg_interactions=100,
                             generate_negative_pairs=True, seed=10, verbose=False)
    # the same configuration over a fit long enough for set-up and the GPU's clock ramp not to dominate (the metric's "CDAE ml-100k"
    # training rate): a third fresh fit, 5000 one-batch epochs
    t0 = time.perf_counter()
    m2 = CDAE(hidden_factors=50, corruption_level=0.2, loss='bce', seed=10, verbose=False, device=str(dev))
    m2.fit(tr, epochs=5000, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
    torch.cuda.synchronize()
    long_s = time.perf_counter() - t0
    cpu_ref = cpu_reference_fit(tr) if with_cpu else None
    return {'value': res['HitRatio@10'], 'ndcg_at_10': res['NDCG@10'], 'cpu_baseline_reference_mode': cpu_ref,
            'fit_seconds_100_steps_of_64': round(fit_s, 3),
            'fit_samples_per_s_incl_host': round(6400 / fit_s, 1),
            'fit_seconds_5000_steps_of_64': round(long_s, 3), 'fit_samples_per_s_5000_steps_incl_host_and_setup': round(320000 / long_s, 1),
            'setup': 'CDAE reference mode (dense Keras Adam), README.md:106-114 configuration, ml-100k-shaped synthetic '
                     f'({len(tr)} train / {len(te)} test rows), protocol examples/cdae.py:15-17'}


def run_columns(args, rank, world, dev, dist, debug_gloo, rccl1):
    """Column-sharded layout (dist.ColumnShardedCdae): the job's global batch (B per GPU x N GPUs) is drawn identically on every
    rank, each rank trains its K/N columns of every table on all of it; one all-reduce of B_global floats per step."""
    from drecpy_amd import synth
    from drecpy_amd.dist import ColumnShardedCdae
    U, N, md, mn, alpha = synth.SHAPES[args.workload]
    if args.users:
        U = args.users
    Bg = args.batch * world
    t_setup = time.time()
    indptr, indices = synth.synth_history(U, N, md, mn, alpha, seed=0, device=dev)          # every rank: all users
    nnz = int(indptr[-1].item())
    model = ColumnShardedCdae(U, N, K, rank, world, dev, indptr, indices, seed=10, lr=1e-3 if args.optimizer == 'adam' else LR, reg=REG,
                              optimizer=args.optimizer, q=Q, cpu_staging=debug_gloo, force_collectives=rccl1)
    eng = model.engine
    uid, iid, y, keep_off = eng.sample_device(Bg, NEG_RATIO, 1000, n_items=N)
    rows_per_sample = kept_count(keep_off, 1000, Q) / Bg + 2.0
    f_solo = 0.0
    for col in (uid.long(), iid.long()):
        _, inv, cnt = torch.unique(col, return_inverse=True, return_counts=True)
        f_solo += float((cnt[inv] == 1).float().mean().item())
    setup_s = time.time() - t_setup
    seed_of = lambda s: 5000 + 7919 * s                                                              # same seeds on every rank
    emu = int(os.environ.get('DRX_BENCH_EMULATE_RANKS', 0))
    if args.prepare == 'auto':
        args.prepare = 'turns' if max(world, emu) >= 4 else 'local'
    if emu > 1 and world == 1 and args.prepare != 'local':
        # debugging aid (one GPU standing in for one rank of `emu`): the batches cycle and what the other ranks would send comes
        # from a cache filled at the first encounter; a device copy stands in for the broadcast / all-gather
        seed_of = lambda s: 5000 + 7919 * (s % args.n_batches)
        cache, recv = {}, [None, None]

        def emulated_build(s, bt, out):
            c = s % args.n_batches
            if c not in cache:
                cache[c] = eng.prepare_sparse(bt).clone()
            return eng.prepare_sparse(bt, out) if s % emu == 0 else eng.prep_buffer(bt, out)

        def emulated_deliver(s, bt, out):
            if s % emu:
                n = eng.prep_result_bytes(bt)
                out[:n].copy_(cache[s % args.n_batches][:n])

        def emulated_parts(s, bt, out):
            c = s % args.n_batches
            if c not in cache:
                cache[c] = torch.cat([eng.prepare_part(bt, r, emu).clone() for r in range(emu)])
            part = eng.prepare_part(bt, 0, emu, slot=s % 2)
            n = cache[c].numel()
            if recv[s % 2] is None or recv[s % 2].numel() < n:
                recv[s % 2] = torch.empty(int(n * 1.05), dtype=torch.uint8, device=dev)
            got = recv[s % 2][:n]
            got.copy_(cache[c])
            got[:part.numel()].copy_(part)
            out, model._oflow[s % 2] = eng.prepare_assemble(bt, got, emu, out, model._oflow[s % 2])
            return out
        model.prepare_mode = args.prepare
        model.prepare, model.build_in_turns, model.deliver_in_turns = emulated_parts, emulated_build, emulated_deliver
    elif world > 1:
        model.prepare_mode = args.prepare
    pipe = model.pipeline(Bg, NEG_RATIO, seed_of, seed_of)
    for _ in range(args.warmup):
        pipe.run_step()
    EVERY = max(1, int(os.environ.get('DRX_BENCH_EVENTS_EVERY', 4)))
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(7)] for _ in range(args.steps)]
    for es in evs:
        for e in es:
            e.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        if s % EVERY == 0:
            evs[s][0].record()                        # [0,1): forward half + all-reduce of the partial dot products
            pipe.run_step(events=evs[s][1:])
        else:
            pipe.run_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ph = np.array([[es[i].elapsed_time(es[i + 1]) for i in range(6)] for es in evs[::EVERY]]).mean(axis=0)
    names = ['k_kshard_fwd+allreduce(dot)', 'k_kshard_rest', 'touch_sort(overlapped on side stream)', 'k_seg_reduce',
             'k_sparse_tail_a(short spans | bias partials)', 'k_sparse_tail_b(long spans | bias update)']
    kl = model.k_hi - model.k_lo
    S_opt = 2.0 if args.optimizer == 'adam' else 1.0
    alg_upd = Bg * 4.0 * kl * (rows_per_sample - f_solo) * (2.0 + 2.0 * S_opt)      # this rank's columns of the global batch
    alg_fwd = Bg * 4.0 * kl * rows_per_sample
    dom, dom_ms, dom_alg = ('k_seg_reduce', ph[3], alg_upd) if ph[3] >= ph[0] else ('k_kshard_fwd+allreduce(dot)', ph[0], alg_fwd)
    step_alg = Bg * 4.0 * kl * rows_per_sample * (3.0 + 2.0 * S_opt)
    if rank == 0:
        copy_gbs = hbm_copy_gbs(dev)
        out = {'metric': 'training samples/sec (user-item pairs)', 'value': Bg * args.steps / dt, 'unit': 'samples/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': f'CDAE hidden_factors={K} sampled-output sparse-{"Adam (lazy)" if S_opt == 2.0 else "Adagrad"} on '
                                      f'{args.workload}-shaped synthetic ({U} users x {N} items, {nnz} positives), corruption {Q}, neg_ratio {NEG_RATIO}',
                          'batch_per_gpu': args.batch, 'global_batch': Bg, 'rows_per_sample': round(rows_per_sample, 3),
                          'sole_toucher_rows_per_sample': round(f_solo, 3),
                          'batches': 'fresh device-sampled global batch every step, drawn identically on every rank (sampler running ahead on a side stream)',
                          'sharding': f'columns: every rank holds all rows x {kl} of {K} columns and trains on the whole global batch; '
                                      f'one all-reduce of {Bg} floats per step; touch list '
                                      + {'local': 'sorted whole on every rank', 'turns': 'of step s sorted by rank s % N and broadcast on a side stream',
                                         'parts': 'sorted in parts (1/N per rank) + one all-gather on a side stream'}[model.prepare_mode]},
               'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': dom_alg / (dom_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': dom_alg / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                            'algorithmic_bytes_per_launch': dom_alg, 'avg_launch_ms': float(dom_ms), 'timed_launches': int(len(evs[::EVERY])),
                            'whole_step_achieved': step_alg / (dt / args.steps) / 1e9,
                            'whole_step_frac': step_alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 'hbm_copy_achievable': copy_gbs,
                            'note': 'per-rank figures: this rank\'s K/N columns of the global batch; algorithmic bytes per SURVEY 8d charge a '
                                    'read-modify-write per touched-row occurrence, the kernel merges occurrences first (DESIGN.md section 3)'},
               'phases_ms': {n: float(v) for n, v in zip(names, ph)}, 'setup_s': round(setup_s, 1), 'cpu_baseline': None, 'hr_at_10': None}
        print(json.dumps(out), flush=True)


def main():
    global K
    args = parse()
    if args.k:
        K = args.k
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    debug_gloo = os.environ.get('DRX_BENCH_BACKEND') == 'gloo'     # debugging aid: N ranks sharing ONE GPU, host-staged exchange
    dev_index = local_rank % torch.cuda.device_count() if debug_gloo else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    import torch.distributed as dist
    rccl1 = world == 1 and os.environ.get('DRX_BENCH_RCCL1') == '1'     # debugging aid: 1-rank RCCL communicator, real collectives
    if rccl1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29455')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if debug_gloo:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    if (world > 1 and args.layout == 'columns' and not args.force_sharded) or args.force_columns:
        run_columns(args, rank, world, dev, dist, debug_gloo, rccl1)
        if world > 1 or rccl1:
            dist.barrier()                      # rank 0 prints (and times a device copy) after the others are done
            dist.destroy_process_group()
        return

    from drecpy_amd import synth
    from drecpy_amd.engine import CdaeEngine
    U, N, md, mn, alpha = synth.SHAPES[args.workload]
    if args.users:
        U = args.users
    B = args.batch
    lo, hi = U * rank // world, U * (rank + 1) // world
    t_setup = time.time()
    indptr, indices = synth.synth_history(U, N, md, mn, alpha, seed=0, device=dev, user_lo=lo, user_hi=hi)
    nnz_local = int(indptr[-1].item())

    if world == 1 and not args.force_sharded:
        eng = CdaeEngine(hi - lo, N, K, device=dev)
        eng.init_glorot_device(10)
        eng.set_history(indptr, indices)
        eng.init_optimizer(args.optimizer, 1e-3 if args.optimizer == 'adam' else LR, REG)
        stepper = None
    else:
        from drecpy_amd.dist import ShardedCdae
        stepper = ShardedCdae(U, N, K, rank, world, dev, indptr, indices, seed=10, lr=LR, reg=REG, q=Q,
                              cpu_staging=debug_gloo, force_collectives=(world == 1 and os.environ.get('DRX_BENCH_RCCL1') == '1'))
        eng = stepper.engine

    micro = max(1, args.micro)       # > 1: micro-batches whose exchanges overlap each other's compute (measured at world 1: the split costs more than it hides)
    # ---- pre-sampled batches, resident in HBM -------------------------------------------------------------
    batches, structs, kept_tot = [], [], 0
    for i in range(args.n_batches):
        seed = 1000 + 7919 * i + 104729 * rank
        uid, iid, y, keep_off = eng.sample_device(B, NEG_RATIO, seed, n_items=N)
        torch.cuda.synchronize()
        n_slots = int(keep_off[-1].item())
        kept_tot += kept_count(keep_off, seed, Q)
        batches.append((uid, iid, y, keep_off, seed))
        if stepper is not None and micro > 1:
            # micro-batches of a sharded step: the triples split by uid so that no user is in two of them
            parts = []
            for m in range(micro):
                ix = torch.nonzero(uid % micro == m).squeeze(1)
                um = uid[ix].contiguous()
                deg = (indptr[um.long() + 1] - indptr[um.long()]).to(torch.int32)
                ko = torch.zeros(um.numel() + 1, dtype=torch.int32, device=dev)
                ko[1:] = torch.cumsum(deg, 0)
                parts.append(eng.make_batch(um, iid[ix].contiguous(), y[ix].contiguous(), keep_off=ko, q=Q, mask_seed=seed + 1000003 * m,
                                            n_touch_slots=int(ko[-1].item())))
            structs.append(([p_[0] for p_ in parts], parts))
        else:
            bt, alive = eng.make_batch(uid, iid, y, keep_off=keep_off, q=Q, mask_seed=seed, n_touch_slots=n_slots)
            structs.append((bt, alive))
    rows_per_sample = kept_tot / (args.n_batches * B) + 2.0          # R: kept W rows + V row + W2T row
    setup_s = time.time() - t_setup

    # The touch list of a batch (sorted row keys) does not depend on the parameters: it is prepared for batch t+1 on a
    # side stream while batch t trains (single GPU; the sharded path prepares inline).
    overlap = not args.no_overlap
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream(priority=-1) if overlap else None
    prep_bufs = [None, None]
    prep_done = [torch.cuda.Event(), torch.cuda.Event()]
    step_done = [torch.cuda.Event(), torch.cuda.Event()]

    fresh = overlap and stepper is None and not args.presampled
    # fresh mode: the device PointSampler draws a NEW batch for every step, two steps ahead on a side stream, and the batch's
    # touch list is sorted one step ahead — drecpy_amd.engine.SampledPipeline, the code path of
    # CDAE.fit(mode='sampled', device_sampler=True).
    spipe = None
    if fresh:
        from drecpy_amd.engine import SampledPipeline
        spipe = SampledPipeline(eng, B, NEG_RATIO, Q, lambda s: 5000 + 7919 * s + 104729 * rank,
                                lambda s: 5000 + 7919 * s + 104729 * rank, n_items=N, prep_ahead=int(os.environ.get('DRX_PREP_AHEAD', 3)))

    def batch_of(s):
        return structs[s % len(structs)][0]

    def prepare(s):
        bt = batch_of(s)
        side.wait_event(step_done[s % 2])            # the buffer's previous user (step s-2) must be finished
        with torch.cuda.stream(side):
            prep_bufs[s % 2] = eng.prepare_sparse(bt, prep_bufs[s % 2])
            prep_done[s % 2].record(side)

    pipe = None
    if stepper is not None and overlap:
        from drecpy_amd.dist import ShardedPipeline
        fresh_sharded = not args.presampled and micro == 1 and not debug_gloo
        if fresh_sharded:      # a new device-sampled batch of this rank's users every step, drawn ahead on its own stream
            from drecpy_amd.engine import DeviceBatchSource
            source = DeviceBatchSource(eng, B, NEG_RATIO, Q, lambda s: 5000 + 7919 * s + 104729 * rank,
                                       lambda s: 5000 + 7919 * s + 104729 * rank, n_items=N)
        else:
            source = lambda s: structs[s % len(structs)][0]
        pipe = ShardedPipeline(stepper, source, args.warmup + args.steps)

    def run_step(s, events=None, last=False):
        if pipe is not None:           # keys of batch s+1 and counts of batch s+2 travel ahead of step s (dist.ShardedPipeline)
            pipe.run_step(events=events)
            return
        if spipe is not None:
            spipe.run_step(events=events)
            return
        if not overlap:
            bt = batch_of(s)
            if stepper is not None:
                stepper.step(s, bt, events=events)
            else:
                eng.step_sparse(s, bt, 'bce', events=events)
            return
        if not last:
            prepare(s + 1)
        bt = batch_of(s)
        main.wait_event(prep_done[s % 2])
        eng.step_sparse(s, bt, 'bce', events=events, prepared=prep_bufs[s % 2])
        step_done[s % 2].record(main)

    for e in step_done:
        e.record(main)
    if overlap and pipe is None and spipe is None:
        prepare(0)
    for s in range(args.warmup):
        run_step(s)
    # The library records the phase events (hipEventRecord between kernels) only on every EVERY-th timed step: six event
    # records per step cost ~4 % of the step (137.7 vs 143.2 vs 144.5 M samples/s at EVERY = 1 / 4 / none).
    EVERY = max(1, int(os.environ.get('DRX_BENCH_EVENTS_EVERY', 4)))
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(args.steps)]
    for es in evs:
        for e in es:
            e.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        run_step(args.warmup + s, events=(evs[s] if (s % EVERY == 0) else None), last=(s == args.steps - 1))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    phases = np.array([[es[i].elapsed_time(es[i + 1]) for i in range(5)] for es in evs[::EVERY]])     # ms
    ph = phases.mean(axis=0)
    # algorithmic bytes per launch: forward reads 4K*R per sample; the update reads and
    # writes parameter + S optimizer slots per touched-row occurrence: 4K*R*(2+2S), S = 1 for Adagrad.
    # With a touch list prepared ahead, V / W2T rows touched by a single triple of the batch are updated by the forward kernel
    # (k_mark_solo): their (2+2S) update bytes are charged to it instead of the segmented reduction.
    f_solo = 0.0
    if stepper is None and overlap:
        uid0, iid0 = batches[0][0].long(), batches[0][1].long()
        for col in (uid0, iid0):
            _, inv, cnt = torch.unique(col, return_inverse=True, return_counts=True)
            f_solo += float((cnt[inv] == 1).float().mean().item())
    # optimizer slots per parameter (row-wise Adagrad keeps 1/K of a slot: one float per row)
    S_opt = {'adam': 2.0, 'adagrad': 1.0, 'rowwise_adagrad': 1.0 / K}[args.optimizer] if stepper is None else 1.0
    alg_fwd = B * 4.0 * K * (rows_per_sample + f_solo * (2.0 + 2.0 * S_opt))
    alg_upd = B * 4.0 * K * (rows_per_sample - f_solo) * (2.0 + 2.0 * S_opt)
    if stepper is None:
        names = ['k_sampled_fwd_bwd', 'touch_sort(overlapped on side stream)' if overlap else 'touch_sort', 'k_seg_reduce',
                 'k_sparse_tail_a(short spans | bias partials)', 'k_sparse_tail_b(long spans | bias update)']
        dom, dom_ms, dom_alg = ('k_seg_reduce', ph[2], alg_upd) if ph[2] >= ph[0] else ('k_sampled_fwd_bwd', ph[0], alg_fwd)
    else:
        names = ['row_gather+first_row_exchange', 'fwd_bwd+local_reduce(+overlapped exchanges)', 'rest_of_grad_exchange', 'owner_apply', 'bias_allreduce']
        # forward reads one row per occurrence, the local reduce one gradient row per occurrence
        dom, dom_ms, dom_alg = 'k_shard_fwd_bwd+k_seg_reduce<LocalPolicy>', ph[1], 2.0 * B * 4.0 * K * rows_per_sample
    achieved = dom_alg / (dom_ms * 1e-3) / 1e9
    step_alg = B * 4.0 * K * rows_per_sample * (3.0 + 2.0 * S_opt)
    # ---- dedup-aware byte model: exact occurrence / distinct-row counts of batches of the TIMED region ----------------------------
    dedup = None
    if stepper is None:
        picks = sorted(set(int(x) for x in np.linspace(0, args.steps - 1, 6)))
        stats = []
        for s_ in picks:
            if fresh:                                   # the pipeline drew step (warmup + s_) from these seeds: draw it again
                seed_ = 5000 + 7919 * (args.warmup + s_) + 104729 * rank
                u_, i_, _, ko_ = eng.sample_device(B, NEG_RATIO, seed_, n_items=N)
            else:
                u_, i_, _, ko_, seed_ = batches[(args.warmup + s_) % len(batches)]
            stats.append(batch_row_stats(indptr, indices, u_, i_, ko_, seed_, Q))
        mean_st = {k_: float(np.mean([st[k_] for st in stats])) for k_ in stats[0]}
        bm = byte_model(mean_st, K, S_opt, fused_solo=overlap)
        dedup = {'batches_counted': len(stats), 'per_batch_mean': {k_: round(v, 1) for k_, v in mean_st.items()}, 'bytes_per_launch': bm}
    kernel_hash = kernel_source_hash()
    # HBM traffic from the PMC passes (profiles/pmc_traffic.json), only when that profile was taken on this very workload AND on
    # these very kernel sources (scripts/profile_round.sh stores their hash); rocprofv3 cannot run inside the bench itself.
    traffic_of, traffic_note = {}, 'no PMC profile for this workload'
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            pmc = json.load(f)
        m = pmc['_meta']
        if m['workload'] == args.workload and m['batch_per_gpu'] == B and m['n_gpus'] == world and not args.users and K == 128 \
                and m.get('optimizer', 'adagrad') == args.optimizer:
            if m.get('kernel_source_hash') == kernel_hash:
                traffic_of = {k_.replace('drx::', ''): v['hbm_bytes_per_launch'] for k_, v in pmc['kernels'].items()}
                traffic_note = f"profiles/pmc_traffic.json ({m.get('round')}), same kernel sources ({kernel_hash})"
            else:
                traffic_note = f"profiles/pmc_traffic.json is STALE: taken on kernel sources {m.get('kernel_source_hash')}, this tree is {kernel_hash}"
    except (OSError, KeyError, ValueError):
        pass
    traffic = traffic_of.get(dom)
    copy_gbs = hbm_copy_gbs(dev) if rank == 0 else None
    if rank == 0:
        out = {
            'metric': 'training samples/sec (user-item pairs)', 'value': world * B * args.steps / dt, 'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'CDAE hidden_factors={K} sampled-output sparse-{dict(adam="Adam (lazy)", adagrad="Adagrad", rowwise_adagrad="row-wise Adagrad")[args.optimizer if stepper is None else "adagrad"]} on {args.workload}-shaped synthetic '
                                   f'({U} users x {N} items, {nnz_local * world if world > 1 else nnz_local} positives), '
                                   f'corruption {Q}, neg_ratio {NEG_RATIO}',
                       'batch_per_gpu': B, 'global_batch': B * world, 'rows_per_sample': round(rows_per_sample, 3),
                       'sole_toucher_rows_per_sample': round(f_solo, 3),
                       'touch_list': ('keys exchanged one batch ahead, counts two (dist.ShardedPipeline)' if pipe is not None else 'prepared one batch ahead on a side stream') if overlap else 'inline',
                       'batches': 'fresh device-sampled batch every step (sampler two steps ahead on the side stream)' if (fresh or (pipe is not None and fresh_sharded))
                       else f'{args.n_batches} pre-sampled batches cycled',
                       'micro_batches': (micro if stepper is not None else None),
                       'sharding': ('single GPU' if stepper is None else 'row-sharded code path at world 1') if world == 1 else f'users row-sharded x{world}, item rows all-to-all'},
            'roofline': None,
            'phases_ms': {n: float(v) for n, v in zip(names, ph)},
            'setup_s': round(setup_s, 1),
            'host_issue_ms_per_step': ([round(1e3 * t / (args.warmup + args.steps), 4) for t in pipe.host_s + [stepper.wait_s]] if pipe is not None else None),
        }
        ms_of = {'k_sampled_fwd_bwd': float(ph[0]), 'k_seg_reduce': float(ph[2])} if stepper is None else {}
        if dedup is not None:
            # `frac` = dedup-aware algorithmic HBM bytes / measured launch time / peak: a fraction by construction (what has to
            # cross the HBM interface at least once; everything re-read is assumed cached).  `model_*` = SURVEY 8d's
            # per-OCCURRENCE model (charges a read-modify-write per touch: the kernel merges touches first, so it can exceed 1).
            per_kernel = {}
            for kn in ('k_sampled_fwd_bwd', 'k_seg_reduce'):
                byt, ms_ = dedup['bytes_per_launch'][kn], ms_of[kn]
                per_kernel[kn] = {'bytes_per_launch': byt, 'avg_launch_ms': ms_, 'achieved': byt / (ms_ * 1e-3) / 1e9,
                                  'frac': byt / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  'traffic': traffic_of.get(kn),
                                  'traffic_frac': (traffic_of[kn] / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS) if kn in traffic_of else None}
            step_bytes = dedup['bytes_per_launch']['k_sampled_fwd_bwd'] + dedup['bytes_per_launch']['k_seg_reduce']
            step_traffic = sum(traffic_of.get(kn, 0.0) for kn in ('k_sampled_fwd_bwd', 'k_seg_reduce', 'k_sparse_tail_a', 'k_sparse_tail_b')) or None
            dk = per_kernel[dom]
            out['roofline'] = {
                'bound': 'hbm', 'kernel': dom, 'achieved': dk['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': dk['frac'],
                'traffic': traffic, 'traffic_source': traffic_note, 'kernel_source_hash': kernel_hash,
                'traffic_rate': (traffic / (dom_ms * 1e-3) / 1e9) if traffic else None,
                'traffic_frac': (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                'bytes_per_launch': dk['bytes_per_launch'], 'avg_launch_ms': float(dom_ms), 'timed_launches': int(len(phases)),
                'kernels': per_kernel, 'row_counts': dedup['per_batch_mean'], 'batches_counted': dedup['batches_counted'],
                'cache_bytes_k_seg_reduce': dedup['bytes_per_launch']['cache_bytes_k_seg_reduce'],
                'whole_step_bytes': step_bytes, 'whole_step_achieved': step_bytes / (dt / args.steps) / 1e9,
                'whole_step_frac': step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                'whole_step_traffic': step_traffic,
                'whole_step_traffic_frac': (step_traffic / (dt / args.steps) / 1e9 / HBM_PEAK_GBS) if step_traffic else None,
                'model_bytes_per_launch': dom_alg, 'model_frac': achieved / HBM_PEAK_GBS,
                'model_whole_step_frac': step_alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                'hbm_copy_achievable': copy_gbs,
                'definition': 'frac = dedup-aware algorithmic HBM bytes (bench.py:byte_model: one gather read per occurrence, dz1/g2 written once, '
                              'one read-modify-write of parameter + slots per DISTINCT row, gradient re-reads assumed cached) / HIP-event launch time / 8 TB/s; '
                              'traffic = PMC bytes per launch (FETCH_SIZE x2 + WRITE_SIZE; counts Infinity-Cache hits); model_* = SURVEY 8d per-occurrence bytes'}
        else:
            out['roofline'] = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': None,
                               'traffic': None, 'model_bytes_per_launch': dom_alg, 'model_frac': achieved / HBM_PEAK_GBS,
                               'avg_launch_ms': float(dom_ms), 'timed_launches': int(len(phases)), 'hbm_copy_achievable': copy_gbs,
                               'definition': 'row-sharded path: only the per-occurrence model (SURVEY 8d) is evaluated here'}
        if world == 1 and not args.no_cpu_baseline:
            uid, iid, y, keep_off, seed = batches[0]
            out['cpu_baseline'] = cpu_baseline(eng, indptr, indices, (uid, iid, y, keep_off), seed, optimizer=args.optimizer)
        else:
            out['cpu_baseline'] = None
        out['hr_at_10'] = hr_at_10(dev, with_cpu=not args.no_cpu_baseline) if (world == 1 and not args.no_hr and not args.users) else None
        out['cpu_baseline_reference_mode'] = (out['hr_at_10'] or {}).get('cpu_baseline_reference_mode')
        print(json.dumps(out), flush=True)
    if world > 1 or rccl1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
