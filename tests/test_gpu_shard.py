"""GPU parity of the row-sharded step's HIP kernels (drx_shard_*): world 1 in-process, and world 2 as two processes
sharing the one GPU of the box (gloo + host-staged exchange stand in for RCCL) — against the single-process oracle step
on the concatenated batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from helpers import new_rendezvous, retry_infra  # noqa: E402

pytestmark = pytest.mark.gpu
U, N, K, B, STEPS, Q = 300, 411, 50, 512, 4, 0.2
K = int(os.environ.get('DRX_TEST_SHARD_K', K))          # (spawned ranks re-import this module: a test hands its K over this way)
N_LIVE = N
N = int(os.environ.get('DRX_TEST_SHARD_N', N))          # > 411: the 411 live items spread over a wider id range (room for exchange chunks)
CHUNKS = int(os.environ.get('DRX_TEST_SHARD_CHUNKS', 1))


def _problem(world):
    from oracle import cdae_oracle as co
    from helpers import synth_history
    rng = np.random.default_rng(5)
    p = co.init_params(rng, U, N, K, np.float32)
    indptr, indices = synth_history(rng, U, N_LIVE, 12, zipf=1.2)
    spread = np.arange(N_LIVE)
    if N > N_LIVE:          # an exchange chunk spans >= 8192 wire keys: live items in every chunk need a wide id range
        spread = np.random.default_rng(6).permutation(np.sort(np.random.default_rng(7).choice(N, size=N_LIVE, replace=False)))
        indices = spread[indices].astype(np.int32)
        for u in range(U):
            indices[indptr[u]:indptr[u + 1]].sort()
    batches = []
    for s in range(STEPS):
        per_rank = []
        for r in range(world):
            lo, hi = U * r // world, U * (r + 1) // world
            per_rank.append((rng.integers(lo, hi, size=B), spread[rng.integers(0, N_LIVE, size=B)], (rng.random(B) < 0.3).astype(np.float32),
                             500 + 31 * s + r))
        batches.append(per_rank)
    return p, indptr, indices, batches


def _oracle(world, micro=1):
    from oracle import cdae_oracle as co
    p, indptr, indices, batches = _problem(world)
    p = {k: v.astype(np.float64) for k, v in p.items()}
    st = co.sparse_state(p, 'adagrad')
    losses = []
    for s in range(STEPS):
        uid = np.concatenate([batches[s][r][0] for r in range(world)])
        iid = np.concatenate([batches[s][r][1] for r in range(world)])
        y = np.concatenate([batches[s][r][2] for r in range(world)])
        kept = [None] * len(uid)
        base = 0
        for r in range(world):
            u_r, _, _, seed = batches[s][r]
            for m in range(micro):                      # a sample's corruption mask is keyed by its micro-batch position
                for b, j in enumerate(np.flatnonzero(u_r % micro == m)):
                    row = indices[indptr[u_r[j]]:indptr[u_r[j] + 1]]
                    kf = co.drx_hash_u32(seed + (100 * m if micro > 1 else 0), np.full(len(row), b), np.arange(len(row))) >= co.q_threshold(Q)
                    kept[base + j] = row[kf].tolist()
            base += len(u_r)
        lval, _ = co.sparse_step(p, st, s, uid, iid, y, kept, float(np.float32(Q)), 0.05, 1e-3, 'bce', 'adagrad')
        losses.append(lval)
    return p, losses


def _run_rank(rank, world, staged, force=False, pipelined=False, micro=1, bypass=True, transport=None, phases=None):
    from drecpy_amd.dist import ShardedCdae
    p, indptr, indices, batches = _problem(world)
    lo, hi = U * rank // world, U * (rank + 1) // world
    lip = indptr[lo:hi + 1] - indptr[lo]
    lidx = indices[indptr[lo]:indptr[hi]]
    m = ShardedCdae(U, N, K, rank, world, 'cuda:0', lip, lidx, q=Q, cpu_staging=staged, force_collectives=force, self_bypass=bypass,
                    chunks=CHUNKS, transport=transport, phases=phases)
    assert m.chunks == CHUNKS, (m.chunks, CHUNKS)
    if transport == 'rccl':
        assert type(m.xfer).__name__ == 'RcclTransport' and m.phases == (True if phases is None else phases)
    m.set_params_global(**p)
    losses, made = [], {}

    def batch_of(s):
        if s not in made:
            uid, iid, y, seed = batches[s][rank]
            if micro == 1:
                made[s] = [m.ops.make_batch(uid - lo, iid, y, q=Q, mask_seed=seed)]
            else:                                       # micro-batches with disjoint users
                made[s] = [m.ops.make_batch(uid[ix] - lo, iid[ix], y[ix], q=Q, mask_seed=seed + 100 * k)
                           for k, ix in enumerate(np.flatnonzero(uid % micro == k) for k in range(micro))]
        return made[s][0][0] if micro == 1 else [bt for bt, _ in made[s]]
    if pipelined:
        from drecpy_amd.dist import ShardedPipeline
        pipe = ShardedPipeline(m, batch_of, STEPS)
        losses = [pipe.run_step(want_loss=True) for _ in range(STEPS)]
    else:
        for s in range(STEPS):
            losses.append(m.step(s, batch_of(s), want_loss=True))
    torch.cuda.synchronize()
    return m.ops.get_params(), losses, (lo, hi)


def _check(world, results, micro=1):
    p, want_losses = _oracle(world, micro)
    ipr = (N + world - 1) // world
    for r, (g, losses, (ulo, uhi)) in enumerate(results):
        lo, hi = r * ipr, min(N, (r + 1) * ipr)
        tol = dict(rtol=0, atol=3e-6)
        np.testing.assert_allclose(g['W'][:hi - lo], p['W'][lo:hi], **tol)
        np.testing.assert_allclose(g['W_'][:, :hi - lo], p['W_'][:, lo:hi], **tol)
        np.testing.assert_allclose(g['b_'][:hi - lo], p['b_'][lo:hi], **tol)
        np.testing.assert_allclose(g['V'], p['V'][ulo:uhi], **tol)
        np.testing.assert_allclose(g['b'], p['b'], **tol)
        np.testing.assert_allclose(losses, want_losses, rtol=1e-5)


@pytest.mark.parametrize('bypass', [True, False])
@pytest.mark.parametrize('pipelined,micro', [(False, 1), (True, 1), (True, 2)])
def test_sharded_world1_matches_oracle(pipelined, micro, bypass):
    """bypass: the rank's own rows are read from its tables and its own gradient chunk stays where the reduction left it
    (DRX_SHARD_SELF_BYPASS) — at world 1 that is every row; off: every row goes through gather / exchange buffer / cache."""
    _check(1, [_run_rank(0, 1, False, pipelined=pipelined, micro=micro, bypass=bypass)], micro)


def _worker(rank, world, rdzv, out, pipelined=False, micro=1, bypass=True):
    dist.init_process_group('gloo', init_method=rdzv, rank=rank, world_size=world)
    res = _run_rank(rank, world, True, pipelined=pipelined, micro=micro, bypass=bypass)
    torch.save(res, f'{out}.{rank}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('pipelined,micro,world,bypass', [(False, 1, 2, True), (True, 1, 2, True), (True, 2, 2, True), (True, 1, 3, True),
                                                          (True, 1, 2, False), (True, 2, 3, False)])
@retry_infra
def test_sharded_world2_on_one_gpu_matches_oracle(tmp_path, pipelined, micro, world, bypass):
    """`world` processes share the one GPU of the box (gloo + host-staged exchanges); world 3 splits 411 items unevenly."""
    out = str(tmp_path / 'shard')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker, args=(world, rdzv, out, pipelined, micro, bypass), nprocs=world, join=True)
    _check(world, [torch.load(f'{out}.{r}', weights_only=False) for r in range(world)], micro)


@pytest.mark.parametrize('world,bypass', [(2, True), (3, False)])
@retry_infra
def test_sharded_ranks_with_the_streamed_local_reduction(tmp_path, monkeypatch, world, bypass):
    """K = 128 (rows of exactly 128 floats: the local reduction is the streamed one) over 2 / 3 ranks sharing the GPU: the parked
    sums go to the chunks of SEVERAL owners (world 3 splits 411 items unevenly), own chunk last with the bypass."""
    import sys as _sys
    monkeypatch.setenv('DRX_TEST_SHARD_K', '128')
    monkeypatch.setattr(_sys.modules[__name__], 'K', 128)
    out = str(tmp_path / 'shard')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker, args=(world, rdzv, out, True, 1, bypass), nprocs=world, join=True)
    _check(world, [torch.load(f'{out}.{r}', weights_only=False) for r in range(world)], 1)


def _worker_rccl(rank, rdzv, out, pipelined, micro, bypass=True):
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method=rdzv, rank=0, world_size=1, device_id=torch.device('cuda', 0))
    res = _run_rank(0, 1, False, force=True, pipelined=pipelined, micro=micro, bypass=bypass)
    torch.save(res, f'{out}.0')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('pipelined,micro,bypass', [(False, 1, False), (True, 1, False), (True, 2, False), (True, 1, True)])
def test_sharded_step_through_rccl_world1(tmp_path, pipelined, micro, bypass):
    """The N-rank call sequence (count / key / row / gradient all-to-all(v), bias all-reduce) on a real 1-rank RCCL
    communicator: device int32 and float32 buffers, uneven-split API, stream ordering with the drx kernels."""
    out = str(tmp_path / 'rccl')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker_rccl, args=(rdzv, out, pipelined, micro, bypass), nprocs=1, join=True)
    _check(1, [torch.load(f'{out}.0', weights_only=False)], micro)


@pytest.mark.parametrize('k', [7, 130, 600, 64, 128, 256])
def test_sharded_world1_other_row_geometries(k, monkeypatch):
    """The shard kernels in the (8,1), (64,1) and (64,4) lane-group geometries (the tests above run K = 50: (16,1)); K = 64 / 128 / 256:
    rows of exactly 64 / 128 / 256 floats, whose local reduction is the STREAMED one (csrc/drx_segstream.hpp with LocalPolicyT: item
    rows' sums parked in the gradient exchange buffer, V rows applied in place)."""
    import sys as _sys
    monkeypatch.setattr(_sys.modules[__name__], 'K', k)
    _check(1, [_run_rank(0, 1, False, pipelined=True)])


# ---- the chunked exchange schedule (r06): every exchange in CHUNKS all-to-alls over key ranges, owner apply of chunk c followed by the
# gather + row exchange of the next step's chunk c -------------------------------------------------------------------------------------
def _chunked(monkeypatch, n_items, chunks, k=None):
    import sys as _sys
    mod = _sys.modules[__name__]
    monkeypatch.setenv('DRX_TEST_SHARD_N', str(n_items))
    monkeypatch.setenv('DRX_TEST_SHARD_CHUNKS', str(chunks))
    monkeypatch.setattr(mod, 'N', n_items)
    monkeypatch.setattr(mod, 'CHUNKS', chunks)
    if k is not None:
        monkeypatch.setenv('DRX_TEST_SHARD_K', str(k))
        monkeypatch.setattr(mod, 'K', k)


@pytest.mark.parametrize('pipelined,micro,bypass,chunks,n_items,k', [(True, 1, False, 4, 9000, 50), (True, 1, True, 2, 5000, 50), (False, 1, False, 2, 5000, 50),
                                                                     (True, 2, False, 2, 5000, 50), (True, 1, False, 4, 9000, 128), (True, 1, True, 8, 17000, 128)])
def test_sharded_world1_chunked_matches_oracle(monkeypatch, pipelined, micro, bypass, chunks, n_items, k):
    _chunked(monkeypatch, n_items, chunks, k)
    _check(1, [_run_rank(0, 1, False, pipelined=pipelined, micro=micro, bypass=bypass)], micro)


@pytest.mark.parametrize('pipelined,micro,world,bypass,chunks,n_items,k', [(True, 1, 2, True, 2, 9000, 50), (True, 1, 3, False, 2, 13000, 50),
                                                                           (True, 2, 2, True, 2, 9000, 50), (True, 1, 2, False, 4, 17000, 128),
                                                                           # (BASELINE configuration 4's world: eight segments per chunk in the owner's kernels)
                                                                           (True, 1, 8, True, 2, 70000, 128), (True, 1, 8, False, 1, 70000, 50), (True, 1, 5, True, 2, 43000, 128)])
@retry_infra
def test_sharded_ranks_on_one_gpu_chunked_match_oracle(tmp_path, monkeypatch, pipelined, micro, world, bypass, chunks, n_items, k):
    """`world` processes share the GPU (gloo + host-staged exchanges); exchanges in 2 / 4 chunks; K = 128: the streamed local reduction
    parks its sums in the units of several owners AND chunks."""
    _chunked(monkeypatch, n_items, chunks, k)
    out = str(tmp_path / 'shard')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker, args=(world, rdzv, out, pipelined, micro, bypass), nprocs=world, join=True)
    _check(world, [torch.load(f'{out}.{r}', weights_only=False) for r in range(world)], micro)


@pytest.mark.parametrize('pipelined,micro,bypass,chunks', [(True, 1, False, 4), (True, 2, False, 2), (True, 1, True, 4)])
def test_sharded_step_through_rccl_world1_chunked(tmp_path, monkeypatch, pipelined, micro, bypass, chunks):
    """The chunked call sequence on a real 1-rank RCCL communicator: asynchronous all-to-all(v) per chunk, the training stream waiting
    chunk by chunk."""
    _chunked(monkeypatch, 9000, chunks)
    out = str(tmp_path / 'rccl')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker_rccl, args=(rdzv, out, pipelined, micro, bypass), nprocs=1, join=True)
    _check(1, [torch.load(f'{out}.0', weights_only=False)], micro)


# ---- the row layout behind the reference's own entry point: CDAE.fit() under a process group (recommender_abc.py:97-98) ------------------
def _rows_frame():
    rng = np.random.default_rng(21)
    n_users, n_items = 240, 10000                       # 2 * items_per_rank > 8192: room for two exchange chunks at world 2
    user = np.concatenate([rng.integers(0, n_users, size=n_items), rng.integers(0, n_users, size=14000)])
    item = np.concatenate([rng.permutation(n_items), rng.integers(0, 300, size=14000)])        # every item once + a hot set
    return {'user': user + 1, 'item': item + 1, 'interaction': rng.integers(1, 6, size=len(user))}


def _fit_rows_worker(rank, world, rdzv, out, chunks):
    dist.init_process_group('gloo', init_method=rdzv, rank=rank, world_size=world)
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    ds = InteractionDataset.read_df(_rows_frame(), verbose=False)
    model = CDAE(hidden_factors=18, mode='sampled', device_sampler=True, seed=3, verbose=False, layout='rows', exchange_chunks=chunks)
    model.fit(ds, epochs=10, batch_size=256, learning_rate=0.05, reg_rate=1e-3, neg_ratio=5)
    torch.cuda.synchronize()
    frame = _rows_frame()
    torch.save({'params': model._engine.get_params(), 'pred': float(model.predict(frame['user'][0], frame['item'][1])),
                'rank': model.rank(frame['user'][0], list(frame['item'][:30]), n=5), 'chunks': chunks}, f'{out}.{rank}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('chunks', [1, 2])
@retry_infra
def test_public_fit_row_layout_equals_a_single_gpu_replay_of_the_same_buckets(tmp_path, chunks):
    """CDAE(mode='sampled', device_sampler=True, layout='rows').fit() as two processes of one job — dist.ShardedCdae + ShardedPipeline
    behind the reference's entry point — then every rank holds the whole model.  SURVEY §8e's parity: equality with a single-process run
    that uses the SAME bucketing — every step's global batch = rank 0's draw of its users + rank 1's draw of its users (replayed here
    with the ranks' seeds and corruption masks) through the plain single-GPU step."""
    from oracle import cdae_oracle as co
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    from drecpy_amd.engine import CdaeEngine
    world, Kf, Bf, steps, seed = 2, 18, 256, 10, 3
    out = str(tmp_path / 'fitrows')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_fit_rows_worker, args=(world, rdzv, out, chunks), nprocs=world, join=True)
    ds = InteractionDataset.read_df(_rows_frame(), verbose=False)
    ds.assign_internal_ids()
    ip, idx = ds.positives_csr(1e-3)
    ip = np.asarray(ip, np.int64)
    Uf, Nf = len(ip) - 1, int(ds.count_unique('iid'))
    assert Nf == 10000
    rng = np.random.default_rng(seed)

    def glorot(shape):
        fi, fo = (shape[0], shape[0]) if len(shape) == 1 else (shape[0], shape[1])
        lim = np.sqrt(6.0 / (fi + fo))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)
    eng = CdaeEngine(Uf, Nf, Kf)
    eng.set_params(W=glorot((Nf, Kf)), W_=glorot((Kf, Nf)), V=glorot((Uf, Kf)), b=glorot((Kf,)), b_=glorot((Nf,)))
    eng.set_history(ip, idx)
    eng.init_optimizer('adagrad', 0.05, 1e-3)
    locals_ = []
    for r in range(world):
        lo, hi = Uf * r // world, Uf * (r + 1) // world
        e = CdaeEngine(hi - lo, Nf, 4)
        e.set_history(ip[lo:hi + 1] - ip[lo], idx[ip[lo]:ip[hi]], with_transpose=False)
        locals_.append((e, lo))
    thr = co.q_threshold(0.2)
    for s in range(steps):
        us, is_, ys, keeps = [], [], [], []
        for r, (e, lo) in enumerate(locals_):
            ss, sm = CDAE.row_seeds(seed, r, world)
            uid, iid, y, ko = [t.cpu().numpy() for t in e.sample_device(Bf, 5, ss(s), n_items=Nf)]
            deg = np.diff(ko)
            rows = np.repeat(np.arange(Bf), deg)
            j = np.arange(int(ko[-1])) - np.repeat(ko[:-1], deg)
            keeps.append((co.drx_hash_u32(sm(s), rows, j) >= thr).astype(np.uint8))
            us.append(uid + lo); is_.append(iid); ys.append(y)
        uid, iid, y, keep = np.concatenate(us), np.concatenate(is_), np.concatenate(ys), np.concatenate(keeps)
        ko = np.zeros(len(uid) + 1, np.int32)
        ko[1:] = np.cumsum(ip[uid + 1] - ip[uid])
        assert ko[-1] == len(keep)
        bt, alive = eng.make_batch(uid, iid, y, keep_off=ko, keep=keep, q=0.2)
        eng.step_sparse(s, bt, 'bce')
    torch.cuda.synchronize()
    want = eng.get_params()
    res = [torch.load(f'{out}.{r}', weights_only=False) for r in range(world)]
    for got in res:
        for k in want:
            np.testing.assert_allclose(got['params'][k], want[k], rtol=0, atol=3e-6, err_msg=k)
    assert res[0]['rank'] == res[1]['rank'] and abs(res[0]['pred'] - res[1]['pred']) < 1e-7


@pytest.mark.parametrize('phases', [True, False])
@pytest.mark.parametrize('pipelined,micro,bypass,chunks', [(True, 1, False, 1), (True, 1, False, 4), (True, 2, False, 2), (False, 1, False, 2), (True, 1, True, 4),
                                                           (True, 1, True, 2), (False, 1, True, 1)])
def test_sharded_step_through_the_librarys_own_communicator(monkeypatch, pipelined, micro, bypass, chunks, phases):
    """transport='rccl' (csrc/drx_comm.hip): count / key / row / gradient exchanges as ncclGroups of send / recv pairs on the library's own
    1-rank communicator and stream, ordered with torch's streams by events and tickets — the call sequence of the N-rank step.
    phases: the exchanges of a step issued by the library (drx_shard_phase_*: keys, rows, local, tail — the split sizes read from the count
    exchange's pinned mailbox) or call by call from dist.py; steps of several micro-batches always take the second form."""
    _chunked(monkeypatch, 9000, chunks)
    res = _run_rank(0, 1, False, force=True, pipelined=pipelined, micro=micro, bypass=bypass, transport='rccl', phases=phases)
    _check(1, [res], micro)


def test_the_librarys_phases_equal_the_call_by_call_step_bit_for_bit(monkeypatch):
    """same kernels, same buffers' geometry, same order: parameters after the pipelined steps are identical, not merely close"""
    _chunked(monkeypatch, 9000, 2)
    a = _run_rank(0, 1, False, force=True, pipelined=True, bypass=False, transport='rccl', phases=True)
    b = _run_rank(0, 1, False, force=True, pipelined=True, bypass=False, transport='rccl', phases=False)
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k]), k
    assert a[1] == b[1]
