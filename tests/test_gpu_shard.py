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


def _run_rank(rank, world, staged, force=False, pipelined=False, micro=1, bypass=True):
    from drecpy_amd.dist import ShardedCdae
    p, indptr, indices, batches = _problem(world)
    lo, hi = U * rank // world, U * (rank + 1) // world
    lip = indptr[lo:hi + 1] - indptr[lo]
    lidx = indices[indptr[lo]:indptr[hi]]
    m = ShardedCdae(U, N, K, rank, world, 'cuda:0', lip, lidx, q=Q, cpu_staging=staged, force_collectives=force, self_bypass=bypass,
                    chunks=CHUNKS)
    assert m.chunks == CHUNKS, (m.chunks, CHUNKS)
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
                                                                           (True, 2, 2, True, 2, 9000, 50), (True, 1, 2, False, 4, 17000, 128)])
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
