"""World-size-2 tests of the row-sharded step (drecpy_amd/dist.py) over gloo on CPU: two processes, each holding half of
the users and half of the item rows, exchange distinct rows / gradient rows by all-to-all; the result must equal the
single-process oracle step on the concatenated batch (SURVEY.md §8e parity caveat: same bucketing)."""
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
from helpers import new_rendezvous  # noqa: E402

U, N, K, B, STEPS, Q = 40, 57, 6, 24, 3, 0.2


def _problem(world=2, n_items=N):
    """n_items > N: the same 57 "live" items spread over a larger id range by a fixed permutation — the exchange CHUNKS cut an owner's
    key range into pieces of at least 8192 keys, so a chunked exchange needs 2 * items_per_rank > 8192 * chunks / 2 to have more than
    one chunk, and live items in all of them."""
    from oracle import cdae_oracle as co
    from helpers import synth_history
    rng = np.random.default_rng(11)
    p = co.init_params(rng, U, n_items, K, np.float64)
    indptr, indices = synth_history(rng, U, N, 7, zipf=1.1)
    spread = np.sort(np.random.default_rng(5).choice(n_items, size=N, replace=False)) if n_items > N else np.arange(N)
    spread = spread[np.random.default_rng(6).permutation(N)] if n_items > N else spread
    # (columns of a history row stay ascending: re-sort every row after the spread)
    ind2 = spread[indices].astype(np.int32)
    for u in range(U):
        ind2[indptr[u]:indptr[u + 1]].sort()
    batches = []
    for s in range(STEPS):
        per_rank = []
        for r in range(world):
            lo, hi = U * r // world, U * (r + 1) // world
            per_rank.append((rng.integers(lo, hi, size=B), spread[rng.integers(0, N, size=B)], (rng.random(B) < 0.3).astype(np.float64),
                             1000 + 17 * s + r))
        batches.append(per_rank)
    return p, indptr, ind2, batches


def _worker(rank, world, rdzv, out, pipelined=False, micro=1, bypass=True, chunks=1, n_items=N):
    dist.init_process_group('gloo', init_method=rdzv, rank=rank, world_size=world)
    from drecpy_amd.dist import ShardedCdae
    from dist_ops_numpy import NumpyShardOps, np_batch
    p, indptr, indices, batches = _problem(world, n_items)
    lo, hi = U * rank // world, U * (rank + 1) // world
    lip = indptr[lo:hi + 1] - indptr[lo]
    lidx = indices[indptr[lo]:indptr[hi]]
    ops = NumpyShardOps(hi - lo, n_items, K, rank, world, lip, lidx, 0.05, 1e-3, self_bypass=bypass, chunks=chunks)
    assert ops.chunks == chunks, (ops.chunks, chunks)
    m = ShardedCdae(U, n_items, K, rank, world, 'cpu', lip, lidx, ops=ops, q=Q, chunks=chunks)
    assert m.chunks == chunks
    m.set_params_global(**p)
    losses = []

    def batch_of(s):
        uid, iid, y, seed = batches[s][rank]
        if micro == 1:
            return np_batch(uid - lo, iid, y, lip, Q, mask_seed=seed)
        parts = [np.flatnonzero(uid % micro == m) for m in range(micro)]          # micro-batches with disjoint users
        return [np_batch(uid[ix] - lo, iid[ix], y[ix], lip, Q, mask_seed=seed + 100 * m) for m, ix in enumerate(parts)]
    if pipelined:                      # run-ahead stage order of the multi-GPU bench (keys s+1, counts s+2, step s)
        from drecpy_amd.dist import ShardedPipeline
        pipe = ShardedPipeline(m, batch_of, STEPS)
        losses = [pipe.run_step(want_loss=True) for _ in range(STEPS)]
    else:
        for s in range(STEPS):
            losses.append(m.step(s, batch_of(s), want_loss=True))
    got = ops.get_params()
    full = m.gather_params_global()                     # every rank the whole model (tensor all-gathers of the padded shards)
    torch.save({'params': {k: np.asarray(v) for k, v in got.items()}, 'losses': losses, 'full': {k: np.asarray(v) for k, v in full.items()}},
               f'{out}.{rank}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('pipelined,micro,world,bypass,chunks,n_items', [
    (False, 1, 2, True, 1, N), (True, 1, 2, True, 1, N), (True, 2, 2, True, 1, N), (False, 3, 2, False, 1, N),
    (True, 1, 3, True, 1, N), (True, 2, 4, True, 1, N), (True, 1, 2, False, 1, N), (True, 2, 3, False, 1, N),
    # the chunked schedule (r06): every exchange in `chunks` all-to-alls, the owner apply of chunk c followed by the gather + row exchange
    # of the NEXT step's chunk c (pipelined) or every step fetching its own rows (inline)
    (True, 1, 2, True, 2, 9000), (True, 1, 2, False, 4, 17000), (False, 1, 3, True, 2, 13000), (True, 2, 2, True, 2, 9000),
    (True, 1, 4, True, 4, 34000), (True, 2, 3, False, 2, 13000),
    (True, 1, 8, True, 2, 70000)])                                             # (the world of BASELINE configuration 4)
def test_sharded_step_equals_single_process_oracle(tmp_path, pipelined, micro, world, bypass, chunks, n_items):
    """micro > 1: every rank's batch is split into micro-batches with disjoint users whose exchanges overlap each other's
    compute; the step must still equal the single-process step on the concatenated batch.  bypass: a rank's own rows never pass
    through the collectives (split size 0 for itself); off: every row travels.  chunks > 1: the exchanges cut into key-range chunks and
    pipelined with the owner apply / the next step's gather (drecpy_amd/dist.py module docstring)."""
    from oracle import cdae_oracle as co
    N = n_items
    out = str(tmp_path / 'shard')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker, args=(world, rdzv, out, pipelined, micro, bypass, chunks, n_items), nprocs=world, join=True)
    p, indptr, indices, batches = _problem(world, n_items)
    st = co.sparse_state(p, 'adagrad')
    want_losses = []
    for s in range(STEPS):
        uid = np.concatenate([batches[s][r][0] for r in range(world)])
        iid = np.concatenate([batches[s][r][1] for r in range(world)])
        y = np.concatenate([batches[s][r][2] for r in range(world)])
        kept = [None] * len(uid)
        base = 0
        for r in range(world):
            u_r, _, _, seed = batches[s][r]
            for m in range(micro):                      # the corruption mask of a sample is keyed by its micro-batch position
                for b, j in enumerate(np.flatnonzero(u_r % micro == m)):
                    row = indices[indptr[u_r[j]]:indptr[u_r[j] + 1]]
                    kf = co.drx_hash_u32(seed + (100 * m if micro > 1 else 0), np.full(len(row), b), np.arange(len(row))) >= co.q_threshold(Q)
                    kept[base + j] = row[kf].tolist()
            base += len(u_r)
        lval, _ = co.sparse_step(p, st, s, uid, iid, y, kept, float(np.float32(Q)), 0.05, 1e-3, 'bce', 'adagrad')
        want_losses.append(lval)
    res = [torch.load(f'{out}.{r}', weights_only=False) for r in range(world)]
    ipr = (N + world - 1) // world
    for r in range(world):
        g = res[r]['params']
        lo, hi = r * ipr, min(N, (r + 1) * ipr)
        np.testing.assert_allclose(g['W'][:hi - lo], p['W'][lo:hi], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g['W_'][:, :hi - lo], p['W_'][:, lo:hi], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g['b_'][:hi - lo], p['b_'][lo:hi], rtol=1e-9, atol=1e-12)
        ulo, uhi = U * r // world, U * (r + 1) // world
        np.testing.assert_allclose(g['V'], p['V'][ulo:uhi], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g['b'], p['b'], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res[r]['losses'], want_losses, rtol=1e-9)
        full = res[r]['full']                           # gather_params_global: the reference-orientation model, identical on every rank
        assert full['W'].shape == p['W'].shape and full['W_'].shape == p['W_'].shape and full['V'].shape == p['V'].shape
        for k in ('W', 'W_', 'V', 'b', 'b_'):
            np.testing.assert_allclose(full[k], p[k], rtol=1e-9, atol=1e-12, err_msg=f'gathered {k}')


# ---- column-sharded layout (dist.ColumnShardedCdae) over gloo on CPU -------------------------------------------------------------
class _NumpyColumnEngine:
    """The two per-rank halves of the column-sharded step, stated with the oracle's arithmetic on a column slice: the forward
    half yields the partial dot products, the step half is oracle.sparse_step fed the all-reduced ones."""

    def __init__(self, p_slice, indptr, indices, lr, reg):
        from oracle import cdae_oracle as co
        self.co, self.p, self.indptr, self.indices, self.lr, self.reg = co, p_slice, indptr, indices, lr, reg
        self.st = co.sparse_state(p_slice, 'adagrad')

    def _kept(self, bt):
        co = self.co
        out = []
        for b, u in enumerate(bt['uid']):
            row = self.indices[self.indptr[u]:self.indptr[u + 1]]
            kf = co.drx_hash_u32(bt['seed'], np.full(len(row), b), np.arange(len(row))) >= co.q_threshold(Q)
            out.append(row[kf].tolist())
        return out

    def kshard_forward(self, bt):
        co, p = self.co, self.p
        s = 1.0 / (1.0 - float(np.float32(Q)))
        z1 = np.stack([p['W'][k].sum(axis=0) * s if len(k) else np.zeros(p['W'].shape[1]) for k in self._kept(bt)]) + p['V'][bt['uid']] + p['b']
        h = co.sigmoid(z1)
        return h, torch.from_numpy((h * p['W_'][:, bt['iid']].T).sum(axis=1))

    def step_sparse(self, step, bt, loss, want_loss=False, events=None, prepared=None, kshard=None):
        total = kshard[1].numpy()
        lval, _ = self.co.sparse_step(self.p, self.st, step, bt['uid'], bt['iid'], bt['y'], self._kept(bt), float(np.float32(Q)), self.lr, self.reg,
                                      'bce', 'adagrad', dot_reduce=lambda d: total)
        return [lval]


def _column_worker(rank, world, rdzv, out):
    dist.init_process_group('gloo', init_method=rdzv, rank=rank, world_size=world)
    from drecpy_amd.dist import ColumnShardedCdae
    p, indptr, indices, batches = _problem(1)
    m = ColumnShardedCdae(U, N, K, rank, world, 'cpu', indptr, indices, q=Q, engine=object())
    lo, hi = m.k_lo, m.k_hi
    m.engine = _NumpyColumnEngine({'W': p['W'][:, lo:hi].copy(), 'W_': p['W_'][lo:hi, :].copy(), 'V': p['V'][:, lo:hi].copy(),
                                   'b': p['b'][lo:hi].copy(), 'b_': p['b_'].copy()}, indptr, indices, 0.05, 1e-3)
    losses = []
    for s in range(STEPS):
        uid, iid, y, seed = batches[s][0]                       # the SAME global batch on every rank
        losses.append(m.step(s, {'uid': uid, 'iid': iid, 'y': y, 'seed': seed}, want_loss=True))
    torch.save({'params': m.engine.p, 'losses': losses, 'cols': (lo, hi)}, f'{out}.{rank}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_column_sharded_step_equals_single_process_oracle(tmp_path, world):
    """Every rank holds all rows x its columns and the same batch; the only exchange is the all-reduce of the partial dot
    products (dist.ColumnShardedCdae.step) — the result must equal the single-process step on all K columns (K = 6: 3 + 3, 2 + 2 + 2)."""
    from oracle import cdae_oracle as co
    out = str(tmp_path / 'cols')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_column_worker, args=(world, rdzv, out), nprocs=world, join=True)
    p, indptr, indices, batches = _problem(1)
    st = co.sparse_state(p, 'adagrad')
    want = []
    for s in range(STEPS):
        uid, iid, y, seed = batches[s][0]
        kept = []
        for b, u in enumerate(uid):
            row = indices[indptr[u]:indptr[u + 1]]
            kept.append(row[co.drx_hash_u32(seed, np.full(len(row), b), np.arange(len(row))) >= co.q_threshold(Q)].tolist())
        lval, _ = co.sparse_step(p, st, s, uid, iid, y, kept, float(np.float32(Q)), 0.05, 1e-3, 'bce', 'adagrad')
        want.append(lval)
    for r in range(world):
        res = torch.load(f'{out}.{r}', weights_only=False)
        lo, hi = res['cols']
        g = res['params']
        tol = dict(rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g['W'], p['W'][:, lo:hi], **tol)
        np.testing.assert_allclose(g['W_'], p['W_'][lo:hi, :], **tol)
        np.testing.assert_allclose(g['V'], p['V'][:, lo:hi], **tol)
        np.testing.assert_allclose(g['b'], p['b'][lo:hi], **tol)
        np.testing.assert_allclose(g['b_'], p['b_'], **tol)
        np.testing.assert_allclose(res['losses'], want, rtol=1e-9)


# ---- who builds the touch list of a step (ColumnShardedCdae prepare='turns') over gloo on CPU ----------------------------------
class _FakePrepEngine:
    """Stands in for CdaeEngine's preparation calls: a 'prepared list' is a byte buffer whose result part is a function of the
    batch and of WHO built it, so the test can see whose list every rank ends up holding."""
    RESULT, TOTAL = 96, 160

    def __init__(self, rank):
        self.rank, self.device = rank, torch.device('cpu')

    def prep_buffer(self, bt, out=None):
        return out if out is not None and out.numel() >= self.TOTAL else torch.full((self.TOTAL,), 255, dtype=torch.uint8)

    def prep_result_bytes(self, bt):
        return self.RESULT

    def prepare_sparse(self, bt, out=None):
        out = self.prep_buffer(bt, out)
        out[:self.RESULT] = torch.arange(self.RESULT, dtype=torch.uint8) + bt['id']        # the list of this batch ...
        out[self.RESULT:] = 100 + self.rank                                                # ... and the builder's private scratch
        return out


def _turns_worker(rank, world, rdzv, out):
    dist.init_process_group('gloo', init_method=rdzv, rank=rank, world_size=world)
    from drecpy_amd.dist import ColumnShardedCdae
    p, indptr, indices, _ = _problem(1)
    m = ColumnShardedCdae(U, N, K, rank, world, 'cpu', indptr, indices, q=Q, engine=_FakePrepEngine(rank), prepare='turns')
    got, bufs = [], [None, None, None]
    for s in range(7):
        bt = {'id': 3 * s + 1}
        if bufs[s % 3] is not None:
            bufs[s % 3].fill_(255)                               # (forget who built the list this buffer held before)
        buf = m.build_in_turns(s, bt, bufs[s % 3])               # rank s % world builds, the others only hold a buffer
        built_here = bool((buf[_FakePrepEngine.RESULT:] == 100 + rank).all())
        m.deliver_in_turns(s, bt, buf)
        bufs[s % 3] = buf
        got.append((built_here, buf[:_FakePrepEngine.RESULT].clone()))
    torch.save(got, f'{out}.{rank}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_lists_built_in_turns_reach_every_rank(tmp_path, world):
    """prepare='turns': the list of step s is built by rank s % world only and every rank ends up with exactly its result bytes
    (the broadcast runs on a communicator of its own; the builder's scratch part never travels)."""
    out = str(tmp_path / 'turns')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_turns_worker, args=(world, rdzv, out), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(f'{out}.{r}', weights_only=False)
        for s, (built_here, res) in enumerate(got):
            assert built_here == (s % world == r)
            assert torch.equal(res, torch.arange(_FakePrepEngine.RESULT, dtype=torch.uint8) + (3 * s + 1))


def test_wire_keys_are_unit_major_and_invertible():
    """include/drx.h WIRE keys: unit v = chunk * world + owner holds one (chunk, owner) key range; keys of a unit are contiguous, the
    map (item, W2T?) -> key is injective and wire_local inverts it; chunks are lowered until a unit spans >= 8192 keys."""
    from drecpy_amd.dist import items_per_rank, wire_chunks, wire_key, wire_local, wire_shift
    for n_items, world, chunks in ((57, 2, 4), (9000, 2, 2), (17000, 2, 4), (34000, 4, 4), (1_000_000, 8, 8)):
        ipr = items_per_rank(n_items, world)
        C = wire_chunks(ipr, chunks)
        cs = wire_shift(ipr) - (C.bit_length() - 1)
        assert cs >= 13 and (C == chunks or wire_shift(ipr) - (chunks.bit_length() - 1) < 13)
        rng = np.random.default_rng(n_items)
        items = np.unique(np.concatenate([rng.integers(0, n_items, size=400), [0, n_items - 1, ipr - 1, min(ipr, n_items - 1)]]))
        seen = set()
        for n in items.tolist():
            for is_out in (False, True):
                k = wire_key(n, ipr, is_out, world, C)
                assert k not in seen and k < (world << wire_shift(ipr))
                seen.add(k)
                o, c, out_, li = wire_local(k, ipr, world, C)
                assert (o, out_, li) == (n // ipr, is_out, n - (n // ipr) * ipr)
                assert (k >> cs) == c * world + o and c == (2 * li + int(is_out)) >> cs
    assert wire_chunks(items_per_rank(57, 2), 4) == 1 and wire_chunks(items_per_rank(1_000_000, 8), 8) == 8


def test_exchange_sizes_of_the_library_equal_the_host_statement():
    """drx_shard_exchange_sizes (csrc/drx_shard_phase.cpp: the geometry the C-issued phases of a step use) against dist.chunk_floats /
    the self-bypass rule stated in Python: floats of the requester's buffers (own units included, at the end), floats of the owner's
    (nothing from the rank itself with the bypass), keys received and sent.  Host arithmetic only: no GPU call."""
    import ctypes as C
    from drecpy_amd import _lib
    from drecpy_amd.dist import chunk_floats, items_per_rank, wire_chunks
    L = _lib.lib()
    rng = np.random.default_rng(5)
    for world, rank, n_items, chunks, bypass, ld in ((1, 0, 9000, 2, True, 128), (4, 2, 40000, 4, True, 64), (4, 1, 40000, 4, False, 128),
                                                     (8, 7, 1_000_000, 2, True, 128), (3, 0, 500, 4, True, 32)):
        ipr = items_per_rank(n_items, world)
        sh = _lib.Shard(world, rank, n_items, ipr, 100, _lib.SHARD_SELF_BYPASS if bypass else 0, wire_chunks(ipr, chunks))
        Cn = int(L.drx_shard_chunks(C.byref(sh)))
        P = _lib.CdaeParams(100, ipr, ld, ld, None, None, None, None, None)
        send = rng.integers(1, 3000, size=world * Cn).astype(np.int64)
        recv = rng.integers(1, 3000, size=world * Cn).astype(np.int64)
        out = (C.c_int64 * 4)()
        assert L.drx_shard_exchange_sizes(C.byref(P), C.byref(sh), send.ctypes.data, recv.ctypes.data, out) == 0
        want_req = sum(chunk_floats(send.tolist(), ld))
        want_own = sum(f for i, f in enumerate(chunk_floats(recv.tolist(), ld)) if not (bypass and i // Cn == rank))
        assert list(out) == [max(32, want_req), max(32, want_own), max(32, int(recv.sum())), int(send.sum())], (world, rank, list(out))
        send[0] = 0                                    # (a unit without its sentinel: rejected)
        assert L.drx_shard_exchange_sizes(C.byref(P), C.byref(sh), send.ctypes.data, recv.ctypes.data, out) != 0


def test_the_phases_geometry_equals_the_call_by_call_steps_at_every_world():
    """drx_shard_phase_layout (what drx_shard_phase_keys / _rows / _tail hand the communicator: csrc/drx_shard_phase.cpp) against the
    arithmetic of dist.ShardedCdae's call-by-call step — the form the gloo tests above hold against the oracle at world 2 / 3 / 4 / 8:
    per exchange chunk and peer the byte offset and size of the key, row and gradient exchanges in both directions, the chunk's place in
    the owner's buffers, the rank's own piece behind everything that travels.  On one GPU the phases only ever run at world 1; this is
    their world > 1.  Host arithmetic only."""
    import ctypes as C
    from drecpy_amd import _lib
    from drecpy_amd.dist import chunk_floats, items_per_rank, wire_chunks
    L = _lib.lib()
    rng = np.random.default_rng(11)
    for world, rank, n_items, chunks, bypass, ld in ((2, 0, 20000, 2, True, 128), (2, 1, 20000, 2, False, 128), (3, 1, 30000, 2, True, 64),
                                                     (4, 3, 80000, 4, True, 128), (8, 5, 1_000_000, 2, True, 128), (8, 0, 1_000_000, 1, False, 128),
                                                     (8, 7, 1_000_000, 4, True, 52)):
        ipr = items_per_rank(n_items, world)
        sh = _lib.Shard(world, rank, n_items, ipr, 100, _lib.SHARD_SELF_BYPASS if bypass else 0, wire_chunks(ipr, chunks))
        Cn = int(L.drx_shard_chunks(C.byref(sh)))
        assert Cn == chunks
        P = _lib.CdaeParams(100, ipr, ld, ld, None, None, None, None, None)
        send = rng.integers(1, 4000, size=world * Cn).astype(np.int64)          # [owner * Cn + chunk]
        recv = rng.integers(1, 4000, size=world * Cn).astype(np.int64)          # [source * Cn + chunk]
        sc = [[int(send[o * Cn + c]) for o in range(world)] for c in range(Cn)]  # as ShardedCdae._split_counts
        rc = [[int(recv[s * Cn + c]) for s in range(world)] for c in range(Cn)]

        def xs(counts):                                                          # HipShardOps.xsplits
            f = chunk_floats(counts, ld)
            return [0 if (bypass and i == rank) else x for i, x in enumerate(f)]
        cache_off = np.concatenate([[0], np.cumsum([sum(xs(sc[c])) for c in range(Cn)])])     # ShardedCdae._fetch_chunk
        own_floats = np.concatenate([[0], np.cumsum([sum(xs(rc[c])) for c in range(Cn)])])    # (the per-chunk tensors, laid end to end)
        for c in range(Cn):
            out = (C.c_int64 * (12 * world + 3))()
            assert L.drx_shard_phase_layout(C.byref(P), C.byref(sh), send.ctypes.data, recv.ctypes.data, c, out) == 0
            o = np.array(list(out)).astype(np.int64)
            blk = lambda b: o[b * world:(b + 1) * world]
            k0 = sum(sum(sc[cc]) for cc in range(c))
            r0 = sum(sum(rc[cc]) for cc in range(c))
            cum = lambda v: np.concatenate([[0], np.cumsum(v)[:-1]])
            # keys: uniq[k0:] split by sc[c]  ->  req, chunk after chunk, split by rc[c]
            assert np.array_equal(blk(0), 4 * (k0 + cum(sc[c]))) and np.array_equal(blk(1), 4 * np.array(sc[c]))
            assert np.array_equal(blk(2), 4 * (r0 + cum(rc[c]))) and np.array_equal(blk(3), 4 * np.array(rc[c]))
            # rows: the owner's gather output of the chunk split by xsplits(rc[c])  ->  cache[cache_off[c]:] split by xsplits(sc[c])
            assert np.array_equal(blk(4), 4 * (own_floats[c] + cum(xs(rc[c])))) and np.array_equal(blk(5), 4 * np.array(xs(rc[c])))
            assert np.array_equal(blk(6), 4 * (cache_off[c] + cum(xs(sc[c])))) and np.array_equal(blk(7), 4 * np.array(xs(sc[c])))
            # gradient rows: gsend[cache_off[c]:] split by xsplits(sc[c])  ->  the chunk's receive buffer split by xsplits(rc[c])
            assert np.array_equal(blk(8), blk(6)) and np.array_equal(blk(9), blk(7))
            assert np.array_equal(blk(10), blk(4)) and np.array_equal(blk(11), blk(5))
            assert o[12 * world] == r0 and o[12 * world + 1] == own_floats[c]
            own = cache_off[Cn] + sum(chunk_floats([sc[cc][rank]], ld)[0] for cc in range(c)) if bypass else 0     # ShardedCdae.step
            assert o[12 * world + 2] == own, (world, rank, c, o[12 * world + 2], own)
