"""GPU parity of the COLUMN-sharded step (dist.ColumnShardedCdae: every rank all rows x its columns, same global batch, one
all-reduce of the partial dot products): world 1 in-process, world 2 / 3 as processes sharing the one GPU of the box (gloo,
host-staged all-reduce), and through a 1-rank RCCL communicator — against the single-process oracle step on all K columns."""
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
U, N, B, STEPS, Q = 300, 411, 700, 4, 0.2


def _problem(K):
    from oracle import cdae_oracle as co
    from helpers import synth_history
    rng = np.random.default_rng(5)
    p = co.init_params(rng, U, N, K, np.float32)
    indptr, indices = synth_history(rng, U, N, 12, zipf=1.2)
    batches = [(rng.integers(0, U, size=B), rng.integers(0, N, size=B), (rng.random(B) < 0.3).astype(np.float32), 500 + 31 * s)
               for s in range(STEPS)]
    return p, indptr, indices, batches


def _oracle(K, opt):
    from oracle import cdae_oracle as co
    p, indptr, indices, batches = _problem(K)
    p = {k: v.astype(np.float64) for k, v in p.items()}
    st = co.sparse_state(p, opt)
    lr = 0.05 if opt == 'adagrad' else 1e-3
    losses = []
    for s, (uid, iid, y, seed) in enumerate(batches):
        kept = []
        for b, u in enumerate(uid):
            row = indices[indptr[u]:indptr[u + 1]]
            kf = co.drx_hash_u32(seed, np.full(len(row), b), np.arange(len(row))) >= co.q_threshold(Q)
            kept.append(row[kf].tolist())
        lval, _ = co.sparse_step(p, st, s, uid, iid, y, kept, float(np.float32(Q)), lr, 1e-3, 'bce', opt)
        losses.append(lval)
    return p, losses


def _run_rank(rank, world, K, opt, staged, force=False, prepared=True, parts=False, turns=False):
    from drecpy_amd.dist import ColumnShardedCdae
    p, indptr, indices, batches = _problem(K)
    m = ColumnShardedCdae(U, N, K, rank, world, 'cuda:0', indptr, indices, q=Q, optimizer=opt, lr=0.05 if opt == 'adagrad' else 1e-3,
                          cpu_staging=staged, force_collectives=force)
    m.set_params_global(**p)
    losses = []
    for s, (uid, iid, y, seed) in enumerate(batches):                      # the SAME batch on every rank
        bt, alive = m.engine.make_batch(uid, iid, y, q=Q, mask_seed=seed)
        if parts:                                                           # touch list built in parts and gathered
            prep = m.prepare(s, bt)
        elif turns:                                                         # built by rank s % world, broadcast
            prep = m.prepare_in_turns(s, bt)
        else:
            prep = m.engine.prepare_sparse(bt) if (prepared and s % 2) else None
        losses.append(m.step(s, bt, prepared=prep, want_loss=True))
    torch.cuda.synchronize()
    if world > 1:                                                           # every rank can assemble the whole model
        full = m.gather_params_global()
        assert full['W'].shape == (N, K) and full['W_'].shape == (K, N) and full['V'].shape == (U, K) and full['b'].shape == (K,)
        np.testing.assert_array_equal(full['W'][:, m.k_lo:m.k_hi], m.get_params()['W'])
    return m.get_params(), losses, (m.k_lo, m.k_hi)


def _check(K, opt, results):
    p, want_losses = _oracle(K, opt)
    tol = dict(rtol=0, atol=3e-6)
    for g, losses, (lo, hi) in results:
        np.testing.assert_allclose(g['W'], p['W'][:, lo:hi], **tol)
        np.testing.assert_allclose(g['W_'], p['W_'][lo:hi, :], **tol)
        np.testing.assert_allclose(g['V'], p['V'][:, lo:hi], **tol)
        np.testing.assert_allclose(g['b'], p['b'][lo:hi], **tol)
        np.testing.assert_allclose(g['b_'], p['b_'], **tol)                 # replicated, updated identically everywhere
        np.testing.assert_allclose(losses, want_losses, rtol=1e-5)


@pytest.mark.parametrize('K,opt', [(50, 'adagrad'), (128, 'adam')])
def test_column_sharded_world1_matches_oracle(K, opt):
    _check(K, opt, [_run_rank(0, 1, K, opt, False)])


def _worker(rank, world, rdzv, out, K, opt, parts=False, turns=False):
    dist.init_process_group('gloo', init_method=rdzv, rank=rank, world_size=world)
    res = _run_rank(rank, world, K, opt, True, parts=parts, turns=turns)
    torch.save(res, f'{out}.{rank}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,K,opt', [(2, 128, 'adagrad'), (3, 50, 'adagrad'), (2, 64, 'adam')])
@retry_infra
def test_column_sharded_processes_on_one_gpu_match_oracle(tmp_path, world, K, opt):
    """world 3 with K = 50: 17 + 17 + 16 columns (uneven, and lane-group geometries differ from the unsharded one)."""
    out = str(tmp_path / 'ks')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker, args=(world, rdzv, out, K, opt), nprocs=world, join=True)
    _check(K, opt, [torch.load(f'{out}.{r}', weights_only=False) for r in range(world)])


def _worker_rccl(rank, rdzv, out, K, opt, parts=False, turns=False):
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method=rdzv, rank=0, world_size=1, device_id=torch.device('cuda', 0))
    res = _run_rank(0, 1, K, opt, False, force=True, parts=parts, turns=turns)
    torch.save(res, f'{out}.0')
    dist.barrier()
    dist.destroy_process_group()


def test_column_sharded_step_through_rccl_world1(tmp_path):
    out = str(tmp_path / 'ksr')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker_rccl, args=(rdzv, out, 128, 'adagrad'), nprocs=1, join=True)
    _check(128, 'adagrad', [torch.load(f'{out}.0', weights_only=False)])


@pytest.mark.parametrize('world,K,opt', [(2, 128, 'adagrad'), (3, 50, 'adam')])
@retry_infra
def test_touch_list_built_in_parts_across_processes(tmp_path, world, K, opt):
    """ColumnShardedCdae.prepare: every rank sorts the touches of the rows it owns, one all-gather, every rank assembles."""
    out = str(tmp_path / 'kp')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker, args=(world, rdzv, out, K, opt, True), nprocs=world, join=True)
    _check(K, opt, [torch.load(f'{out}.{r}', weights_only=False) for r in range(world)])


@pytest.mark.parametrize('world,K,opt', [(2, 128, 'adagrad'), (3, 50, 'adam')])
@retry_infra
def test_touch_list_built_in_turns_across_processes(tmp_path, world, K, opt):
    """ColumnShardedCdae.prepare_in_turns: rank s % world sorts the list of step s and broadcasts the leading
    drx_cdae_prep_result_bytes of its prepared buffer.  world 3, K = 50: the ranks' row widths differ (32, 32, 16 floats), so a
    list marked for sole touchers by one geometry is consumed by another."""
    out = str(tmp_path / 'kt')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker, args=(world, rdzv, out, K, opt, False, True), nprocs=world, join=True)
    _check(K, opt, [torch.load(f'{out}.{r}', weights_only=False) for r in range(world)])


def test_touch_list_built_in_turns_through_rccl_world1(tmp_path):
    out = str(tmp_path / 'ktr')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker_rccl, args=(rdzv, out, 128, 'adagrad', False, True), nprocs=1, join=True)
    _check(128, 'adagrad', [torch.load(f'{out}.0', weights_only=False)])


def test_touch_list_built_in_parts_through_rccl_world1(tmp_path):
    out = str(tmp_path / 'kpr')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_worker_rccl, args=(rdzv, out, 128, 'adagrad', True), nprocs=1, join=True)
    _check(128, 'adagrad', [torch.load(f'{out}.0', weights_only=False)])


@pytest.mark.parametrize('parts', [1, 2, 3, 8, 64])
def test_parts_of_a_touch_list_hold_every_touch_once(parts):
    """The exchanged format of include/drx.h (header | runs | samples), decoded on the host: the parts partition the touches
    of the batch by row id % parts, keys ascend inside a part and the samples of a key keep their order; the list
    assembled from them trains to the same result as the locally sorted one."""
    import ctypes as C
    from drecpy_amd import _lib
    from drecpy_amd.engine import CdaeEngine
    from oracle import cdae_oracle as co
    p, indptr, indices, batches = _problem(32)
    uid, iid, y, seed = batches[0]
    e = CdaeEngine(U, N, 32); e.set_params(**p); e.set_history(indptr, indices); e.init_optimizer('adagrad', 0.05, 1e-3)
    bt, alive = e.make_batch(uid, iid, y, q=Q, mask_seed=seed)
    want = []                                                               # (key, sample) of every touch
    for b, u in enumerate(uid):
        row = indices[indptr[u]:indptr[u + 1]]
        kf = co.drx_hash_u32(seed, np.full(len(row), b), np.arange(len(row))) >= co.q_threshold(Q)
        want += [(int(n), b) for n in row[kf]] + [(N + int(iid[b]), b), (2 * N + int(u), b)]
    lay = (C.c_size_t * 4)()
    assert _lib.lib().drx_cdae_prep_part_layout(C.byref(e._params), bt.B, bt.n_touch_slots, parts, lay) == 0
    runs_off, vals_off, rcap, cap = [int(v) for v in lay]
    got, blobs = [], []
    for r in range(parts):
        blob = e.prepare_part(bt, r, parts).clone()
        blobs.append(blob)
        raw = blob.cpu().numpy()
        n_touch, n_runs, overflow, _ = raw[:16].view(np.int32)
        assert overflow == 0 and n_touch <= cap and n_runs <= rcap
        runs = raw[runs_off:runs_off + 8 * n_runs].view(np.uint64)
        vals = raw[vals_off:vals_off + 4 * n_touch].view(np.uint32)
        keys, starts = (runs >> np.uint64(32)).astype(np.int64), (runs & np.uint64(0xFFFFFFFF)).astype(np.int64)
        assert np.all(np.diff(keys) > 0) and (n_runs == 0 or starts[0] == 0) and np.all(np.diff(starts) > 0)
        ends = np.append(starts[1:], n_touch)
        for k, a, z in zip(keys, starts, ends):
            row = k if k < N else k - N if k < 2 * N else k - 2 * N
            assert row % parts == r
            assert np.all(np.diff(vals[a:z].astype(np.int64)) >= 0)         # sample order kept inside a key
            got += [(int(k), int(v)) for v in vals[a:z]]
    assert sorted(got) == sorted(want)
    prep, overflow = e.prepare_assemble(bt, torch.cat(blobs), parts)
    e.step_sparse(0, bt, prepared=prep)
    e2 = CdaeEngine(U, N, 32); e2.set_params(**p); e2.set_history(indptr, indices); e2.init_optimizer('adagrad', 0.05, 1e-3)
    bt2, alive2 = e2.make_batch(uid, iid, y, q=Q, mask_seed=seed)
    e2.step_sparse(0, bt2, prepared=e2.prepare_sparse(bt2))
    torch.cuda.synchronize()
    assert int(overflow[0]) == 0
    for a, b in zip(e.tables(), e2.tables()):
        assert float((a - b).abs().max()) < 1e-6


def test_a_part_that_does_not_fit_is_reported():
    """All touches on one row of one part: 1.25 x the even share + 16384 cannot hold them, the flag must say so."""
    from drecpy_amd.engine import CdaeEngine
    rng = np.random.default_rng(1)
    Ul, Nl, Bl = 8, 16, 60000
    indptr = np.arange(Ul + 1, dtype=np.int64)
    indices = np.zeros(Ul, np.int64)
    e = CdaeEngine(Ul, Nl, 8); e.init_glorot(1); e.set_history(indptr, indices); e.init_optimizer('adagrad', 0.05, 1e-3)
    uid, iid, y = np.zeros(Bl, np.int64), np.zeros(Bl, np.int64), np.ones(Bl, np.float32)     # every touch on row 0 of W, W2T, V
    bt, alive = e.make_batch(uid, iid, y, q=0.0, mask_seed=0)
    blobs = [e.prepare_part(bt, r, 4).clone() for r in range(4)]
    prep, overflow = e.prepare_assemble(bt, torch.cat(blobs), 4)
    torch.cuda.synchronize()
    assert int(overflow[0]) == 1
    assert int(blobs[0].cpu().numpy()[:16].view(np.int32)[2]) == 1


@pytest.mark.parametrize('prepare', ['local', 'turns', 'parts'])
def test_column_sharded_pipeline_equals_stepping_inline(prepare):
    """ColumnShardedCdae.pipeline (device sampler two batches ahead, touch list one ahead) vs the same seeds stepped inline."""
    from drecpy_amd.dist import ColumnShardedCdae
    p, indptr, indices, _ = _problem(64)
    outs = []
    for piped in (True, False):
        m = ColumnShardedCdae(U, N, 64, 0, 1, 'cuda:0', indptr, indices, q=Q, prepare=prepare)
        m.set_params_global(**p)
        if piped:
            pipe = m.pipeline(512, 5, lambda s: 77 + s, lambda s: 1000 + s)
            for _ in range(6):
                pipe.run_step()
        else:
            for s in range(6):
                uid, iid, y, ko = m.engine.sample_device(512, 5, 77 + s)
                bt, alive = m.engine.make_batch(uid, iid, y, keep_off=ko, q=Q, mask_seed=1000 + s)
                m.step(s, bt)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in m.engine.tables()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _pipe_worker(rank, world, rdzv, out, prepare):
    dist.init_process_group('gloo', init_method=rdzv, rank=rank, world_size=world)
    from drecpy_amd.dist import ColumnShardedCdae
    p, indptr, indices, _ = _problem(64)
    m = ColumnShardedCdae(U, N, 64, rank, world, 'cuda:0', indptr, indices, q=Q, cpu_staging=True, prepare=prepare)
    m.set_params_global(**p)
    pipe = m.pipeline(512, 5, lambda s: 77 + s, lambda s: 1000 + s)
    for _ in range(9):
        pipe.run_step()
    torch.cuda.synchronize()
    torch.save((m.get_params(), (m.k_lo, m.k_hi)), f'{out}.{rank}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,prepare', [(2, 'turns'), (3, 'turns'), (2, 'parts')])
@retry_infra
def test_pipelines_of_several_processes_equal_the_single_process_run(tmp_path, world, prepare):
    """The run-ahead pipeline with lists built in turns (built three steps ahead by rank s % world, broadcast one step ahead)
    or in parts, as `world` processes sharing the GPU: every rank's columns equal those of one process stepping inline."""
    from drecpy_amd.dist import ColumnShardedCdae
    out = str(tmp_path / 'pp')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_pipe_worker, args=(world, rdzv, out, prepare), nprocs=world, join=True)
    p, indptr, indices, _ = _problem(64)
    m = ColumnShardedCdae(U, N, 64, 0, 1, 'cuda:0', indptr, indices, q=Q)
    m.set_params_global(**p)
    for s in range(9):
        uid, iid, y, ko = m.engine.sample_device(512, 5, 77 + s)
        bt, alive = m.engine.make_batch(uid, iid, y, keep_off=ko, q=Q, mask_seed=1000 + s)
        m.step(s, bt)
    torch.cuda.synchronize()
    want = m.get_params()
    for r in range(world):
        got, (lo, hi) = torch.load(f'{out}.{r}', weights_only=False)
        np.testing.assert_allclose(got['W'], want['W'][:, lo:hi], rtol=0, atol=2e-6)
        np.testing.assert_allclose(got['W_'], want['W_'][lo:hi, :], rtol=0, atol=2e-6)
        np.testing.assert_allclose(got['V'], want['V'][:, lo:hi], rtol=0, atol=2e-6)
        np.testing.assert_allclose(got['b_'], want['b_'], rtol=0, atol=2e-6)


def _fit_worker(rank, world, rdzv, out, prepare='local'):
    dist.init_process_group('gloo', init_method=rdzv, rank=rank, world_size=world)
    from test_gpu_fit import _frame
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    ds = InteractionDataset.read_df(_frame(), verbose=False)
    model = CDAE(hidden_factors=18, mode='sampled', device_sampler=True, seed=3, verbose=False, layout='columns')
    model.fit(ds, epochs=20, batch_size=384, learning_rate=0.05, reg_rate=1e-3, neg_ratio=5, prepare=prepare)
    torch.cuda.synchronize()
    frame = _frame()
    torch.save({'params': model._engine.get_params(), 'pred': float(model.predict(frame['user'][0], frame['item'][1])),
                'rank': model.rank(frame['user'][0], list(frame['item'][:30]), n=5)}, f'{out}.{rank}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('prepare', ['local', 'turns'])
@retry_infra
def test_public_fit_under_a_process_group_equals_the_single_gpu_fit(tmp_path, prepare):
    """CDAE.fit(mode='sampled', device_sampler=True, layout='columns') as two processes of one job (column-sharded training, then every rank
    holds the whole model): parameters, a prediction and a ranking equal the single-process fit with the same seed."""
    from test_gpu_fit import _frame
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    out = str(tmp_path / 'fit')
    rdzv = new_rendezvous(tmp_path)
    mp.spawn(_fit_worker, args=(2, rdzv, out, prepare), nprocs=2, join=True)
    frame = _frame()
    ds = InteractionDataset.read_df(frame, verbose=False)
    single = CDAE(hidden_factors=18, mode='sampled', device_sampler=True, seed=3, verbose=False)
    single.fit(ds, epochs=20, batch_size=384, learning_rate=0.05, reg_rate=1e-3, neg_ratio=5)
    want = single._engine.get_params()
    want_pred = float(single.predict(frame['user'][0], frame['item'][1]))
    want_rank = single.rank(frame['user'][0], list(frame['item'][:30]), n=5)
    for r in range(2):
        got = torch.load(f'{out}.{r}', weights_only=False)
        for k in want:
            np.testing.assert_allclose(got['params'][k], want[k], rtol=0, atol=2e-6, err_msg=k)
        assert abs(got['pred'] - want_pred) < 1e-6
        assert [i for _, i in got['rank']] == [i for _, i in want_rank]


def test_column_sharded_long_histories_take_the_workgroup_forward():
    """Mean history of ~60 items, 48 triples: k_kshard_fwd_wg (one workgroup per triple) — against the direct single-GPU step on
    the same columns, which runs its own workgroup variant."""
    from oracle import cdae_oracle as co
    from helpers import synth_history
    from drecpy_amd.dist import ColumnShardedCdae
    from drecpy_amd.engine import CdaeEngine
    rng = np.random.default_rng(9)
    Ul, Nl, Kl, Bl = 40, 400, 64, 48
    p = co.init_params(rng, Ul, Nl, Kl, np.float32)
    indptr, indices = synth_history(rng, Ul, Nl, 60, zipf=1.05)
    m = ColumnShardedCdae(Ul, Nl, Kl, 0, 1, 'cuda:0', indptr, indices, q=Q)
    m.set_params_global(**p)
    e = CdaeEngine(Ul, Nl, Kl); e.set_params(**p); e.set_history(indptr, indices); e.init_optimizer('adagrad', 0.05, 1e-3)
    for s in range(4):
        uid, iid, y = rng.integers(0, Ul, size=Bl), rng.integers(0, Nl, size=Bl), (rng.random(Bl) < 0.3).astype(np.float32)
        bt, alive = m.engine.make_batch(uid, iid, y, q=Q, mask_seed=s)
        assert bt.n_touch_slots > 16 * Bl
        m.step(s, bt)
        bt2, alive2 = e.make_batch(uid, iid, y, q=Q, mask_seed=s)
        e.step_sparse(s, bt2)
    torch.cuda.synchronize()
    for a, b in zip(m.engine.tables(), e.tables()):
        assert float((a - b).abs().max()) < 1e-6
