"""The sampled step at BASELINE sizes: size-independent properties (the oracle cannot follow whole tables of this scale) and — r06 —
one full-size step against the oracle on tables compacted to the rows the batch names (the last test of this file):
  * ml-1m-shaped (config 2 of BASELINE.json: U=6040, N=3706, K=128), a 1M-user slice of the 10M x 1M set, and the
    whole 10M-user x 1M-item x ~200M-interaction set of config 4 (one GPU holds all of it: 12.3 GB of tables + slots)
  * bit-reproducibility of a step sequence, `prepared` (side-stream) == inline touch lists bit for bit,
    rows no triple touches keep their bits (parameters AND optimizer slots), the row-sharded path at world 1 agrees with
    the direct path, the forward of a user is the same whether it is computed alone or inside a large batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(shape, users=None, K=128):
    from drecpy_amd import synth
    from drecpy_amd.engine import CdaeEngine
    U, N, md, mn, a = synth.SHAPES[shape]
    U = users or U
    ip, idx = synth.synth_history(U, N, md, mn, a, seed=0, device='cuda', user_hi=U)
    eng = CdaeEngine(U, N, K)
    eng.init_glorot_device(10)
    eng.set_history(ip, idx)
    eng.init_optimizer('adagrad', 0.05, 1e-3)
    return eng, U, N, ip, idx


def _run(eng, N, B, steps, prepared):
    for s in range(steps):
        uid, iid, y, ko = eng.sample_device(B, 5, 100 + s, n_items=N)
        bt, alive = eng.make_batch(uid, iid, y, keep_off=ko, q=0.2, mask_seed=7 + s)
        prep = eng.prepare_sparse(bt) if prepared else None
        eng.step_sparse(s, bt, prepared=prep)
    torch.cuda.synchronize()
    return [t.clone() for t in eng.tables()] + [t.clone() for t in eng.s1]


@pytest.mark.parametrize('shape,users,B', [('ml-1m', None, 16384), ('synth-10m', 1_000_000, 65536), ('synth-10m', None, 65536)])
def test_full_size_properties(shape, users, B):
    eng, U, N, ip, idx = _setup(shape, users)
    a = _run(eng, N, B, 3, prepared=False)
    eng2, _, _, _, _ = _setup(shape, users)
    shared_form = eng2._hist_t is not None             # (MovieLens shapes: lists prepared AHEAD are in the shared form, DRX_BATCH_SHARE_USERS)
    eng2.share_users = False
    b = _run(eng2, N, B, 3, prepared=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)                       # deterministic, and prepared == inline bit for bit
    if shared_form:
        # the shared form: another association of the same sums — bit-reproducible, and the inline steps' tables to rounding
        outs = []
        for _ in range(2):
            eng3, _, _, _, _ = _setup(shape, users)
            assert eng3.share_users
            outs.append(_run(eng3, N, B, 3, prepared=True))
        for x, y, z in zip(outs[0], outs[1], a):
            assert torch.equal(x, y)
            assert float((x - z).abs().max()) < 2e-5 * max(1.0, float(z.abs().max()))
    # rows that no triple of a further step touches keep their bits (parameters and Adagrad accumulators)
    before = [t.clone() for t in eng.tables()] + [t.clone() for t in eng.s1]
    uid, iid, y, ko = eng.sample_device(B, 5, 999, n_items=N)
    bt, alive = eng.make_batch(uid, iid, y, keep_off=ko, q=0.0, mask_seed=1)       # q = 0: every history item is touched
    eng.step_sparse(3, bt)
    torch.cuda.synchronize()
    after = [t.clone() for t in eng.tables()] + [t.clone() for t in eng.s1]
    touched_u = torch.zeros(U, dtype=torch.bool, device='cuda'); touched_u[uid.long()] = True
    touched_o = torch.zeros(N, dtype=torch.bool, device='cuda'); touched_o[iid.long()] = True
    touched_w = torch.zeros(N, dtype=torch.bool, device='cuda')
    deg = (ip[uid.long() + 1] - ip[uid.long()])
    rows = torch.repeat_interleave(ip[uid.long()], deg) + (torch.arange(int(deg.sum()), device='cuda') - torch.repeat_interleave(torch.cumsum(deg, 0) - deg, deg))
    touched_w[idx[rows].long()] = True
    for off in (0, 5):                                    # parameters, then optimizer slots
        W0, O0, V0, _, b20 = before[off:off + 5]
        W1, O1, V1, _, b21 = after[off:off + 5]
        assert torch.equal(W0[~touched_w], W1[~touched_w]) and torch.equal(O0[~touched_o], O1[~touched_o])
        assert torch.equal(V0[~touched_u], V1[~touched_u]) and torch.equal(b20[~touched_o], b21[~touched_o])
        if off == 0:        # (g^2 of a single user's row can vanish against the 0.1 accumulator in fp32: parameters only)
            assert not torch.equal(W0[touched_w], W1[touched_w]) and not torch.equal(V0[touched_u], V1[touched_u])
    # a user's forward does not depend on its batch neighbours
    probe = torch.randint(0, U, (4096,), device='cuda', dtype=torch.int32)
    h_big, _ = eng.forward(probe, want_pred=False)
    h_one, _ = eng.forward(probe[17:18], want_pred=False)
    assert torch.equal(h_big[17], h_one[0]) or float((h_big[17] - h_one[0]).abs().max()) < 1e-6


def test_sharded_world1_agrees_with_direct_at_scale():
    from drecpy_amd import synth
    from drecpy_amd.dist import ShardedCdae
    eng, U, N, ip, idx = _setup('synth-10m', 500_000)
    m = ShardedCdae(U, N, 128, 0, 1, 'cuda:0', ip, idx, q=0.2)
    for dst, src in zip(m.engine.tables(), eng.tables()):
        dst.copy_(src)
    B = 32768
    for s in range(2):
        uid, iid, y, ko = eng.sample_device(B, 5, 50 + s, n_items=N)
        bt, alive = eng.make_batch(uid, iid, y, keep_off=ko, q=0.2, mask_seed=3 + s)
        eng.step_sparse(s, bt)
        bt2, alive2 = m.engine.make_batch(uid, iid, y, keep_off=ko, q=0.2, mask_seed=3 + s)
        m.step(s, bt2)
    torch.cuda.synchronize()
    for name, x, y in zip(('W', 'W2T', 'V', 'b', 'b2'), eng.tables(), m.engine.tables()):
        assert float((x - y).abs().max()) < 2e-6, name        # same sums, different chunking of the sorted touches


def test_pipeline_batches_are_the_draws_of_their_seeds_while_the_callers_stream_is_busy():
    """SampledPipeline's first run-ahead draws must not share anything with work on the caller's stream (r03: placeholder draws queued
    there used the same sampler scratch and ring tensors, and whichever ran last won — a batch whose offsets belonged to another draw
    than its users).  The caller's stream is kept busy while the pipeline is built; every batch it then trains on must be the draw of
    its own seed, and its touch count the sum of its users' history lengths."""
    from drecpy_amd.engine import SampledPipeline
    eng, U, N, ip, idx = _setup('synth-10m', users=200_000)
    B = 8192
    seed_of = lambda s: 4242 + 31 * s
    x = torch.randn(4096, 4096, device='cuda')
    for _ in range(60):                                  # tens of milliseconds of queued work on the caller's stream
        x = (x @ x).clamp_(-1, 1)
    pipe = SampledPipeline(eng, B, 5, 0.2, seed_of, seed_of, n_items=N)
    n_first = pipe.SA
    counts = [pipe._touch_count(s % pipe.RS, s) for s in range(n_first)]
    torch.cuda.synchronize()
    ring = [[t.clone() for t in pipe.ring[s % pipe.RS]] for s in range(n_first)]
    for s in range(n_first):
        want = eng.sample_device(B, 5, seed_of(s), n_items=N)
        torch.cuda.synchronize()
        for got, w in zip(ring[s], want):
            assert torch.equal(got, w), s
        assert counts[s] == int(want[3][-1].item()), s
    for _ in range(6):                                   # and the pipeline trains on from there
        pipe.run_step()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(eng.W).all())


def test_one_full_size_step_matches_the_oracle_on_compacted_tables():
    """BASELINE configuration 4 at FULL size against the oracle: one device-sampled batch of 65 536 triples on the
    real 10M-user x 1M-item tables through (a) step_sparse on a list prepared ahead — the streamed reduction, the bench's code path —
    and (b) ShardedCdae at world 1, against cdae_oracle.sparse_step(accumulate='matrix') (cdae.py:59-76 restricted to the sampled output
    unit; fp64) on a copy of the tables COMPACTED to the rows the batch's users and items name (bench_cpu.py's compaction: same
    arithmetic per triple, the untouched 99 % of the tables left out).  Touched rows to atol 2e-5, the parameters no triple touches bit-equal,
    predictions of touched users <= 1e-5 relative after the step."""
    from oracle import cdae_oracle as co
    from drecpy_amd.dist import ShardedCdae
    B, K, q, lr, reg, seed = 65536, 128, 0.2, 0.05, 1e-3, 4242
    eng, U, N, ip, idx = _setup('synth-10m')
    m = ShardedCdae(U, N, K, 0, 1, 'cuda:0', ip, idx, q=q, lr=lr, reg=reg)
    for dst, src in zip(m.engine.tables(), eng.tables()):
        dst.copy_(src)
    uid, iid, y, ko = eng.sample_device(B, 5, seed, n_items=N)
    torch.cuda.synchronize()
    # ---- compaction: every history item of the batch's users (kept or dropped) + the output items; the batch's users
    ul_, deg = uid.long(), (ip[uid.long() + 1] - ip[uid.long()])
    T = int(deg.sum())
    assert T == int(ko[-1].item())
    row = torch.repeat_interleave(torch.arange(B, device='cuda'), deg)
    j = torch.arange(T, device='cuda') - torch.repeat_interleave(torch.cumsum(deg, 0) - deg, deg)
    items = idx[(ip[ul_][row] + j)].long()
    keep = torch.as_tensor(co.drx_hash_u32(seed, row.cpu().numpy(), j.cpu().numpy()) >= co.q_threshold(q)).cuda()
    il = torch.unique(torch.cat([items, iid.long()]))
    uu = torch.unique(ul_)
    imap = torch.full((N,), -1, dtype=torch.long, device='cuda'); imap[il] = torch.arange(il.numel(), device='cuda')
    umap = torch.full((U,), -1, dtype=torch.long, device='cuda'); umap[uu] = torch.arange(uu.numel(), device='cuda')
    f64 = lambda t: t.double().cpu().numpy()
    p = {'W': f64(eng.W[il, :K]), 'W_': np.ascontiguousarray(f64(eng.W2T[il, :K]).T), 'V': f64(eng.V[uu, :K]), 'b': f64(eng.b[:K]),
         'b_': f64(eng.b2[il])}
    before = [t.clone() for t in eng.tables()]
    kept_flat = imap[items[keep]].cpu().numpy()
    kept_len = torch.zeros(B, dtype=torch.long, device='cuda').index_add_(0, row[keep], torch.ones(int(keep.sum()), dtype=torch.long, device='cuda'))
    kept = np.split(kept_flat, np.cumsum(kept_len.cpu().numpy())[:-1])
    cu, ci, yy = umap[ul_].cpu().numpy(), imap[iid.long()].cpu().numpy(), y.cpu().numpy().astype(np.float64)
    # ---- the two device paths
    bt, alive = eng.make_batch(uid, iid, y, keep_off=ko, q=q, mask_seed=seed)
    prep = eng.prepare_sparse(bt)
    eng.step_sparse(0, bt, prepared=prep)
    bt2, alive2 = m.engine.make_batch(uid, iid, y, keep_off=ko, q=q, mask_seed=seed)
    m.step(0, bt2)
    torch.cuda.synchronize()
    # ---- the oracle on the compacted copy
    st = co.sparse_state(p, 'adagrad')
    co.sparse_step(p, st, 0, cu, ci, yy, kept, float(np.float32(q)), lr, reg, 'bce', 'adagrad', accumulate='matrix')
    for name, e in (('direct', eng), ('rows layout, world 1', m.engine)):
        got = {'W': f64(e.W[il, :K]), 'W_': f64(e.W2T[il, :K]).T, 'V': f64(e.V[uu, :K]), 'b': f64(e.b[:K]), 'b_': f64(e.b2[il])}
        for k_ in p:
            np.testing.assert_allclose(got[k_], p[k_], rtol=0, atol=2e-5, err_msg=f'{name}: {k_}')
    # ---- rows the batch does not name keep their bits
    tw = torch.zeros(N, dtype=torch.bool, device='cuda'); tw[items[keep]] = True
    to = torch.zeros(N, dtype=torch.bool, device='cuda'); to[iid.long()] = True
    tu = torch.zeros(U, dtype=torch.bool, device='cuda'); tu[uu] = True
    for e in (eng, m.engine):
        assert torch.equal(before[0][~tw], e.W[~tw]) and torch.equal(before[1][~to], e.W2T[~to])
        assert torch.equal(before[2][~tu], e.V[~tu]) and torch.equal(before[4][~to], e.b2[~to])
    assert int(tw.sum()) > 150_000 and int(tu.sum()) > 60_000          # (the batch's footprint: ~200 k distinct W rows, ~65 k users)
    # ---- predictions of touched users after the step (cdae.py:67-71: uncorrupted, unscaled input) on the compacted item columns
    probe = uu[torch.linspace(0, uu.numel() - 1, 48, device='cuda').long()]
    _, pred = eng.forward(probe.to(torch.int32))
    pu = umap[probe].cpu().numpy()
    t = np.zeros((len(pu), il.numel()))
    for r, u in enumerate(probe.tolist()):
        t[r, imap[idx[ip[u]:ip[u + 1]].long()].cpu().numpy()] = 1.0
    _, po = co.forward(p, pu, t)
    rel = np.abs(f64(pred[:, il]) - po) / np.abs(po)
    assert rel.max() < 1e-5, rel.max()
