"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Gates: predictions within 1e-5 relative of the fp64 oracle after 0, 1, 10 and 50 steps
from injected weights / masks; integer outputs bit-exact."""
import numpy as np
import pytest

from oracle import cdae_oracle as co
from helpers import batch_rows, hash_u32, q_threshold, synth_history, x_tilde

pytestmark = pytest.mark.gpu
REL = 1e-5


def _engine(U, N, K, seed=0):
    from drecpy_amd.engine import CdaeEngine
    rng = np.random.default_rng(seed)
    p = co.init_params(rng, U, N, K, np.float64)
    eng = CdaeEngine(U, N, K)
    eng.set_params(**{k: v for k, v in p.items()})
    return eng, p, rng


def _relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


@pytest.mark.parametrize('K', [8, 50, 128, 200, 300])
def test_forward_matches_oracle(K):
    U, N = 57, 211
    eng, p, rng = _engine(U, N, K)
    indptr, indices = synth_history(rng, U, N, 12)
    eng.set_history(indptr, indices)
    uids = rng.integers(0, U, size=37)
    # inference path (cdae.py:67-71): uncorrupted, unscaled
    h, pred = eng.forward(uids)
    t, keep_off, _ = batch_rows(indptr, indices, uids, N)
    ho, po = co.forward(p, uids, t.astype(np.float64))
    assert _relerr(h.cpu().numpy(), ho) < REL
    assert _relerr(pred.cpu().numpy(), po) < REL
    # training forward with an explicit keep stream
    keep = (rng.random(keep_off[-1]) >= 0.2).astype(np.uint8)
    _, _, kept = batch_rows(indptr, indices, uids, N, keep)
    h, pred = eng.forward(uids, keep_off=keep_off, keep=keep, q=0.2)
    ho, po = co.forward(p, uids, x_tilde(t, kept, 0.2, np.float64))
    assert _relerr(pred.cpu().numpy(), po) < REL
    # counter-based mask
    seed = 0x1234ABCD5678
    h, pred = eng.forward(uids, q=0.3, mask_seed=seed)
    thr = q_threshold(0.3)
    keep2 = np.concatenate([hash_u32(seed, np.full(keep_off[b + 1] - keep_off[b], b), np.arange(keep_off[b + 1] - keep_off[b])) >= thr
                            for b in range(len(uids))]).astype(np.uint8)
    _, _, kept2 = batch_rows(indptr, indices, uids, N, keep2)
    ho, po = co.forward(p, uids, x_tilde(t, kept2, float(np.float32(0.3)), np.float64))
    assert _relerr(pred.cpu().numpy(), po) < REL


def test_dense_step_with_more_output_tiles_than_workgroups():
    """N = 4500 output units = 563 tiles of 8 for the 512 persistent workgroups of k_out_dense_tile: some take two tiles (the dh
    partial of a workgroup then sums both), against the oracle after two steps."""
    U, N, K, B = 40, 4500, 16, 32
    eng, p, rng = _engine(U, N, K, seed=3)
    indptr, indices = synth_history(rng, U, N, 20)
    eng.set_history(indptr, indices)
    eng.init_optimizer('adam', 1e-3, 1e-3)
    st = co.adam_state(p)
    for step in range(2):
        uids = rng.integers(0, U, size=B)
        t, keep_off, _ = batch_rows(indptr, indices, uids, N)
        keep = (rng.random(keep_off[-1]) >= 0.2).astype(np.uint8)
        _, _, kept = batch_rows(indptr, indices, uids, N, keep)
        bt, alive = eng.make_batch(uids, keep_off=keep_off, keep=keep, q=0.2)
        lo = co.dense_step(p, st, step, uids, x_tilde(t, kept, float(np.float32(0.2)), np.float64), t, 1e-3, 1e-3, 'bce', 'reference')
        lg = eng.step_dense(step, bt, 'bce', 'reference', want_loss=True).cpu().numpy()
        assert abs(lg.sum() - lo) / abs(lo) < 1e-4
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('K,B,loss,targets', [(50, 64, 'bce', 'reference'), (128, 64, 'bce', 'reference'),
                                              (8, 33, 'mse', 'reference'), (50, 40, 'bce', 'per_row'),
                                              (300, 16, 'mse', 'per_row'), (128, 700, 'bce', 'reference')])
def test_dense_steps_match_oracle(K, B, loss, targets):
    U, N = 90, 173
    eng, p, rng = _engine(U, N, K, seed=1)
    indptr, indices = synth_history(rng, U, N, 15)
    eng.set_history(indptr, indices)
    eng.init_optimizer('adam', 1e-3, 1e-3)
    st = co.adam_state(p)
    p32 = {k: v.astype(np.float32) for k, v in p.items()}     # fp32 oracle: what TF itself computes in
    st32 = co.adam_state(p32)
    q = 0.2
    checks = (1, 10, 50) if B <= 64 else (1, 3)
    probe = rng.integers(0, U, size=20)
    tp, _, _ = batch_rows(indptr, indices, probe, N)
    for step in range(max(checks)):
        uids = rng.integers(0, U, size=B)
        t, keep_off, _ = batch_rows(indptr, indices, uids, N)
        keep = (rng.random(keep_off[-1]) >= q).astype(np.uint8)
        _, _, kept = batch_rows(indptr, indices, uids, N, keep)
        bt, alive = eng.make_batch(uids, keep_off=keep_off, keep=keep, q=q)
        lo = co.dense_step(p, st, step, uids, x_tilde(t, kept, float(np.float32(q)), np.float64), t, 1e-3, 1e-3, loss, targets)
        co.dense_step(p32, st32, step, uids, x_tilde(t, kept, float(np.float32(q)), np.float32), t, 1e-3, 1e-3, loss, targets)
        lg = eng.step_dense(step, bt, loss, targets, want_loss=True).cpu().numpy()
        assert abs(lg.sum() - lo) / abs(lo) < 1e-4, (step, lg, lo)
        if step + 1 in checks:
            _, pred = eng.forward(probe)
            _, po = co.forward(p, probe, tp.astype(np.float64))
            _, po32 = co.forward(p32, probe, tp.astype(np.float32))
            err, drift32 = _relerr(pred.cpu().numpy(), po), _relerr(po32, po)
            # gate: 1e-5 vs the fp64 oracle; after many Adam steps fp32 rounding (TF's own included) is amplified by
            # m/(sqrt(v)+eps), so the budget there is twice the fp32 oracle's own drift from fp64
            assert err < max(REL, 2.0 * drift32), (step, err, drift32)
            assert err < 5e-5
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=5e-5)


@pytest.mark.parametrize('K,B,opt,loss,explicit', [(128, 256, 'adagrad', 'bce', True), (50, 64, 'adagrad', 'bce', False),
                                                   (8, 100, 'adam', 'mse', True), (128, 4096, 'adagrad', 'bce', False),
                                                   (300, 50, 'adam', 'bce', False), (128, 300, 'rowwise_adagrad', 'bce', False),
                                                   (50, 64, 'rowwise_adagrad', 'mse', True)])
@pytest.mark.parametrize('prepared', [False, True])
def test_sparse_steps_match_oracle(K, B, opt, loss, explicit, prepared):
    """prepared=True: the touch list is built ahead by drx_cdae_sparse_prepare, which also marks the V / W2T rows a single
    sample touches; those are then updated by the forward kernel (batches of 50-100 over 120 users and 260 items mix sole
    and shared rows)."""
    U, N = 120, 260
    eng, p, rng = _engine(U, N, K, seed=2)
    indptr, indices = synth_history(rng, U, N, 14, zipf=1.1)
    eng.set_history(indptr, indices)
    lr = 1e-3 if opt == 'adam' else 0.05
    eng.init_optimizer(opt, lr, 1e-3)
    st = co.sparse_state(p, opt)
    q = 0.2
    qf = float(np.float32(q))
    probe = np.arange(U)
    tp, _, _ = batch_rows(indptr, indices, probe, N)
    n_steps = 10 if B <= 256 else 3
    for step in range(n_steps):
        uids = rng.integers(0, U, size=B)
        iids = rng.integers(0, N, size=B)
        y = (rng.random(B) < 0.3).astype(np.float32)
        t, keep_off, _ = batch_rows(indptr, indices, uids, N)
        seed = 977 + step * 7919
        if explicit:
            keep = (rng.random(keep_off[-1]) >= q).astype(np.uint8)
            bt, alive = eng.make_batch(uids, iids, y, keep_off=keep_off, keep=keep, q=q)
        else:
            thr = q_threshold(q)
            keep = np.concatenate([hash_u32(seed, np.full(keep_off[b + 1] - keep_off[b], b), np.arange(keep_off[b + 1] - keep_off[b])) >= thr
                                   for b in range(B)]).astype(np.uint8)
            bt, alive = eng.make_batch(uids, iids, y, q=q, mask_seed=seed)
        _, _, kept = batch_rows(indptr, indices, uids, N, keep)
        lo, _ = co.sparse_step(p, st, step, uids, iids, y, kept, qf, lr, 1e-3, loss, opt)
        lg = eng.step_sparse(step, bt, loss, want_loss=True, prepared=eng.prepare_sparse(bt) if prepared else None).cpu().numpy()
        assert abs(lg[0] - lo) / abs(lo) < 1e-4, (step, lg, lo)
    _, pred = eng.forward(probe)
    _, po = co.forward(p, probe, tp.astype(np.float64))
    assert _relerr(pred.cpu().numpy(), po) < REL
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=2e-5)


@pytest.mark.parametrize('prepared', [False, True])
@pytest.mark.parametrize('K', [50, 128])
def test_sparse_steps_with_many_more_users_than_triples(K, prepared):
    """3000 users, 64 triples: a key space far wider than the batch.  Users repeat inside a batch (shared V rows go through the
    list) and across batches."""
    U, N, B = 3000, 260, 64
    eng, p, rng = _engine(U, N, K, seed=4)
    indptr, indices = synth_history(rng, U, N, 9, zipf=1.1)
    eng.set_history(indptr, indices)
    eng.init_optimizer('adagrad', 0.05, 1e-3)
    st = co.sparse_state(p, 'adagrad')
    q = 0.2
    qf = float(np.float32(q))
    pool = rng.integers(0, U, size=200)
    for step in range(8):
        uids = np.concatenate([rng.integers(0, U, size=B - 12), rng.choice(pool[:5], size=12)])      # a few users several times
        rng.shuffle(uids)
        iids = rng.integers(0, N, size=B)
        y = (rng.random(B) < 0.3).astype(np.float32)
        t, keep_off, _ = batch_rows(indptr, indices, uids, N)
        seed = 4242 + step
        keep = np.concatenate([hash_u32(seed, np.full(keep_off[b + 1] - keep_off[b], b), np.arange(keep_off[b + 1] - keep_off[b])) >= q_threshold(q)
                               for b in range(B)]).astype(np.uint8)
        bt, alive = eng.make_batch(uids, iids, y, q=q, mask_seed=seed)
        _, _, kept = batch_rows(indptr, indices, uids, N, keep)
        lo, _ = co.sparse_step(p, st, step, uids, iids, y, kept, qf, 0.05, 1e-3, 'bce', 'adagrad')
        lg = eng.step_sparse(step, bt, 'bce', want_loss=True, prepared=eng.prepare_sparse(bt) if prepared else None).cpu().numpy()
        assert abs(lg[0] - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=2e-5)


@pytest.mark.parametrize('prepared', [False, True])
@pytest.mark.parametrize('K,opt', [(128, 'adagrad'), (50, 'adam'), (16, 'adagrad')])
def test_sparse_steps_with_hot_segments(K, opt, prepared):
    """60 items under a steep Zipf law, 768 triples: most rows collect hundreds of touches — segments that cross many chunk
    borders (block partials of all-inner workgroups, short and long spans of the span launch) beside short ones: same oracle."""
    U, N, B = 400, 60, 768
    eng, p, rng = _engine(U, N, K, seed=11)
    indptr, indices = synth_history(rng, U, N, 12, zipf=1.4)
    eng.set_history(indptr, indices)
    lr = 1e-3 if opt == 'adam' else 0.05
    eng.init_optimizer(opt, lr, 1e-3)
    st = co.sparse_state(p, opt)
    q = 0.2
    qf = float(np.float32(q))
    for step in range(5):
        uids = rng.integers(0, U, size=B)
        iids = np.minimum((rng.pareto(1.2, size=B)).astype(np.int64), N - 1)          # hot OUTPUT rows too
        y = (rng.random(B) < 0.3).astype(np.float32)
        t, keep_off, _ = batch_rows(indptr, indices, uids, N)
        seed = 900 + step
        keep = np.concatenate([hash_u32(seed, np.full(keep_off[b + 1] - keep_off[b], b), np.arange(keep_off[b + 1] - keep_off[b])) >= q_threshold(q)
                               for b in range(B)]).astype(np.uint8)
        bt, alive = eng.make_batch(uids, iids, y, q=q, mask_seed=seed)
        _, _, kept = batch_rows(indptr, indices, uids, N, keep)
        lo, _ = co.sparse_step(p, st, step, uids, iids, y, kept, qf, lr, 1e-3, 'bce', opt)
        lg = eng.step_sparse(step, bt, 'bce', want_loss=True, prepared=eng.prepare_sparse(bt) if prepared else None).cpu().numpy()
        assert abs(lg[0] - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5)


@pytest.mark.parametrize('prepared', [False, True])
@pytest.mark.parametrize('K,users', [(128, 30000), (64, 30000), (256, 12000), (125, 30000)])
def test_streamed_reduction_with_hot_rows_in_a_short_list(K, users, prepared):
    """Lists of SHORT segments over rows of exactly 64 / 128 / 256 floats with Adagrad take the streamed reduction
    (csrc/drx_segstream.hpp: one chunk per wave, its rows through an LDS ring by LDS-DMA).  A key space far wider than the batch
    (30 000 users: fewer than 8 touches per table row, the rule of csrc/drx_prep.hpp) with 60 items under a steep Zipf law and hot
    OUTPUT rows: W and W2T rows that collect hundreds of touches (block partials of all-inner blocks, head / tail partials, short and
    long spans, the output bias's scalar sums) beside rows of a few, users that repeat (V rows through the list), blanked touches of
    sole-toucher rows when the list is prepared ahead.  K = 125: rows padded to 128 floats.  Same oracle, same tolerance."""
    U, N, B = users, 60, 768
    eng, p, rng = _engine(U, N, K, seed=13)
    indptr, indices = synth_history(rng, U, N, 12, zipf=1.4)
    eng.set_history(indptr, indices)
    eng.init_optimizer('adagrad', 0.05, 1e-3)
    st = co.sparse_state(p, 'adagrad')
    q = 0.2
    qf = float(np.float32(q))
    pool = rng.integers(0, U, size=40)
    for step in range(4):
        uids = np.concatenate([rng.integers(0, U, size=B - 64), rng.choice(pool[:6], size=64)])      # a few users many times
        rng.shuffle(uids)
        iids = np.minimum((rng.pareto(1.2, size=B)).astype(np.int64), N - 1)
        y = (rng.random(B) < 0.3).astype(np.float32)
        t, keep_off, _ = batch_rows(indptr, indices, uids, N)
        seed = 7100 + step
        keep = np.concatenate([hash_u32(seed, np.full(keep_off[b + 1] - keep_off[b], b), np.arange(keep_off[b + 1] - keep_off[b])) >= q_threshold(q)
                               for b in range(B)]).astype(np.uint8)
        bt, alive = eng.make_batch(uids, iids, y, q=q, mask_seed=seed)
        assert bt.n_touch_slots + 2 * B <= 8 * (2 * N + U)                        # short segments: the rule of csrc/drx_prep.hpp
        _, _, kept = batch_rows(indptr, indices, uids, N, keep)
        lo, _ = co.sparse_step(p, st, step, uids, iids, y, kept, qf, 0.05, 1e-3, 'bce', 'adagrad')
        lg = eng.step_sparse(step, bt, 'bce', want_loss=True, prepared=eng.prepare_sparse(bt) if prepared else None).cpu().numpy()
        assert abs(lg[0] - lo) / abs(lo) < 1e-4, (step, lg, lo)
    probe = np.concatenate([pool[:6], rng.integers(0, U, size=58)])
    tp, _, _ = batch_rows(indptr, indices, probe, N)
    _, pred = eng.forward(probe)
    _, po = co.forward(p, probe, tp.astype(np.float64))
    assert _relerr(pred.cpu().numpy(), po) < REL
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)


@pytest.mark.parametrize('prepared', [False, True])
def test_sparse_long_histories_take_the_workgroup_path(prepared):
    """Mean history of ~60 items, 48 triples: the forward/backward runs one WORKGROUP per triple (k_sampled_fwd_bwd_wg: its
    groups split the history, partial bags summed in LDS) — same oracle, same tolerance."""
    U, N, K, B = 40, 400, 128, 48
    eng, p, rng = _engine(U, N, K, seed=9)
    indptr, indices = synth_history(rng, U, N, 60, zipf=1.05)
    assert (indptr[-1] / U) > 40
    eng.set_history(indptr, indices)
    eng.init_optimizer('adagrad', 0.05, 1e-3)
    st = co.sparse_state(p, 'adagrad')
    q = 0.2
    qf = float(np.float32(q))
    for step in range(6):
        uids = rng.integers(0, U, size=B)
        iids = rng.integers(0, N, size=B)
        y = (rng.random(B) < 0.3).astype(np.float32)
        t, keep_off, _ = batch_rows(indptr, indices, uids, N)
        seed = 31 + step
        keep = np.concatenate([hash_u32(seed, np.full(keep_off[b + 1] - keep_off[b], b), np.arange(keep_off[b + 1] - keep_off[b])) >= q_threshold(q)
                               for b in range(B)]).astype(np.uint8)
        bt, alive = eng.make_batch(uids, iids, y, q=q, mask_seed=seed)
        _, _, kept = batch_rows(indptr, indices, uids, N, keep)
        lo, _ = co.sparse_step(p, st, step, uids, iids, y, kept, qf, 0.05, 1e-3, 'bce', 'adagrad')
        lg = eng.step_sparse(step, bt, 'bce', want_loss=True, prepared=eng.prepare_sparse(bt) if prepared else None).cpu().numpy()
        assert abs(lg[0] - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=5e-5)



def test_sparse_step_is_deterministic():
    import torch
    U, N, K, B = 200, 300, 128, 2048
    outs = []
    for rep in range(2):
        eng, p, rng = _engine(U, N, K, seed=5)
        indptr, indices = synth_history(rng, U, N, 20, zipf=1.2)
        eng.set_history(indptr, indices)
        eng.init_optimizer('adagrad', 0.05, 1e-3)
        for step in range(3):
            uids = rng.integers(0, U, size=B); iids = rng.integers(0, N, size=B)
            y = (rng.random(B) < 0.3).astype(np.float32)
            bt, alive = eng.make_batch(uids, iids, y, q=0.2, mask_seed=step)
            eng.step_sparse(step, bt)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in eng.tables()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('explicit', [False, True])
@pytest.mark.parametrize('K,opt,U', [(128, 'adagrad', 150), (50, 'adam', 150), (16, 'adagrad', 150), (128, 'adagrad', 24), (200, 'adam', 24),
                                     (128, 'adagrad', 8)])
def test_lists_prepared_through_the_historys_transpose_match_the_oracle_and_the_sorted_lists(K, opt, U, explicit):
    """Lists of long segments (more than 8 touches per table row: MovieLens shapes) are prepared by expanding the history's transpose
    (DrxHistory::t_rank: only the batch's 2B (id, sample) pairs are sorted) instead of sorting every (row, sample) pair: same oracle;
    against the sort path the parameters agree to rounding (inside a segment the touches come user by user instead of sample by
    sample: another fixed order of the same sum), and two runs of the transposed path agree bit for bit.  'shared'
    (DRX_BATCH_SHARE_USERS): the ~7 triples of a user share their gather (full sum minus the dropped rows) and their gradient (the user's
    summed row minus the droppers'): same oracle, same tolerance.  U = 24: ~43 triples per user, three work items of up to 16 triples
    each (csrc/drx_prep.hpp k_tp_item_*); U = 8: ~128 triples per user — more than the 64 whose keep bits the count pass hands the
    write pass (k_tp_write evaluates the rest itself); K = 16: rows too narrow for the shared form (share_geometry_ok) — the plain transposed list."""
    N, B = 70, 1024
    results = []
    for mode in ('transpose', 'transpose', 'sort', 'shared', 'shared'):
        eng, p, rng = _engine(U, N, K, seed=21)
        indptr, indices = synth_history(rng, U, N, 14, zipf=1.0)
        eng.set_history(indptr, indices, with_transpose=mode != 'sort')
        eng.share_users = mode == 'shared'             # DRX_BATCH_SHARE_USERS: a user's triples share their gather and their gradient
        assert (eng._hist_t is not None) == (mode != 'sort')
        lr = 1e-3 if opt == 'adam' else 0.05
        eng.init_optimizer(opt, lr, 1e-3)
        st = co.sparse_state(p, opt)
        q = 0.2
        qf = float(np.float32(q))
        for step in range(4):
            uids = rng.integers(0, U, size=B)
            iids = rng.integers(0, N, size=B)
            y = (rng.random(B) < 0.3).astype(np.float32)
            t, keep_off, _ = batch_rows(indptr, indices, uids, N)
            seed = 300 + step
            if explicit:
                keep = (rng.random(keep_off[-1]) >= q).astype(np.uint8)
                bt, alive = eng.make_batch(uids, iids, y, keep_off=keep_off, keep=keep, q=q)
            else:
                keep = np.concatenate([hash_u32(seed, np.full(keep_off[b + 1] - keep_off[b], b), np.arange(keep_off[b + 1] - keep_off[b])) >= q_threshold(q)
                                       for b in range(B)]).astype(np.uint8)
                bt, alive = eng.make_batch(uids, iids, y, q=q, mask_seed=seed)
            assert bt.n_touch_slots + 2 * B > 8 * (2 * N + U)                      # long segments: the rule of csrc/drx_prep.hpp
            _, _, kept = batch_rows(indptr, indices, uids, N, keep)
            lo, _ = co.sparse_step(p, st, step, uids, iids, y, kept, qf, lr, 1e-3, 'bce', opt)
            lg = eng.step_sparse(step, bt, 'bce', want_loss=True, prepared=eng.prepare_sparse(bt)).cpu().numpy()
            assert abs(lg[0] - lo) / abs(lo) < 1e-4, (mode, step, lg, lo)
        g = eng.get_params()
        for k in p:
            np.testing.assert_allclose(g[k], p[k], rtol=0, atol=2e-5, err_msg=f'{mode} {k}')
        # ... and the gate every other sparse test applies: predictions within 1e-5 relative of the oracle's
        probe = np.arange(min(U, 64))
        tpr, _, _ = batch_rows(indptr, indices, probe, N)
        _, pred = eng.forward(probe)
        _, po = co.forward(p, probe, tpr.astype(np.float64))
        assert _relerr(pred.cpu().numpy(), po) < REL, mode
        results.append(g)
    for k in results[0]:
        assert np.array_equal(results[0][k], results[1][k]), k                      # the transposed path twice: bit for bit
        np.testing.assert_allclose(results[0][k], results[2][k], rtol=0, atol=2e-6, err_msg=k)
        assert np.array_equal(results[3][k], results[4][k]), k                      # ... and the shared form twice
        np.testing.assert_allclose(results[3][k], results[2][k], rtol=0, atol=5e-6, err_msg=k)


@pytest.mark.parametrize('K', [16, 32, 128])
@pytest.mark.parametrize('fill', [0, 0x11])
def test_compact_lists_never_read_the_slots_behind_their_real_length(K, fill):
    """A list laid down by the transposed preparation is COMPACT: its real touches, a few blanked chunks, then whatever the buffer held
    before (SpanPlan::cnt[20] says where the list ends).  With rows of 16 / 32 floats a reduction workgroup covers 32 / 16 chunks:
    more than the blanked tail.  The buffer is handed over full of bytes that read as VALID keys and samples (key 0 / sample 0, or
    0x11111111 — an out-of-range key and sample): the step must give the result of a clean buffer, bit for bit."""
    import torch
    U, N, B = 150, 70, 1024
    results = []
    for poison in (None, fill, fill):
        eng, p, rng = _engine(U, N, K, seed=33)
        indptr, indices = synth_history(rng, U, N, 14, zipf=1.0)
        eng.set_history(indptr, indices, with_transpose=True)
        eng.init_optimizer('adagrad', 0.05, 1e-3)
        buf = None
        for step in range(3):
            uids = rng.integers(0, U, size=B); iids = rng.integers(0, N, size=B)
            y = (rng.random(B) < 0.3).astype(np.float32)
            bt, alive = eng.make_batch(uids, iids, y, q=0.2, mask_seed=900 + step)
            buf = eng.prep_buffer(bt, buf)
            if poison is None:
                buf.fill_(0xFF)                      # DRX_KEY_NONE everywhere: what the blanked tail looks like
            else:
                buf.fill_(poison)
            eng.step_sparse(step, bt, 'bce', prepared=eng.prepare_sparse(bt, buf))
        torch.cuda.synchronize()
        results.append(eng.get_params())
    for k in results[0]:
        assert np.array_equal(results[0][k], results[1][k]), (K, fill, k)
        assert np.array_equal(results[0][k], results[2][k]), (K, fill, k)


def test_device_sampler_in_user_order_draws_the_same_triples():
    """drx_point_sample_by_user: the draws of drx_point_sample_recorded for the same seed, sorted by user (a user's triples in draw
    order), keep_off the prefix sums of the sorted users' degrees, the total posted to the mailbox."""
    import torch
    U, N, B = 200, 90, 4096
    eng, p, rng = _engine(U, N, 16, seed=8)
    indptr, indices = synth_history(rng, U, N, 9, zipf=1.0, min_deg=1)
    eng.set_history(indptr, indices)
    assert eng._hist_t is not None
    mb = torch.zeros(1, dtype=torch.int64).pin_memory()
    su, si, sy, sko = [t.clone() for t in eng.sample_device(B, 3, 77, mailbox=mb, tag=5)]
    eng.sample_by_user = False
    du, di, dy, _ = [t.clone() for t in eng.sample_device(B, 3, 77)]
    torch.cuda.synchronize()
    du, di, dy = du.cpu().numpy(), di.cpu().numpy(), dy.cpu().numpy()
    order = np.argsort(du, kind='stable')
    assert np.array_equal(su.cpu().numpy(), du[order]) and np.array_equal(si.cpu().numpy(), di[order]) and np.array_equal(sy.cpu().numpy(), dy[order])
    deg = (indptr[1:] - indptr[:-1])[du[order]]
    assert np.array_equal(sko.cpu().numpy(), np.concatenate([[0], np.cumsum(deg)]))
    assert int(mb[0]) == (5 << 32) | int(deg.sum())
