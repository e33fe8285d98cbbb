"""End-to-end drop-in test: CDAE(...).fit() on the GPU against the CPU oracle driven by the SAME seeded inputs
(injected initial weights, bit-exact PointSampler stream, MT19937 corruption stream) — SURVEY.md §8d parity gates."""
import random

import numpy as np
import pytest

from oracle import cdae_oracle as co
from oracle import data_oracle as do
from helpers import load_frames

pytestmark = pytest.mark.gpu


def _frame():
    f = load_frames()['pt_int_dense']          # 64 users x 40 items, values 0..5 (zeros exercise the threshold)
    return {k: v.copy() for k, v in f.items()}


def _oracle_fit(frame, K, epochs, B, seed, weights, loss='bce', q=0.2, lr=1e-3, reg=1e-3, thr=1e-3, neg_ratio=5):
    uid, _ = do.first_appearance_codes(frame['user'].tolist())
    iid, _ = do.first_appearance_codes(frame['item'].tolist())
    U, N = int(uid.max()) + 1, int(iid.max()) + 1
    indptr, cols, vals = do.interaction_csr(uid, iid, frame['interaction'], U, N)
    sampler = do.PointSamplerOracle(uid, iid, frame['interaction'], neg_ratio, thr, seed)
    rng = random.Random(seed)                    # RecommenderABC._rng (recommender_abc.py:74)
    p = {k: np.asarray(v, np.float64).copy() for k, v in weights.items()}
    st = co.adam_state(p)
    tmat = np.zeros((U, N), dtype=bool)
    for u in range(U):
        s, e = indptr[u], indptr[u + 1]
        tmat[u, cols[s:e][vals[s:e] >= thr]] = True
    losses = []
    for step in range(epochs):
        batch = sampler.sample(B)
        uids = np.array([b[0] for b in batch])
        t = tmat[uids]
        keep = do.corruption_keep_mask(rng, B, N, q)                     # N draws per row (cdae.py:63)
        xt = (t & keep).astype(np.float64) / (1.0 - q)
        losses.append(co.dense_step(p, st, step, uids, xt, t, lr, reg, loss))
    return p, tmat, losses


@pytest.mark.parametrize('K,loss', [(50, 'bce'), (16, 'mse')])
def test_cdae_fit_matches_oracle(K, loss):
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    frame = _frame()
    ds = InteractionDataset.read_df(frame, verbose=False)
    rng = np.random.default_rng(7)
    U, N = 64, 40
    w = co.init_params(rng, U, N, K, np.float32)
    epochs, B, seed = 12, 32, 10
    model = CDAE(hidden_factors=K, corruption_level=0.2, loss=loss, seed=seed, verbose=False)
    model.fit(ds, epochs=epochs, batch_size=B, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5, initial_weights=w)
    assert model.n_users == U and model.n_items == N
    p, tmat, _ = _oracle_fit(frame, K, epochs, B, seed, w, loss)
    # predict(): raw ids in, full-precision match on every (user, item)
    worst = 0.0
    for u in range(U):
        row = model._predict(u)
        want = co.predict_row(p, u, tmat[u])
        worst = max(worst, float(np.max(np.abs(row - want) / np.abs(want))))
    assert worst < 1e-5, worst
    raw_u, raw_i = frame['user'][0], frame['item'][3]
    uid, iid = ds.user_to_uid(raw_u), ds.item_to_iid(raw_i)
    assert abs(model.predict(raw_u, raw_i) - co.predict_row(p, uid, tmat[uid])[iid]) < 1e-6
    # rank(): heapq.nlargest over (prediction, iid), novelty removes every item the user has a row for
    cand_raw = [ds.iid_to_item(i) for i in range(0, N, 2)] + [10 ** 9]            # one unknown item is skipped
    got = model.rank(raw_u, cand_raw, novelty=True, n=5)
    rows = np.flatnonzero(ds._cols['uid'] == uid)
    seen = set(ds._cols['iid'][rows].tolist())
    want = co.rank_row(co.predict_row(p, uid, tmat[uid]).astype(np.float32), range(0, N, 2), 5, exclude=seen)
    assert [ds.item_to_iid(i) for _, i in got] == [i for _, i in want]
    rec = model.recommend(raw_u, n=7, novelty=False)
    want = co.rank_row(co.predict_row(p, uid, tmat[uid]).astype(np.float32), range(N), 7)
    assert [ds.item_to_iid(i) for _, i in rec] == [i for _, i in want]


def test_cdae_loss_tracking_and_hooks():
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE, MaxValidationValueRule
    frame = _frame()
    ds = InteractionDataset.read_df(frame, verbose=False)
    w = co.init_params(np.random.default_rng(3), 64, 40, 8, np.float32)
    calls = []

    def cb(m):
        calls.append(len(calls))
        return {'val': -abs(len(calls) - 2)}          # best at the 2nd callback (epoch 4)
    model = CDAE(hidden_factors=8, seed=10, verbose=False)
    model.fit(ds, epochs=8, batch_size=16, initial_weights=w, epoch_callback_fn=cb, epoch_callback_freq=2,
              early_stopping_rule=MaxValidationValueRule('val'), early_stopping_freq=2)
    _, _, losses = _oracle_fit(frame, 8, 8, 16, 10, w)
    got = model._loss_tracker.epoch_losses
    assert len(got) == 8 and max(abs(a - b) / b for a, b in zip(got, losses)) < 1e-4
    # weights were reverted to the rule's best epoch (4)
    p4, tmat, _ = _oracle_fit(frame, 8, 4, 16, 10, w)
    row = model._predict(5)
    want = co.predict_row(p4, 5, tmat[5])
    assert float(np.max(np.abs(row - want) / np.abs(want))) < 1e-5
    # API-compat hooks
    batch = model._sampler.sample(4)
    preds, desired = model._predict_batch(batch)
    assert tuple(preds.shape) == (4, 40) and desired.shape == (4, 40)
    assert model._compute_batch_loss(preds, desired) > 0 and model._compute_reg_loss(1e-3, 4) > 0


def test_cdae_sampled_mode_learns():
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    ds = InteractionDataset.read_df(_frame(), verbose=False)
    model = CDAE(hidden_factors=16, mode='sampled', seed=3, verbose=False)
    model.fit(ds, epochs=1, batch_size=256, learning_rate=0.05)
    l0 = model._do_batch(model._sample_batch(256), step=1, want_loss=True)
    for s in range(2, 60):
        model._do_batch(model._sample_batch(256), step=s)
    l1 = model._do_batch(model._sample_batch(256), step=60, want_loss=True)
    assert l1 < l0


def test_cdae_sampled_mode_with_device_sampler_learns():
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    ds = InteractionDataset.read_df(_frame(), verbose=False)
    model = CDAE(hidden_factors=16, mode='sampled', device_sampler=True, seed=3, verbose=False)
    model.fit(ds, epochs=1, batch_size=512, learning_rate=0.05)
    l0 = model._do_batch(model._sample_batch(512), step=1, want_loss=True)
    for s in range(2, 80):
        model._do_batch(model._sample_batch(512), step=s)
    l1 = model._do_batch(model._sample_batch(512), step=80, want_loss=True)
    assert l1 < l0
    u, i, y, ko = model._engine.sample_device(2048, 5, 12345)
    u, i, y = u.cpu().numpy(), i.cpu().numpy(), y.cpu().numpy()
    pairs = set(zip(ds._cols['uid'].tolist(), ds._cols['iid'].tolist()))
    pos = set((a, b) for a, b, v in zip(ds._cols['uid'].tolist(), ds._cols['iid'].tolist(), ds._cols['interaction'].tolist()) if v >= 1e-3)
    # negatives are outside the user's positives, positives inside; roughly neg_ratio/(neg_ratio+1) negatives
    for a, b, t in zip(u.tolist(), i.tolist(), y.tolist()):
        assert ((a, b) in pos) == (t == 1.0)
    assert 0.75 < (y == 0).mean() < 0.9


def test_save_load_and_predict_error_semantics(tmp_path):
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE, RecommenderABC
    frame = _frame()
    ds = InteractionDataset.read_df(frame, verbose=False)
    model = CDAE(hidden_factors=8, seed=10, verbose=False)
    with pytest.raises(AssertionError):
        model.predict(frame['user'][0], frame['item'][0])                  # not fitted yet
    model.fit(ds, epochs=5, batch_size=16)
    raw_u, raw_i = frame['user'][0], frame['item'][0]
    p0 = model.predict(raw_u, raw_i)
    with pytest.raises(AssertionError):
        model.predict(10 ** 9, raw_i)                                       # unknown user
    assert model.predict(10 ** 9, raw_i, skip_errors=True) is None
    # reference quirk kept: an unknown ITEM with skip_errors=True reaches _predict(uid, None) = the whole row (cdae.py:88)
    assert len(model.predict(raw_u, 10 ** 9, skip_errors=True)) == model.n_items
    with pytest.raises(Exception):
        model.rank(raw_u, [10 ** 9], skip_invalid_items=False)
    with pytest.raises(AssertionError):
        model.rank(raw_u, [raw_i], n=5)                                     # n larger than the candidate list
    path = str(tmp_path / 'cdae.bin')
    model.save(path)
    again = RecommenderABC.load(path)
    assert abs(again.predict(raw_u, raw_i) - p0) < 1e-7
    a = model.recommend(raw_u, n=6, novelty=True)
    b = again.recommend(raw_u, n=6, novelty=True)
    assert [i for _, i in a] == [i for _, i in b]


def test_rank_is_thread_safe():
    """The reference evaluators call model.rank() from a 4-thread pool (ranking_evaluation.py:107-113)."""
    from concurrent.futures import ThreadPoolExecutor
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    frame = _frame()
    ds = InteractionDataset.read_df(frame, verbose=False)
    model = CDAE(hidden_factors=16, seed=1, verbose=False)
    model.fit(ds, epochs=10, batch_size=32)
    users = [ds.uid_to_user(u) for u in range(model.n_users)]
    items = [ds.iid_to_item(i) for i in range(model.n_items)]
    serial = [model.rank(u, items, novelty=True, n=7) for u in users]
    with ThreadPoolExecutor(max_workers=4) as pool:
        threaded = list(pool.map(lambda u: model.rank(u, items, novelty=True, n=7), users * 3))
    assert threaded == serial * 3


def test_pipelined_sampled_fit_equals_the_inline_sequence():
    """CDAE.fit(mode='sampled', device_sampler=True) runs through SampledPipeline (sampler two batches ahead, touch list one
    ahead, side stream); the parameters must be bit-identical to sampling, indexing and stepping inline with the same seeds."""
    import torch
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    from drecpy_amd.engine import CdaeEngine
    ds = InteractionDataset.read_df(_frame(), verbose=False)
    model = CDAE(hidden_factors=16, mode='sampled', device_sampler=True, seed=3, verbose=False)
    model.fit(ds, epochs=25, batch_size=384, learning_rate=0.05, reg_rate=1e-3, neg_ratio=5)
    torch.cuda.synchronize()
    got = [t.clone() for t in model._engine.tables()]
    eng = CdaeEngine(model.n_users, model.n_items, 16)
    eng.init_glorot(3)
    eng.set_history(model._hist_indptr, model._hist_indices)
    eng.set_recorded_pairs(*ds.interaction_csr()[:2])       # (the frame records pairs below the threshold: negatives avoid those too)
    eng.init_optimizer('adagrad', 0.05, 1e-3)
    ms = model._mask_seed
    for s in range(25):
        uid, iid, y, ko = eng.sample_device(384, 5, ms * 7919 + s + 1)
        bt, alive = eng.make_batch(uid, iid, y, keep_off=ko, q=0.2, mask_seed=ms + 0x9E3779B9 * (s + 1))
        eng.step_sparse(s, bt, 'bce')
    torch.cuda.synchronize()
    for a, b in zip(got, eng.tables()):
        assert torch.equal(a, b)


def test_second_fit_on_another_dataset_drops_the_first_ones_caches():
    """per-model caches derived from the dataset (user -> items table of rank(), epoch snapshots) must not survive
    into a second fit() on different data."""
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    rng = np.random.default_rng(0)

    def frame(U, N, n, seed):
        r = np.random.default_rng(seed)
        return InteractionDataset.read_df({'user': r.integers(0, U, n), 'item': r.integers(0, N, n), 'interaction': r.integers(1, 6, n)},
                                          verbose=False)
    a, b = frame(30, 50, 600, 1), frame(45, 80, 900, 2)
    m = CDAE(hidden_factors=8, seed=3, verbose=False)
    m.fit(a, epochs=3, batch_size=16)
    u = a.uid_to_user(0)
    m.rank(u, [a.iid_to_item(i) for i in range(50)], novelty=True, n=5)          # builds the user -> items table of dataset a
    m.epoch_weights[2] = 'stale'
    m.fit(b, epochs=3, batch_size=16)
    assert m.epoch_weights == {} and len(m.trainable_weights) == 5
    for uid in (0, 44):
        raw = b.uid_to_user(uid)
        got = m.rank(raw, [b.iid_to_item(i) for i in range(80)], novelty=True, n=80)
        seen = set(b.select(f'uid == {uid}').values_list('iid', to_list=True))
        assert {b.item_to_iid(i) for _, i in got} == set(range(80)) - seen       # the novelty mask is dataset b's


@pytest.mark.parametrize('epochs', [1, 3, 57])
def test_quiet_fit_in_the_library_equals_the_step_by_step_loop(epochs):
    """fit() with nobody watching runs its loop inside libdrx (drx_cdae_fit_dense): bit-identical tables, and the sampler and
    corruption streams left exactly where the Python loop leaves them (the next fit continues identically)."""
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    ds = InteractionDataset.read_df(_frame(), verbose=False)
    w = co.init_params(np.random.default_rng(5), 64, 40, 24, np.float32)
    out = []
    for fused in (True, False):
        m = CDAE(hidden_factors=24, corruption_level=0.3, seed=21, verbose=False)
        m.fused_fit = fused
        m.fit(ds, epochs=epochs, batch_size=16, learning_rate=2e-3, reg_rate=1e-3, neg_ratio=3, initial_weights=w)
        state = (m._draw_ticket, m._mask_pos, list(m._mask_at))
        nxt = m._sampler.sample(5)                      # where the sampler stream stands
        keep = m._corruption_keep(np.array([1, 2, 3], np.int32))     # where the corruption stream stands
        e = m._engine
        out.append(([t.cpu().numpy().copy() for t in (e.W, e.W2T, e.V, e.b, e.b2)], state, nxt, [np.asarray(k).copy() for k in keep]))
    (pa, sa, na, ka), (pb, sb, nb, kb) = out
    assert sa[0] == sb[0] == epochs and sa[1] == sb[1]
    assert na == nb and all(np.array_equal(x, y) for x, y in zip(ka, kb))
    for x, y in zip(pa, pb):
        assert np.array_equal(x, y)


def test_quiet_fit_is_left_to_python_when_a_hook_is_replaced():
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    ds = InteractionDataset.read_df(_frame(), verbose=False)
    seen = []

    class Watching(CDAE):
        def _do_batch(self, batch_samples, step=0, **kwds):
            seen.append(step)
            return super()._do_batch(batch_samples, step=step, **kwds)
    Watching(hidden_factors=8, seed=1, verbose=False).fit(ds, epochs=5, batch_size=8)
    assert seen == [0, 1, 2, 3, 4]
    m = CDAE(hidden_factors=8, seed=1, verbose=False)
    orig = m._sample_batch
    m._sample_batch = lambda *a, **k: (seen.append('s'), orig(*a, **k))[1]
    m.fit(ds, epochs=2, batch_size=8)
    assert seen[-2:] == ['s', 's']


@pytest.mark.parametrize('implicit', [True, False])
def test_device_point_sampler_has_the_reference_samplers_distribution(implicit):
    """drx_point_sample (the throughput mode's triples) against the reference-exact PointSampler stream over many draws: the share of
    negatives, the users of positives (uniform USER, point_sampler.py:44-61), the (user, item) cells of positives and the users of
    negatives are distributed alike (two-sample chi-square per degree of freedom ~ 1)."""
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import CDAE
    from drecpy_amd.Sampler import PointSampler
    # implicit: every recorded pair is a positive.  Otherwise the frame also records pairs BELOW the interaction threshold: the reference
    # draws its negatives among the pairs ABSENT from the frame (point_sampler.py:56), and so does the device sampler once it is given
    # the CSR of all recorded pairs (CDAE._pre_fit: engine.set_recorded_pairs)
    frame = _frame()
    if implicit:
        keep = frame['interaction'] >= 1
        frame = {k: v[keep] for k, v in frame.items()}
    ds = InteractionDataset.read_df(frame, verbose=False)
    model = CDAE(hidden_factors=8, mode='sampled', device_sampler=True, seed=3, verbose=False)
    model.fit(ds, epochs=1, batch_size=64, learning_rate=0.05)
    n = 60000
    u, i, y, _ = model._engine.sample_device(n, 5, 99)
    u, i, y = u.cpu().numpy().astype(np.int64), i.cpu().numpy().astype(np.int64), y.cpu().numpy()
    ru, ri, rv, rneg = PointSampler(ds, 5, 1e-3, 11).sample_arrays(n)
    ru, ri, rneg = np.asarray(ru, np.int64), np.asarray(ri, np.int64), np.asarray(rneg).astype(bool)
    U, N = model.n_users, model.n_items
    assert abs((y == 0).mean() - rneg.mean()) < 0.01

    def close(x, ycol, bins, what):
        hx, hy = np.bincount(x, minlength=bins).astype(float), np.bincount(ycol, minlength=bins).astype(float)
        hx, hy = hx * (hy.sum() / hx.sum()), hy                       # (the two label shares differ by a fraction of a percent)
        live = (hx + hy) > 0
        chi = float((((hx - hy) ** 2) / (hx + hy))[live].sum() / max(1, live.sum() - 1))
        assert chi < 1.6, (what, chi)
    pos_d, pos_r = y == 1, ~rneg
    close(u[pos_d], ru[pos_r], U, 'users of positives')
    close(u[pos_d] * N + i[pos_d], ru[pos_r] * N + ri[pos_r], U * N, 'cells of positives')
    close(u[~pos_d], ru[~pos_r], U, 'users of negatives')
    close(i[~pos_d], ri[~pos_r], N, 'items of negatives')


@pytest.mark.parametrize('device_sampler', [False, True])
def test_dmf_and_caser_survive_save_and_load(tmp_path, device_sampler):
    """RecommenderABC.save / load (recommender_abc.py:503-524 dumps and restores the whole object) for the two models besides CDAE: the
    loaded model ranks like the fitted one — also after fit(device_sampler=True), whose run-ahead streams and events are not saved."""
    import pandas as pd
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import DMF, Caser, RecommenderABC
    rng = np.random.default_rng(2)
    rows = []
    for u in range(40):
        for t, i in enumerate(rng.choice(60, size=14, replace=False)):
            rows.append((u + 1, int(i) + 1, int(rng.integers(1, 6)), t))
    frame = pd.DataFrame(rows, columns=['user', 'item', 'interaction', 'timestamp'])
    ds = InteractionDataset.read_df(frame, verbose=False)
    dmf = DMF(user_factors=[16, 8], item_factors=[16, 8], seed=10, verbose=False)
    dmf.fit(ds, epochs=6, batch_size=64, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=3, device_sampler=device_sampler)
    caser = Caser(L=3, T=2, d=8, n_v=2, n_h=4, dropout_rate=0.5, sort_column='timestamp', seed=10, verbose=False)
    caser.fit(ds, epochs=6, batch_size=32, learning_rate=5e-3, reg_rate=1e-5, neg_ratio=2, device_sampler=device_sampler)
    for name, model in (('dmf', dmf), ('caser', caser)):
        path = str(tmp_path / f'{name}.bin')
        model.save(path)
        again = RecommenderABC.load(path)
        for u in (1, 7, 40):
            a = model.recommend(u, n=5, novelty=True)
            b = again.recommend(u, n=5, novelty=True)
            assert [i for _, i in a] == [i for _, i in b], (name, u)
            np.testing.assert_allclose([s for s, _ in a], [s for s, _ in b], rtol=0, atol=1e-6)


@pytest.mark.parametrize('device_sampler', [False, True])
def test_dmf_and_caser_epoch_callbacks_early_stopping_and_revert(device_sampler):
    """recommender_abc.py:170-256 for the two models besides CDAE: the callback runs every `epoch_callback_freq` epochs, the rule every
    `early_stopping_freq`, and after training the weights are those of the rule's best epoch — equal to a fresh fit of that many epochs
    from the same seed (the draws and the steps are deterministic), with the host samplers and with the device samplers."""
    import pandas as pd
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import DMF, Caser, MaxValidationValueRule
    rng = np.random.default_rng(4)
    rows = []
    for u in range(40):
        for t, i in enumerate(rng.choice(60, size=14, replace=False)):
            rows.append((u + 1, int(i) + 1, int(rng.integers(1, 6)), t))
    frame = pd.DataFrame(rows, columns=['user', 'item', 'interaction', 'timestamp'])
    ds = InteractionDataset.read_df(frame, verbose=False)
    make = {'dmf': lambda: DMF(user_factors=[16, 8], item_factors=[16, 8], seed=10, verbose=False),
            'caser': lambda: Caser(L=3, T=2, d=8, n_v=2, n_h=4, dropout_rate=0.5, sort_column='timestamp', seed=10, verbose=False)}
    args = {'dmf': dict(batch_size=64, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=3),
            'caser': dict(batch_size=32, learning_rate=5e-3, reg_rate=1e-5, neg_ratio=2)}
    for name in ('dmf', 'caser'):
        calls = []

        def cb(m):
            calls.append(1)
            return {'val': -abs(len(calls) - 2)}          # best at the 2nd callback = epoch 4
        model = make[name]()
        model.fit(ds, epochs=8, epoch_callback_fn=cb, epoch_callback_freq=2, early_stopping_rule=MaxValidationValueRule('val'),
                  early_stopping_freq=2, device_sampler=device_sampler, **args[name])
        assert len(calls) == 4 and len(model._loss_tracker.epoch_losses) == 8, (name, len(calls))
        ref = make[name]()
        ref.fit(ds, epochs=4, device_sampler=device_sampler, **args[name])
        a, b = model._engine.get_params(), ref._engine.get_params()
        for k in a:
            np.testing.assert_allclose(a[k], b[k], rtol=0, atol=1e-6, err_msg=f'{name} {k}')
