"""GPU parity of the Caser step (drx_caser_* + drx_scatter_rows + drx_adam_*) against oracle/caser_oracle.py."""
import numpy as np
import pytest

from oracle import caser_oracle as ca

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / np.maximum(np.abs(b), 1e-12)))


@pytest.mark.parametrize('update', ['csr', 'scatter'])      # the two ways the lookup tables are updated (engine_caser.py)
@pytest.mark.parametrize('d,L,n_v,n_h,T,neg,B,drop', [(50, 5, 4, 16, 3, 3, 64, False), (16, 3, 2, 8, 2, 2, 37, True),
                                                      (64, 5, 4, 16, 3, 3, 300, True)])
def test_caser_steps_match_oracle(d, L, n_v, n_h, T, neg, B, drop, update):
    """(50, 5, 4, 16, 3, 3) is the training kernel's compile-time instantiation (examples/caser.py), (64, 5, ...) keeps only the
    convolution weights in LDS; more geometries of the tile kernel: test_caser_tile_geometries below."""
    _steps_match_oracle(d, L, n_v, n_h, T, neg, B, drop, update)


# the tile kernel's other paths: no weights in LDS (L = 8, d = 32: 96 KB of small weights beside 108 KB of tile scratch), two 16-filter
# tiles per horizontal convolution (n_h = 24), two for the vertical one (n_v = 18), more than 16 targets per sample (the rows of the
# first 16 are fetched ahead, the rest where they are used), several tiles per workgroup (B = 4200 > 256 tiles of 16)
@pytest.mark.parametrize('d,L,n_v,n_h,T,neg,B,drop', [(32, 8, 4, 16, 2, 3, 100, True), (20, 4, 3, 24, 2, 2, 50, True),
                                                      (16, 3, 18, 8, 2, 2, 33, False), (24, 5, 4, 16, 3, 6, 41, True),
                                                      (16, 3, 2, 8, 1, 2, 4200, True),
                                                      (5, 1, 1, 1, 1, 0, 19, False), (7, 2, 3, 5, 1, 1, 16, True)])     # (the smallest shapes)
def test_caser_tile_geometries(d, L, n_v, n_h, T, neg, B, drop):
    _steps_match_oracle(d, L, n_v, n_h, T, neg, B, drop, 'csr', steps=2 if B > 1000 else 3)


def _steps_match_oracle(d, L, n_v, n_h, T, neg, B, drop, update, steps=3):
    from drecpy_amd.engine_caser import CaserEngine
    rng = np.random.default_rng(d + B)
    U, N = 40, 150
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    eng = CaserEngine(U, N, L, T, neg, d, n_v, n_h)
    eng.table_update = update
    eng.set_params(p)
    eng.lr, eng.reg = 5e-3, 1e-4
    st = ca.adam_state(p)
    g0 = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g0[k], p[k].astype(np.float32), rtol=0, atol=0)
    nx = n_v + L * n_h
    for step in range(2 * steps):
        uids = rng.integers(0, U, size=B)
        before = rng.integers(0, N, size=(B, L))
        after = rng.integers(0, N, size=(B, T + T * neg))
        keep = (rng.random((B, nx)) >= 0.5) if drop else None
        rate = 0.5 if drop else 0.0
        lo = ca.step(p, st, step, uids, before, after, T, 5e-3, 1e-4, keep, rate)
        lg = eng.step(step, uids, before, after, keep, rate, want_loss=True)
        assert abs(lg - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)
    # inference: scores of all items from the last L items (caser.py:128-137)
    uids = rng.integers(0, U, size=5)
    before = rng.integers(0, N, size=(5, L))
    sc = eng.scores_all(uids, before).cpu().numpy()
    for r in range(5):
        want = ca.rank_scores(p, uids[r], before[r])
        assert np.max(np.abs(sc[r] - want)) < 2e-5 * max(1.0, np.max(np.abs(want)))


@pytest.mark.parametrize('d,L,n_v,n_h,T,neg,B,drop', [(100, 9, 4, 16, 3, 3, 64, True), (128, 16, 2, 8, 2, 2, 37, False), (72, 5, 4, 16, 3, 3, 50, True)])
def test_torch_checker_matches_oracle_also_beyond_the_kernels_domain(d, L, n_v, n_h, T, neg, B, drop):
    """The second checker (tests/caser_torch_checker.py: the forward of caser.py:97-120 in torch operations, torch.autograd where the
    reference has tf.GradientTape, the library's Keras-Adam kernel per registered layer) against the NumPy oracle — two independent
    statements of the reference's step agree, also at shapes the HIP kernel does not take (the product rejects those)."""
    from caser_torch_checker import CaserTorchChecker as CaserWideEngine
    rng = np.random.default_rng(d + B)
    U, N = 40, 150
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    eng = CaserWideEngine(U, N, L, T, neg, d, n_v, n_h)
    eng.set_params(p)
    eng.lr, eng.reg = 5e-3, 1e-4
    st = ca.adam_state(p)
    nx = n_v + L * n_h
    for step in range(5):
        uids = rng.integers(0, U, size=B)
        before = rng.integers(0, N, size=(B, L))
        after = rng.integers(0, N, size=(B, T + T * neg))
        keep = (rng.random((B, nx)) >= 0.5) if drop else None
        rate = 0.5 if drop else 0.0
        lo = ca.step(p, st, step, uids, before, after, T, 5e-3, 1e-4, keep, rate)
        lg = eng.step(step, uids, before, after, keep, rate, want_loss=True)
        assert abs(lg - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)
    uids = rng.integers(0, U, size=5)
    before = rng.integers(0, N, size=(5, L))
    sc = eng.scores_all(uids, before).cpu().numpy()
    for r in range(5):
        want = ca.rank_scores(p, uids[r], before[r])
        assert np.max(np.abs(sc[r] - want)) < 2e-5 * max(1.0, np.max(np.abs(want)))


def test_public_class_rejects_shapes_without_a_hip_kernel():
    """Caser(L, d) outside 1 <= L <= 8, 1 <= d <= 64 and L > n_h (where caser.py:108's squeeze fails in the reference) are rejected
    in the constructor: no second backend behind the class."""
    from drecpy_amd.Recommender import Caser
    for kw in (dict(L=9, d=50), dict(L=5, d=100), dict(L=65, d=16), dict(L=3, d=128, n_h=4)):
        with pytest.raises(Exception, match='supports 1 <= L <= 8 and 1 <= d <= 64'):
            Caser(verbose=False, **kw)
    with pytest.raises(Exception, match='needs L <= n_h'):
        Caser(L=5, d=16, n_h=4, verbose=False)
    Caser(L=8, d=64, n_h=8, verbose=False)
    import drecpy_amd
    import glob
    import os
    pkg = os.path.dirname(drecpy_amd.__file__)
    for f in glob.glob(os.path.join(pkg, '*.py')) + glob.glob(os.path.join(pkg, 'Recommender', '*.py')):
        src = open(f).read()
        if os.path.basename(f) == 'recommender_abc.py':        # the tape step for USER-written hooks (the plugin API's autodiff slot)
            continue
        assert 'torch.einsum' not in src and 'autograd' not in src and '.backward(' not in src, f


@pytest.mark.parametrize('d,L,n_v,n_h,T,neg,B', [(50, 5, 4, 16, 3, 3, 64), (64, 8, 4, 8, 2, 2, 48)])
def test_hip_step_matches_the_torch_checker(d, L, n_v, n_h, T, neg, B):
    """The HIP training step (k_caser_tile) against the torch-autograd statement of the same step on the same inputs: parameters after
    three steps and the inference scores agree to fp32 accuracy (both are fp32; the NumPy oracle is the fp64 judge of either)."""
    from caser_torch_checker import CaserTorchChecker
    from drecpy_amd.engine_caser import CaserEngine
    rng = np.random.default_rng(d + L)
    U, N = 40, 150
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    eng, chk = CaserEngine(U, N, L, T, neg, d, n_v, n_h), CaserTorchChecker(U, N, L, T, neg, d, n_v, n_h)
    for e in (eng, chk):
        e.set_params(p)
        e.lr, e.reg = 5e-3, 1e-4
    nx = n_v + L * n_h
    for step in range(3):
        uids = rng.integers(0, U, size=B)
        before = rng.integers(0, N, size=(B, L))
        after = rng.integers(0, N, size=(B, T + T * neg))
        keep = rng.random((B, nx)) >= 0.5
        la = eng.step(step, uids, before, after, keep, 0.5, want_loss=True)
        lb = chk.step(step, uids, before, after, keep, 0.5, want_loss=True)
        assert abs(la - lb) / abs(lb) < 2e-5, (step, la, lb)
    a, b = eng.get_params(), chk.get_params()
    for k in b:
        np.testing.assert_allclose(a[k].reshape(b[k].shape), b[k], rtol=0, atol=2e-5, err_msg=k)


def test_caser_fit_matches_oracle_end_to_end():
    """Caser.fit() with the reference-exact ListSampler stream and injected weights / dropout masks vs the oracle."""
    from helpers import load_frames
    from oracle import data_oracle as do
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import Caser
    frame = {k: v.copy() for k, v in load_frames()['ls_int_ts'].items()}
    ds = InteractionDataset.read_df(frame, verbose=False)
    L, T, d, n_v, n_h, neg, B, epochs, seed = 5, 3, 16, 2, 8, 2, 24, 5, 10
    uid, _ = do.first_appearance_codes(frame['user'].tolist())
    iid, _ = do.first_appearance_codes(frame['item'].tolist())
    U, N = int(uid.max()) + 1, int(iid.max()) + 1
    p = ca.init_params(np.random.default_rng(1), U, N, L, d, n_v, n_h, np.float32)
    nx = n_v + L * n_h
    masks = [np.random.default_rng(100 + s).random((B, nx)) >= 0.5 for s in range(epochs)]
    model = Caser(L=L, T=T, d=d, n_v=n_v, n_h=n_h, dropout_rate=0.5, seed=seed, verbose=False)
    model.fit(ds, epochs=epochs, batch_size=B, learning_rate=5e-3, reg_rate=1e-5, neg_ratio=neg, initial_weights=p,
              dropout_mask_fn=lambda s, b, n: masks[s])
    # oracle side: reference ListSampler semantics restated on stdlib random
    f = dict(frame)
    f['uid'], f['iid'] = uid.astype(np.int16), iid.astype(np.int16)
    smp = do.ListSamplerOracle(f, 'uid', neg_ratio=neg, n_targets=T, interaction_threshold=1e-3, sort_column='timestamp',
                               min_positive_records=L, max_positive_records=L, seed=seed)
    po = {k: v.astype(np.float64) for k, v in p.items()}
    st = ca.adam_state(po)
    for s in range(epochs):
        recs = smp.sample_group_records(B)
        uids = np.array([int(f['uid'][b[0]]) for b, _, _ in recs])
        before = np.array([[int(f['iid'][r]) for r in b] for b, _, _ in recs])
        after = np.array([[int(f['iid'][r]) for r in a] + ng for _, a, ng in recs])
        ca.step(po, st, s, uids, before, after, T, 5e-3, 1e-5, masks[s], 0.5)
    g = model._engine.get_params()
    for k in po:
        np.testing.assert_allclose(g[k], po[k], rtol=0, atol=3e-5, err_msg=k)
    # rank(): all-item scores from the user's last L items, novelty removes the user's own items
    u0 = 3
    rows = np.flatnonzero(uid == u0)
    seq = iid[rows][np.argsort(frame['timestamp'][rows], kind='stable')]
    want_sc = ca.rank_scores(po, u0, seq[-L:])
    raw_items = [ds.iid_to_item(i) for i in range(0, N, 3)]
    got = model.rank(ds.uid_to_user(u0), raw_items, novelty=True, n=8)
    from oracle import cdae_oracle as co
    want = co.rank_row(want_sc.astype(np.float32), range(0, N, 3), 8, exclude=set(seq.tolist()))
    assert [ds.item_to_iid(i) for _, i in got] == [i for _, i in want]


def test_caser_counter_based_dropout_mask_matches_oracle():
    """keep = None, rate > 0: the kernel evaluates the mask drx_hash_u32(mask_seed, b, j) >= rate * 2^32 itself; the same mask
    rebuilt on the host (helpers.hash_u32) and injected into the oracle gives the same step."""
    from helpers import hash_u32, q_threshold
    from drecpy_amd.engine_caser import CaserEngine
    rng = np.random.default_rng(5)
    U, N, L, T, neg, d, n_v, n_h, B = 30, 80, 5, 3, 2, 24, 4, 16, 70
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    eng = CaserEngine(U, N, L, T, neg, d, n_v, n_h)
    eng.set_params(p)
    eng.lr, eng.reg = 5e-3, 1e-4
    st = ca.adam_state(p)
    nx = n_v + L * n_h
    for step in range(3):
        uids = rng.integers(0, U, size=B)
        before = rng.integers(0, N, size=(B, L))
        after = rng.integers(0, N, size=(B, T + T * neg))
        seed = 0xABCDEF12345 + step
        bb, jj = np.meshgrid(np.arange(B), np.arange(nx), indexing='ij')
        keep = (hash_u32(seed, bb.ravel(), jj.ravel()) >= q_threshold(0.5)).reshape(B, nx)
        assert 0.4 < keep.mean() < 0.6
        lo = ca.step(p, st, step, uids, before, after, T, 5e-3, 1e-4, keep, 0.5)
        lg = eng.step(step, uids, before, after, None, 0.5, want_loss=True, mask_seed=seed)
        assert abs(lg - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)


@pytest.mark.parametrize('act_h,act_mlp', [('tanh', 'sigmoid'), ('sigmoid', 'linear'), ('linear', 'tanh')])
def test_caser_activations_match_oracle(act_h, act_mlp):
    """act_h / act_mlp of caser.py:29-30 beyond the relu default."""
    from drecpy_amd.engine_caser import CaserEngine
    rng = np.random.default_rng(len(act_h) * 31 + len(act_mlp))
    U, N, L, T, neg, d, n_v, n_h, B = 25, 60, 4, 2, 2, 20, 3, 8, 41
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    for k in p:
        if k.endswith('_b'):
            p[k] = rng.normal(0, 0.1, size=p[k].shape)
    eng = CaserEngine(U, N, L, T, neg, d, n_v, n_h, act_h=act_h, act_mlp=act_mlp)
    eng.set_params(p)
    eng.lr, eng.reg = 5e-3, 1e-4
    st = ca.adam_state(p)
    nx = n_v + L * n_h
    for step in range(4):
        uids = rng.integers(0, U, size=B)
        before = rng.integers(0, N, size=(B, L))
        after = rng.integers(0, N, size=(B, T + T * neg))
        keep = rng.random((B, nx)) >= 0.3
        lo = ca.step(p, st, step, uids, before, after, T, 5e-3, 1e-4, keep, 0.3, act_h, act_mlp)
        lg = eng.step(step, uids, before, after, keep, 0.3, want_loss=True)
        assert abs(lg - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)
    from drecpy_amd.Recommender import Caser, DMF
    with pytest.raises(Exception, match='supports the activations'):
        Caser(act_h='gelu')
    with pytest.raises(Exception, match='1 <= L <= 8 and 1 <= d <= 64'):
        Caser(d=2048)
    with pytest.raises(Exception, match='towers of 1..4 layers of width 1..128'):
        DMF(user_factors=[256, 64], item_factors=[64])


@pytest.mark.parametrize('update', ['csr', 'scatter'])
def test_caser_steps_with_repeated_lookups_and_single_sample(update):
    """Edge cases of the lookup-table paths: one user in every sample, one item in every lookup of the batch (a table row that collects
    hundreds of gradient rows while the others collect none), a window that names the same item L times, a batch of one — vs the oracle."""
    from drecpy_amd.engine_caser import CaserEngine
    rng = np.random.default_rng(5)
    U, N, L, T, neg, d, n_v, n_h = 12, 30, 4, 2, 2, 20, 2, 8
    Tp = T + T * neg
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    eng = CaserEngine(U, N, L, T, neg, d, n_v, n_h)
    eng.table_update = update
    eng.set_params(p)
    eng.lr, eng.reg = 5e-3, 1e-4
    st = ca.adam_state(p)
    B = 40
    batches = [(np.full(B, 7), rng.integers(0, N, size=(B, L)), rng.integers(0, N, size=(B, Tp))),            # one user
               (rng.integers(0, U, size=B), np.full((B, L), 11), np.full((B, Tp), 11)),                        # one item everywhere
               (rng.integers(0, U, size=B), np.repeat(rng.integers(0, N, size=(B, 1)), L, axis=1), rng.integers(0, N, size=(B, Tp))),
               (np.array([3]), rng.integers(0, N, size=(1, L)), rng.integers(0, N, size=(1, Tp))),             # a batch of one
               (rng.integers(0, U, size=B), rng.integers(0, N, size=(B, L)), rng.integers(0, N, size=(B, Tp)))]
    for step, (uids, before, after) in enumerate(batches):
        lo = ca.step(p, st, step, uids, before, after, T, 5e-3, 1e-4, None, 0.0)
        lg = eng.step(step, uids, before, after, None, 0.0, want_loss=True)
        assert abs(lg - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)


def _caser_frame():
    from helpers import load_frames
    return {k: v.copy() for k, v in load_frames()['ls_int_ts'].items()}


@pytest.mark.parametrize('L,T,neg,n,seed', [(5, 3, 3, 600, 12345), (2, 1, 6, 257, 7), (3, 2, 0, 64, 99),
                                            (2, 2, 12, 300, 5), (2, 1, 40, 130, 8), (2, 2, 32, 70, 6), (2, 2, 40, 64, 3)])
def test_device_list_sampler_equals_its_cpu_restatement(L, T, neg, n, seed):
    """drx_list_sample_device (throughput mode of ListSampler) against oracle/data_oracle.py::list_sample_counter, bit for bit.
    The kernel with sixteen lanes per window takes up to 64 negatives per window — 9, 6, 0: one chunk of 16; 24, 40, 64: several —, the
    thread-per-window kernel the rest (80)."""
    from oracle import data_oracle as do
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Sampler import ListSampler
    ds = InteractionDataset.read_df(_caser_frame(), verbose=False)
    ds.assign_internal_ids()
    s = ListSampler(ds, ['uid'], neg_ratio=neg, n_targets=T, interaction_threshold=1e-3, negative_ids_col='iid', min_positive_records=L,
                    max_positive_records=L, sort_column='timestamp', seed=10)
    s.device_twin('cuda:0')
    g, b, a = s.sample_device(n, seed)
    wg, wb, wa = do.list_sample_counter(s.twin_host_arrays(), n, L, T, neg, seed)
    assert np.array_equal(g.cpu().numpy(), wg) and np.array_equal(b.cpu().numpy(), wb) and np.array_equal(a.cpu().numpy(), wa)


def test_device_list_sampler_on_a_catalogue_barely_larger_than_the_draw():
    """Six ids, a group holds three of them, three negatives wanted: every draw collides often and the fall-back walk runs."""
    from oracle import data_oracle as do
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Sampler import ListSampler
    rng = np.random.default_rng(3)
    users, items = [], []
    for u in range(40):
        users += [u] * 3
        items += rng.choice(6, size=3, replace=False).tolist()
    frame = {'user': np.array(users), 'item': np.array(items), 'interaction': np.ones(len(users)), 'timestamp': np.arange(len(users))}
    ds = InteractionDataset.read_df(frame, verbose=False)
    ds.assign_internal_ids()
    s = ListSampler(ds, ['uid'], neg_ratio=3, n_targets=1, interaction_threshold=1e-3, negative_ids_col='iid', min_positive_records=2,
                    max_positive_records=2, sort_column='timestamp', seed=1)
    s.device_twin('cuda:0')
    g, b, a = s.sample_device(500, 31)
    wg, wb, wa = do.list_sample_counter(s.twin_host_arrays(), 500, 2, 1, 3, 31)
    assert np.array_equal(a.cpu().numpy(), wa) and np.array_equal(b.cpu().numpy(), wb) and np.array_equal(g.cpu().numpy(), wg)
    negs = a.cpu().numpy()[:, 1:]
    assert all(len(set(r.tolist())) == 3 for r in negs)


def test_caser_fit_with_the_device_sampler_matches_the_oracle_fed_the_same_draws():
    """Caser.fit(device_sampler=True): windows from drx_list_sample_device, dropout mask from the counter-based hash — the oracle
    stepping on the CPU restatement of both reaches the same weights."""
    from helpers import hash_u32, q_threshold
    from oracle import data_oracle as do
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import Caser
    frame = _caser_frame()
    ds = InteractionDataset.read_df(frame, verbose=False)
    L, T, d, n_v, n_h, neg, B, epochs, seed = 5, 3, 16, 2, 8, 2, 48, 4, 10
    uid, _ = do.first_appearance_codes(frame['user'].tolist())
    iid, _ = do.first_appearance_codes(frame['item'].tolist())
    U, N = int(uid.max()) + 1, int(iid.max()) + 1
    p = ca.init_params(np.random.default_rng(1), U, N, L, d, n_v, n_h, np.float32)
    model = Caser(L=L, T=T, d=d, n_v=n_v, n_h=n_h, dropout_rate=0.5, seed=seed, verbose=False)
    model.fit(ds, epochs=epochs, batch_size=B, learning_rate=5e-3, reg_rate=1e-5, neg_ratio=neg, initial_weights=p, device_sampler=True)
    assert model._sampler_kind.startswith('device')
    twin = model._sampler.twin_host_arrays()
    po = {k: v.astype(np.float64) for k, v in p.items()}
    st = ca.adam_state(po)
    nx = n_v + L * n_h
    M = (1 << 64) - 1
    for s in range(epochs):
        uids, before, after = do.list_sample_counter(twin, B, L, T, neg, (seed * 1000003 + s + 1) & M)
        bb, jj = np.meshgrid(np.arange(B), np.arange(nx), indexing='ij')
        keep = (hash_u32((seed * 0x9E3779B97F4A7C15 + s + 1) & M, bb.ravel(), jj.ravel()) >= q_threshold(0.5)).reshape(B, nx)
        ca.step(po, st, s, uids, before, after, T, 5e-3, 1e-5, keep, 0.5)
    g = model._engine.get_params()
    for k in po:
        np.testing.assert_allclose(g[k], po[k], rtol=0, atol=3e-5, err_msg=k)


def test_fused_launches_equal_the_separate_ones_bit_for_bit():
    """drx_rows_csr_adam_multi == drx_rows_csr_adam / drx_rows_csr_adam_outer table by table, and drx_caser_step_small ==
    drx_caser_fwd_bwd + drx_adam_segments: the fused entry points run the same operations in the same order."""
    import ctypes as C
    import torch
    from drecpy_amd import _lib
    from drecpy_amd._lib import check, lib, stream_ptr
    dev = torch.device('cuda:0')
    L_ = lib()
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    rnd = lambda *s: torch.rand(*s, generator=gen, device=dev) - 0.5
    st = stream_ptr(dev)
    # (table rows, ld, lookups, group); the last one's rows are named by ~150 lookups each: their sums are split over the workgroup
    specs = [(300, 52, 900, 0), (120, 100, 840, 12), (50, 16, 64, 0), (20, 52, 3000, 0)]
    tabs = (_lib.CsrAdamTable * 4)()
    keep, single = [], []
    for t, (n_rows, ld, T, group) in zip(tabs, specs):
        keys = torch.randint(0, n_rows, (T,), generator=gen, device=dev)
        order = torch.argsort(keys, stable=True).to(torch.int32)
        ptr = torch.zeros(n_rows + 1, dtype=torch.int32, device=dev)
        ptr[1:] = torch.cumsum(torch.bincount(keys, minlength=n_rows), 0).to(torch.int32)
        src = rnd(T // group if group else T, ld)
        scale = rnd(T) if (group or ld == 16) else None
        state = [rnd(n_rows, ld), rnd(n_rows, ld), rnd(n_rows, ld).abs()]
        sstate = [rnd(n_rows), rnd(n_rows), rnd(n_rows).abs()] if scale is not None else [None] * 3
        copies = [x.clone() for x in state] + [x.clone() if x is not None else None for x in sstate]
        keep += [order, ptr, src, scale, state, sstate, copies]
        ps = lambda x: x.data_ptr() if x is not None else None
        t.row_ptr, t.order, t.src, t.scale, t.group, t.ld, t.n_rows = ptr.data_ptr(), order.data_ptr(), src.data_ptr(), ps(scale), group, ld, n_rows
        t.p, t.m, t.v, t.p_s, t.m_s, t.v_s = [x.data_ptr() for x in state] + [ps(x) for x in sstate]
        t.alpha, t.alpha_s, t.l2_coef = 1e-2, 2e-2, 1e-3
        c = copies
        if group:
            check(L_.drx_rows_csr_adam_outer(ptr.data_ptr(), order.data_ptr(), scale.data_ptr(), src.data_ptr(), group, ld, n_rows, c[0].data_ptr(),
                                             c[1].data_ptr(), c[2].data_ptr(), ps(c[3]), ps(c[4]), ps(c[5]), 1e-2, 2e-2, 1e-3, 0.9, 0.999, 1e-7, st), 'outer')
        else:
            check(L_.drx_rows_csr_adam(ptr.data_ptr(), order.data_ptr(), src.data_ptr(), ps(scale), ld, n_rows, c[0].data_ptr(), c[1].data_ptr(),
                                       c[2].data_ptr(), ps(c[3]), ps(c[4]), ps(c[5]), 1e-2, 2e-2, 1e-3, 0.9, 0.999, 1e-7, st), 'plain')
        single.append((state + sstate, copies))
    pre = [x.clone() for x in keep[7 * 3 + 4]]                      # the heavy table's p, m, v before the update
    check(L_.drx_rows_csr_adam_multi(tabs, 4, 0.9, 0.999, 1e-7, st), 'multi')
    torch.cuda.synchronize()
    for fused, sep in single:
        for a, b in zip(fused, sep):
            if a is not None:
                assert torch.equal(a, b)
    # the split sums against fp64: g = sum of the lookups' rows, ApplyAdam(g + l2 p)
    order, ptr, src, state = keep[7 * 3], keep[7 * 3 + 1], keep[7 * 3 + 2], keep[7 * 3 + 4]
    keys = torch.empty(3000, dtype=torch.long, device=dev)
    keys[order.long()] = torch.repeat_interleave(torch.arange(20, device=dev), (ptr[1:] - ptr[:-1]).long())
    assert int((ptr[1:] - ptr[:-1]).min()) >= 64
    g64 = torch.zeros(20, 52, dtype=torch.float64, device=dev).index_add_(0, keys, src.double())
    p0, m0, v0 = (x.double() for x in pre)
    gg = g64 + 1e-3 * p0
    m1 = m0 + (gg - m0) * (1 - 0.9)
    v1 = v0 + (gg * gg - v0) * (1 - 0.999)
    p1 = p0 - m1 * 1e-2 / (v1.sqrt() + 1e-7)
    assert torch.allclose(state[0].double(), p1, rtol=0, atol=1e-5) and torch.allclose(state[1].double(), m1, rtol=0, atol=1e-5)
    # the outer form against the rows written out
    n_rows, ld, T, group = specs[1]
    order, ptr, src, scale = keep[7], keep[8], keep[9], keep[10]
    rows = (scale[:, None] * src[torch.arange(T, device=dev) // group]).contiguous()
    a = [rnd(n_rows, ld), rnd(n_rows, ld), rnd(n_rows, ld).abs(), rnd(n_rows), rnd(n_rows), rnd(n_rows).abs()]
    b = [x.clone() for x in a]
    check(L_.drx_rows_csr_adam_outer(ptr.data_ptr(), order.data_ptr(), scale.data_ptr(), src.data_ptr(), group, ld, n_rows, *[x.data_ptr() for x in a],
                                     1e-2, 2e-2, 1e-3, 0.9, 0.999, 1e-7, st), 'outer')
    check(L_.drx_rows_csr_adam(ptr.data_ptr(), order.data_ptr(), rows.data_ptr(), scale.data_ptr(), ld, n_rows, *[x.data_ptr() for x in b],
                               1e-2, 2e-2, 1e-3, 0.9, 0.999, 1e-7, st), 'plain')
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=0, atol=2e-6)           # (a fused multiply-add where the written rows were rounded first)

    # drx_caser_step_small against drx_caser_fwd_bwd + drx_adam_segments
    from drecpy_amd.engine_caser import CaserEngine
    rng = np.random.default_rng(3)
    U, N, Lw, T_, neg, d, n_v, n_h, B = 30, 90, 4, 2, 2, 20, 3, 8, 70
    eng = CaserEngine(U, N, Lw, T_, neg, d, n_v, n_h)
    eng.set_params(ca.init_params(rng, U, N, Lw, d, n_v, n_h, np.float64))
    eng.lr, eng.reg = 5e-3, 1e-4
    uids, before, after = rng.integers(0, U, size=B), rng.integers(0, N, size=(B, Lw)), rng.integers(0, N, size=(B, T_ + T_ * neg))
    sw0 = eng.sw.clone()
    eng.step(0, uids, before, after, None, 0.0)
    sw_fused, m_fused, v_fused = eng.sw.clone(), eng.state['sw'][0].clone(), eng.state['sw'][1].clone()
    # the small-weight half by hand from the same pre-step parameters: drx_caser_fwd_bwd, then drx_adam_segments
    eng2 = CaserEngine(U, N, Lw, T_, neg, d, n_v, n_h)
    eng2.set_params(ca.init_params(np.random.default_rng(3), U, N, Lw, d, n_v, n_h, np.float64))
    assert torch.equal(eng2.sw, sw0)
    eng2.lr, eng2.reg = 5e-3, 1e-4
    i32 = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.int32)).to(dev)
    uid_t, bef_t, aft_t = i32(uids), i32(before), i32(after)
    grid = L_.drx_caser_grid(C.byref(eng2.D), B)
    f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
    dE, dW1, db1, dPu = f(B * Lw, eng2.ld), f(B * eng2.Tp, eng2.ld2), f(B * eng2.Tp), f(B, eng2.ld)
    gpart, lpart, gsw = f(grid, eng2.D.n_small), f(grid), f(eng2.D.n_small + 1)
    A = _lib.CaserArgs()
    A.item_emb, A.user_emb, A.W1, A.b1, A.sw = (x.data_ptr() for x in (eng2.item_emb, eng2.user_emb, eng2.W1, eng2.b1, eng2.sw))
    A.uid, A.before, A.after, A.keep, A.rate, A.B, A.mask_seed = uid_t.data_ptr(), bef_t.data_ptr(), aft_t.data_ptr(), None, 0.0, B, 0
    A.dE, A.dW1, A.db1, A.dPu, A.gsw_part, A.loss_part, A.cat_out = (dE.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dPu.data_ptr(),
                                                                      gpart.data_ptr(), lpart.data_ptr(), None)
    check(L_.drx_caser_fwd_bwd(C.byref(eng2.D), C.byref(A), gsw.data_ptr(), st), 'drx_caser_fwd_bwd')
    alpha, l2c = eng2._alphas(0), 2.0 * eng2.reg
    sg = _lib.AdamSegments()
    sg.n = len(eng2.seg)
    for i, (_, start, n, regd, layer) in enumerate(eng2.seg):
        sg.start[i], sg.len[i], sg.alpha[i], sg.l2_coef[i] = start, n, alpha[layer], (l2c if regd else 0.0)
    m2, v2 = eng2.state['sw']
    check(L_.drx_adam_segments(eng2.sw.data_ptr(), m2.data_ptr(), v2.data_ptr(), gsw.data_ptr(), C.byref(sg), eng2.beta1, eng2.beta2, eng2.eps, st),
          'drx_adam_segments')
    torch.cuda.synchronize()
    assert torch.equal(eng2.sw, sw_fused) and torch.equal(m2, m_fused) and torch.equal(v2, v_fused)


@pytest.mark.parametrize('lists', [((500, 40),), ((20480, 3706), (49152, 3706), (4096, 6040)), ((7, 3), (1, 1), (300, 100000), (64, 2))])
def test_device_csr_of_key_lists_equals_the_host_counting_sort(lists):
    """drx_batch_csr_device (one stable sort of all lists' (row, lookup) pairs + a binary search per row) against drx_batch_csr: the same
    row_ptr and the same order — a row's lookups ascending — for every list, rows nobody names included."""
    import ctypes as C
    import torch
    from drecpy_amd import _lib
    L_ = _lib.lib()
    rng = np.random.default_rng(len(lists))
    dev = torch.device('cuda:0')
    ls = (_lib.CsrList * len(lists))()
    keep, want = [], []
    for l, (T, n_rows) in zip(ls, lists):
        keys = (rng.zipf(1.3, size=T) % n_rows).astype(np.int32) if n_rows > 3 else rng.integers(0, n_rows, size=T).astype(np.int32)
        rp, od = np.empty(n_rows + 1, np.int32), np.empty(T, np.int32)
        assert L_.drx_batch_csr(keys.ctypes.data, T, n_rows, rp.ctypes.data, od.ctypes.data) == 0
        want.append((rp, od))
        k_d = torch.as_tensor(keys).to(dev)
        rp_d = torch.full((n_rows + 1,), -7, dtype=torch.int32, device=dev)
        od_d = torch.full((T,), -7, dtype=torch.int32, device=dev)
        keep.append((k_d, rp_d, od_d))
        l.keys, l.T, l.n_rows, l.row_ptr, l.order = k_d.data_ptr(), T, n_rows, rp_d.data_ptr(), od_d.data_ptr()
    need = int(L_.drx_batch_csr_device_bytes(ls, len(lists)))
    assert need > 0
    scratch = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.check(L_.drx_batch_csr_device(ls, len(lists), scratch.data_ptr(), need, _lib.stream_ptr(dev)), 'drx_batch_csr_device')
    assert L_.drx_batch_csr_device(ls, len(lists), scratch.data_ptr(), need - 1024, _lib.stream_ptr(dev)) != 0      # (scratch too small: refused)
    torch.cuda.synchronize()
    for (rp, od), (_, rp_d, od_d) in zip(want, keep):
        assert np.array_equal(rp_d.cpu().numpy(), rp) and np.array_equal(od_d.cpu().numpy(), od)


@pytest.mark.parametrize('drop', [False, True])
def test_a_batch_on_the_device_takes_the_same_step_as_the_same_batch_from_the_host(drop):
    """Device tensors in (Caser.fit(device_sampler=True)): the lookups are grouped by row on the device and the step runs the launches of
    a host batch — the same parameters BIT FOR BIT; with device_csr off the scatter + dense-Adam update (r02 – r05) only agrees to rounding."""
    import torch
    from drecpy_amd.engine_caser import CaserEngine
    d, L, n_v, n_h, T, neg, B, U, N = 50, 5, 4, 16, 3, 3, 700, 60, 90
    rng = np.random.default_rng(4)
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    engs = []
    for kind in ('host', 'device', 'device-scatter'):
        e = CaserEngine(U, N, L, T, neg, d, n_v, n_h)
        e.set_params(p)
        e.lr, e.reg = 5e-3, 1e-4
        e.device_csr = kind != 'device-scatter'
        engs.append(e)
    nx = n_v + L * n_h
    dev = engs[0].device
    for step in range(4):
        uids = rng.integers(0, U, size=B)
        before = rng.zipf(1.5, size=(B, L)) % N
        after = rng.integers(0, N, size=(B, T + T * neg))
        keep = (rng.random((B, nx)) >= 0.5) if drop else None
        rate = 0.5 if drop else 0.0
        losses = [engs[0].step(step, uids, before, after, keep, rate, want_loss=True)]
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        for e in engs[1:]:
            losses.append(e.step(step, t(uids), t(before), t(after), keep, rate, want_loss=True))
        assert losses[0] == losses[1] and abs(losses[2] - losses[0]) < 1e-5 * abs(losses[0])
    a, b, c = (e.get_params() for e in engs)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
        np.testing.assert_allclose(c[k], a[k], rtol=0, atol=2e-6, err_msg=k)


def test_ring_slots_with_cached_argument_structs_take_the_same_steps(monkeypatch):
    """Caser.fit(device_sampler=True) fills ring slots the engine owns (device_slot / group_slot) and steps on them with argument structs
    built once per slot (_step_slot): the same parameters, bit for bit, as the same draws through prepare_device_batch and the general
    step — over more steps than the ring has slots, with dropout, a learning-rate change in between."""
    import torch
    from drecpy_amd.engine_caser import CaserEngine
    d, L, n_v, n_h, T, neg, B, U, N = 50, 5, 4, 16, 3, 3, 300, 70, 120
    rng = np.random.default_rng(9)
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    engs = []
    for _ in range(2):
        e = CaserEngine(U, N, L, T, neg, d, n_v, n_h)
        e.set_params(p)
        e.lr, e.reg = 5e-3, 1e-4
        engs.append(e)
    dev = engs[0].device
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
    for step in range(10):
        uids = rng.integers(0, U, size=B)
        before = rng.zipf(1.5, size=(B, L)) % N
        after = rng.integers(0, N, size=(B, T + T * neg))
        if step == 6:
            for e in engs:
                e.lr = 1e-2
        sl = engs[0].device_slot(B)
        sl['uid'].copy_(t(uids)); sl['before'].copy_(t(before)); sl['after'].copy_(t(after))
        engs[0].step(step, engs[0].group_slot(sl), rate=0.5, mask_seed=77 + step)
        engs[1].step(step, engs[1].prepare_device_batch(t(uids), t(before), t(after)), rate=0.5, mask_seed=77 + step)
    assert 'step' in engs[0]._dev_slots['slots'][0]                    # (the cached structs were in use)
    la = engs[0].step(10, engs[0].group_slot(engs[0].device_slot(B)), rate=0.5, mask_seed=5, want_loss=True)      # (a slot through the general path)
    assert np.isfinite(la)
    a, b = engs[0].get_params(), engs[1].get_params()
    engs[1].step(10, engs[1].prepare_device_batch(*(engs[0]._dev_slots['slots'][(engs[0]._dev_slots['i'] - 1) % 3][k] for k in ('uid', 'before', 'after'))),
                 rate=0.5, mask_seed=5, want_loss=True)
    a, b = engs[0].get_params(), engs[1].get_params()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
