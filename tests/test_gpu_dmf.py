"""GPU parity of the DMF step (drx_dmf_* + drx_scatter_rows + drx_adam_*) against oracle/dmf_oracle.py, and of the
bf16-MFMA all-pairs scorer against the fp32 cosine."""
import numpy as np
import pytest

from oracle import data_oracle as do
from oracle import dmf_oracle as dm

pytestmark = pytest.mark.gpu


def _problem(rng, U, N, nnz):
    u = rng.integers(0, U, size=nnz)
    i = rng.integers(0, N, size=nnz)
    _, first = np.unique(u * N + i, return_index=True)
    u, i = u[np.sort(first)], i[np.sort(first)]
    v = rng.integers(1, 6, size=len(u)).astype(np.float64)
    csr = do.interaction_csr(u, i, v, U, N)
    csc = do.interaction_csr(i, u, v, N, U)
    dense = np.zeros((U, N))
    dense[u, i] = v
    return csr, csc, dense


@pytest.mark.parametrize('update', ['scan', 'scatter'])      # the two ways the first-layer kernels are updated (engine_dmf.py)
@pytest.mark.parametrize('uf,itf,l2n,B', [((64, 32), (64, 32), True, 64), ((16,), (24, 16), True, 33), ((32, 20, 8), (12, 8), False, 50),
                                          # towers wider than a wavefront (a lane holds units k and k + 64): the reference's own
                                          # examples/consistency_eval/dmf.py:20 builds [128, 64]
                                          ((128, 64), (128, 64), True, 64), ((100, 70, 40), (90, 40), True, 33), ((128,), (72, 128), False, 50)])
def test_dmf_steps_match_oracle(uf, itf, l2n, B, update):
    from drecpy_amd.engine_dmf import DmfEngine
    rng = np.random.default_rng(len(uf) * 7 + B)
    U, N = 70, 90
    csr, csc, dense = _problem(rng, U, N, 1500)
    p = dm.init_params(rng, U, N, uf, itf, np.float64)
    for k in p:
        if k.endswith('_b'):
            p[k] = rng.normal(0, 0.05, size=p[k].shape)
    eng = DmfEngine(U, N, uf, itf, l2n)
    eng.set_interactions(csr, csc)
    wide0 = max(uf[0], itf[0]) > 64              # (first layers wider than 64: a quarter-wave cannot hold a gradient row — touches + scatter)
    assert eng.first_layer_update == ('scatter' if wide0 else 'scan')
    if wide0 and update == 'scan':
        pytest.skip('the scan update of the first-layer kernels takes rows of up to 64 floats')
    eng.first_layer_update = update
    eng.set_params(p)
    eng.lr, eng.reg = 2e-3, 1e-3
    st = dm.adam_state(p)
    for step in range(6):
        uids = rng.integers(0, U, size=B)
        iids = rng.integers(0, N, size=B)
        y = rng.random(B)
        lo = dm.step(p, st, step, dense[uids], dense[:, iids].T.copy(), y, 2e-3, 1e-3, len(uf), len(itf), l2n)
        lg = eng.step(step, uids, iids, y, want_loss=True)
        assert abs(lg - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)
    uids = rng.integers(0, U, size=40)
    iids = rng.integers(0, N, size=40)
    pred = eng.predict(uids, iids).cpu().numpy()
    want, _ = dm.forward(p, dense[uids], dense[:, iids].T.copy(), len(uf), len(itf), l2n)
    assert np.max(np.abs(pred - want) / np.maximum(np.abs(want), 1e-6)) < 1e-4


def test_mfma_bf16_all_pairs_scorer():
    from drecpy_amd.engine_dmf import DmfEngine
    rng = np.random.default_rng(3)
    U, N = 100, 333
    csr, csc, dense = _problem(rng, U, N, 6000)
    p = dm.init_params(rng, U, N, (64, 32), (64, 32), np.float64)
    eng = DmfEngine(U, N)
    eng.set_interactions(csr, csc)
    eng.set_params(p)
    uids = np.arange(0, U, 3)
    sc = eng.score_matrix_bf16(uids).cpu().numpy()
    assert sc.shape == (len(uids), N)
    for r, u in enumerate(uids[:8]):
        want, _ = dm.forward(p, np.repeat(dense[u:u + 1], N, axis=0), dense.T.copy(), 2, 2)
        assert np.max(np.abs(sc[r] - want)) < 1.5e-2          # bf16 operands (8 mantissa bits), fp32 accumulation
    # exactness of the MFMA lane maps: integers are exact in bf16
    import torch
    from drecpy_amd import _lib
    a = torch.zeros(70, 64, device='cuda'); b = torch.zeros(45, 64, device='cuda')
    a[:, :32] = torch.randint(-4, 5, (70, 32), device='cuda').float()
    b[:, :32] = torch.randint(-4, 5, (45, 32), device='cuda').float()
    out = torch.empty(70, 45, device='cuda')
    _lib.check(_lib.lib().drx_score_pairs_bf16(_lib.ptr(a), 70, _lib.ptr(b), 45, 64, 32, None, _lib.ptr(out), 45, _lib.stream_ptr()), "score")
    want = torch.clamp(a[:, :32] @ b[:, :32].t(), min=1e-6)
    assert torch.equal(out, want)


def test_dmf_fit_matches_oracle_end_to_end():
    """DMF.fit() with the reference-exact PointSampler stream and injected weights vs the oracle."""
    from helpers import load_frames
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import DMF
    frame = {k: v.copy() for k, v in load_frames()['pt_int_dense'].items()}     # 64 users x 40 items, values 0..5
    ds = InteractionDataset.read_df(frame, verbose=False)
    uid, _ = do.first_appearance_codes(frame['user'].tolist())
    iid, _ = do.first_appearance_codes(frame['item'].tolist())
    U, N = int(uid.max()) + 1, int(iid.max()) + 1
    p = dm.init_params(np.random.default_rng(2), U, N, (16, 8), (16, 8), np.float32)
    # 7 epochs: the 8th batch of this sampler stream holds a pair (user 53, item 7) whose two towers are left with the same single
    # active unit -> cosine exactly 1 with target 0.  Keras' clip_by_value passes a gradient of 1 / 2e-7 or none at all depending on
    # whether the fp32 dot product comes out at 1 - 2^-24 or 1 - 2^-23: the last bit of a sum decides, in TF as much as here — not
    # something a parity test can pin (the fp64 oracle sees exactly 1.0 and clips).
    epochs, B, seed = 7, 32, 10
    model = DMF(user_factors=[16, 8], item_factors=[16, 8], seed=seed, verbose=False)
    model.fit(ds, epochs=epochs, batch_size=B, learning_rate=2e-3, reg_rate=1e-3, neg_ratio=3, initial_weights=p)
    # oracle: same sampler stream (stdlib random restatement), standardised targets (min 0, max 5)
    dense = np.zeros((U, N))
    np.add.at(dense, (uid, iid), frame['interaction'].astype(np.float64))
    smp = do.PointSamplerOracle(uid, iid, frame['interaction'], 3, 1e-3, seed)
    po = {k: v.astype(np.float64) for k, v in p.items()}
    st = dm.adam_state(po)
    mn, mx = float(frame['interaction'].min()), float(frame['interaction'].max())
    mn = 0.0 if mn == 1 else mn
    for s in range(epochs):
        batch = smp.sample(B)
        u = np.array([t[0] for t in batch]); i = np.array([t[1] for t in batch])
        y = (np.array([float(t[2]) for t in batch]) - mn) / (mx - mn)
        dm.step(po, st, s, dense[u], dense[:, i].T.copy(), y.astype(np.float32).astype(np.float64), 2e-3, 1e-3, 2, 2)
    g = model._engine.get_params()
    for k in po:
        np.testing.assert_allclose(g[k], po[k], rtol=0, atol=3e-5, err_msg=k)
    raw_u, raw_i = frame['user'][0], frame['item'][5]
    u0, i0 = ds.user_to_uid(raw_u), ds.item_to_iid(raw_i)
    want, _ = dm.forward(po, dense[u0:u0 + 1], dense[:, i0:i0 + 1].T.copy(), 2, 2)
    assert abs(model.predict(raw_u, raw_i) - (mn + (mx - mn) * want[0])) < 1e-4
    ranked = model.rank(raw_u, [ds.iid_to_item(j) for j in range(N)], novelty=True, n=5)
    assert len(ranked) == 5 and all(ranked[j][0] >= ranked[j + 1][0] for j in range(4))
    sm = model.score_matrix([raw_u]).cpu().numpy()
    assert sm.shape == (1, N)


# ---- ModifiedDMF: the model BASELINE.json config 3 names (examples/extending_recommender_dmf.py) ------------------------------
def _modified_engine(U, N, csr, csc, p, uf=(64, 32), itf=(64, 32)):
    from drecpy_amd.engine_dmf import DmfEngine
    from drecpy_amd.Recommender import Variable
    eng = DmfEngine(U, N, uf, itf, True)
    eng.set_interactions(csr, csc)
    w = Variable([1.0])
    eng.bind_prediction_scale(w, broadcast_targets=True)
    eng.set_params(p)
    return eng, w


@pytest.mark.parametrize('shape,B', [('small', 48), ('ml-1m', 256)])
def test_modified_dmf_steps_match_oracle(shape, B):
    """The registered scalar multiplies every prediction, the loss is the (B,B) Keras broadcast, three Adam applies per step with
    the scalar first (t = 3s+1, 3s+2, 3s+3): HIP step against the fp64 oracle, which computes the broadcast literally."""
    rng = np.random.default_rng(B)
    if shape == 'small':
        U, N = 70, 90
        csr, csc, dense = _problem(rng, U, N, 1500)
        uf, itf, steps = (16, 8), (24, 8), 8
    else:
        from test_gpu_baseline_shapes import _ml1m_ratings
        U, N, csr, csc, dense = _ml1m_ratings()
        uf, itf, steps = (64, 32), (64, 32), 3
    p = dm.init_params(rng, U, N, uf, itf, np.float64)
    p['extra_w'] = np.array([0.9])
    eng, w = _modified_engine(U, N, csr, csc, p, uf, itf)
    eng.lr, eng.reg = 2e-3, 1e-3
    st = dm.adam_state(p)
    for step in range(steps):
        uids = rng.integers(0, U, size=B)
        iids = rng.integers(0, N, size=B)
        y = rng.random(B)
        lo = dm.step(p, st, step, dense[uids], dense[:, iids].T.copy(), y, 2e-3, 1e-3, len(uf), len(itf), True, broadcast_targets=True)
        lg = eng.step(step, uids, iids, y, want_loss=True)
        assert abs(lg - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    assert abs(g['extra_w'][0] - 0.9) > 1e-3                     # the scalar trains ...
    assert w.numpy()[0] == g['extra_w'][0]                       # ... and the registered handle sees the engine's memory
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)
    uids = rng.integers(0, U, size=40)
    iids = rng.integers(0, N, size=40)
    want, _ = dm.forward(p, dense[uids], dense[:, iids].T.copy(), len(uf), len(itf), True)
    assert np.max(np.abs(eng.predict(uids, iids).cpu().numpy() - want) / np.maximum(np.abs(want), 1e-6)) < 1e-4
    assert np.max(np.abs(eng.predict(uids, iids, scaled=False).cpu().numpy() * p['extra_w'][0] - want)) < 1e-5
    sc = eng.score_matrix_bf16(uids[:4]).cpu().numpy()           # the MFMA scorer applies the scale too
    want4, _ = dm.forward(p, np.repeat(dense[uids[0]:uids[0] + 1], N, axis=0), dense.T.copy(), len(uf), len(itf), True)
    assert np.max(np.abs(sc[0] - want4)) < 1.5e-2


def test_modified_dmf_fit_matches_oracle_end_to_end():
    """examples/extending_recommender_dmf.py's ModifiedDMF through the public fit(): registration order -> Adam counters,
    reference-exact PointSampler stream, _predict through the overridden _predict_batch."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))
    from extending_recommender_dmf import ModifiedDMF
    from helpers import load_frames
    from drecpy_amd.Dataset import InteractionDataset
    frame = {k: v.copy() for k, v in load_frames()['pt_int_dense'].items()}
    ds = InteractionDataset.read_df(frame, verbose=False)
    uid, _ = do.first_appearance_codes(frame['user'].tolist())
    iid, _ = do.first_appearance_codes(frame['item'].tolist())
    U, N = int(uid.max()) + 1, int(iid.max()) + 1
    p = dm.init_params(np.random.default_rng(2), U, N, (16, 8), (16, 8), np.float32)
    epochs, B, seed = 8, 32, 10
    model = ModifiedDMF(user_factors=[16, 8], item_factors=[16, 8], seed=seed, verbose=False)
    model.fit(ds, epochs=epochs, batch_size=B, learning_rate=2e-3, reg_rate=1e-3, neg_ratio=3, initial_weights=p)
    assert model.trainable_weights == [model._extra_weight] and len(model.trainable_models) == 2
    assert model._apply_order()[0] is model._extra_weight            # the tf.Variable's apply comes first (recommender_abc.py:194-196)
    dense = np.zeros((U, N))
    np.add.at(dense, (uid, iid), frame['interaction'].astype(np.float64))
    smp = do.PointSamplerOracle(uid, iid, frame['interaction'], 3, 1e-3, seed)
    po = {k: v.astype(np.float64) for k, v in p.items()}
    po['extra_w'] = np.array([1.0])
    st = dm.adam_state(po)
    mn, mx = float(frame['interaction'].min()), float(frame['interaction'].max())
    mn = 0.0 if mn == 1 else mn
    for s in range(epochs):
        batch = smp.sample(B)
        u = np.array([t[0] for t in batch]); i = np.array([t[1] for t in batch])
        y = (np.array([float(t[2]) for t in batch]) - mn) / (mx - mn)
        dm.step(po, st, s, dense[u], dense[:, i].T.copy(), y.astype(np.float32).astype(np.float64), 2e-3, 1e-3, 2, 2, True, broadcast_targets=True)
    g = model._engine.get_params()
    for k in po:
        np.testing.assert_allclose(g[k], po[k], rtol=0, atol=3e-5, err_msg=k)
    assert abs(g['extra_w'][0] - 1.0) > 1e-4
    raw_u, raw_i = frame['user'][0], frame['item'][5]
    u0, i0 = ds.user_to_uid(raw_u), ds.item_to_iid(raw_i)
    want, _ = dm.forward(po, dense[u0:u0 + 1], dense[:, i0:i0 + 1].T.copy(), 2, 2)         # includes the scale
    assert abs(model.predict(raw_u, raw_i) - (mn + (mx - mn) * want[0])) < 1e-4
    ranked = model.rank(raw_u, [ds.iid_to_item(j) for j in range(N)], novelty=True, n=5)
    assert len(ranked) == 5 and all(ranked[j][0] >= ranked[j + 1][0] for j in range(4))
    # a plain DMF that registers something its fused step does not know fails loudly
    from drecpy_amd.Recommender import DMF, Variable

    class Stray(DMF):
        def _pre_fit(self, learning_rate, neg_ratio, reg_rate, **kwds):
            super()._pre_fit(learning_rate, neg_ratio, reg_rate, **kwds)
            self._register_trainable(Variable([1.]))
    with pytest.raises(NotImplementedError, match='does not update them'):
        Stray(user_factors=[16, 8], item_factors=[16, 8], seed=seed, verbose=False).fit(ds, epochs=1, batch_size=8)
    with pytest.raises(Exception, match='supports towers of 1..4 layers of width 1..128'):
        DMF(user_factors=[256, 64], item_factors=[64], verbose=False)


def test_update_weights_applies_keras_adam_on_the_device():
    """RecommenderABC._update_weights (recommender_abc.py:328-334) with the registered optimizers.Adam: one apply per item, the
    counter advancing per call — against the closed form."""
    import torch
    from drecpy_amd import optimizers
    from drecpy_amd.Recommender import DMF, Variable
    from oracle import cdae_oracle as co
    m = DMF(verbose=False)
    m._register_optimizer(optimizers.Adam(learning_rate=0.01))
    a, b = Variable(np.arange(8, dtype=np.float32)), Variable([3.0, -1.0, 2.0])          # 3 elements: the padded path
    ga, gb = torch.full((8,), 0.5, device='cuda'), torch.tensor([1.0, -2.0, 0.25], device='cuda')
    m._update_weights([ga, [gb]], [a, [b]])
    for var, g0, t, x0 in ((a, np.full(8, 0.5), 1, np.arange(8.0)), (b, np.array([1.0, -2.0, 0.25]), 2, np.array([3.0, -1.0, 2.0]))):
        want = x0 - co.adam_alpha(0.01, t) * (co.ADAM_OMB1 * g0) / (np.sqrt(co.ADAM_OMB2 * g0 * g0) + co.ADAM_EPS)
        np.testing.assert_allclose(var.numpy(), want, rtol=2e-6)
    assert m.optimizer.iterations == 2


def test_update_weights_keeps_adam_moments_of_a_layer_handle_across_steps():
    """A layer / model handle hands out FRESH view objects of the same memory on every `trainable_weights` read (trainables._Handle):
    the moments must follow the memory, not the object (keyed by id() they restarted from zero each step while
    `iterations` went on).  Three steps with changing gradients against closed-form Keras Adam."""
    import torch
    from drecpy_amd import optimizers
    from drecpy_amd.Recommender import DMF
    from drecpy_amd.Recommender.trainables import TrainableLayer
    from oracle import cdae_oracle as co
    m = DMF(verbose=False)
    m._register_optimizer(optimizers.Adam(learning_rate=0.01))
    flat = torch.arange(24, dtype=torch.float32, device='cuda') / 7.0
    layer = TrainableLayer('dense', lambda: [flat[0:16].view(4, 4), flat[16:24]])          # kernel and bias as views of one array
    x = [flat[0:16].cpu().numpy().astype(np.float64).reshape(4, 4), flat[16:24].cpu().numpy().astype(np.float64)]
    mom = [[np.zeros_like(a), np.zeros_like(a)] for a in x]
    rng = np.random.default_rng(3)
    for t in range(1, 4):
        g = [rng.normal(size=a.shape) for a in x]
        m._update_weights([[torch.tensor(a, dtype=torch.float32, device='cuda') for a in g]], [layer])
        for a, (m1, v1), ga in zip(x, mom, g):
            m1 += (ga - m1) * co.ADAM_OMB1
            v1 += (ga * ga - v1) * co.ADAM_OMB2
            a -= co.adam_alpha(0.01, t) * m1 / (np.sqrt(v1) + co.ADAM_EPS)
    got = [w.cpu().numpy() for w in layer.trainable_weights]
    for a, w in zip(x, got):
        np.testing.assert_allclose(w, a, rtol=5e-6, atol=1e-7)
    assert m.optimizer.iterations == 3


@pytest.mark.parametrize('update', ['scan', 'scatter'])
def test_dmf_steps_with_empty_rows_columns_and_repeated_ids(update):
    """Edge cases of the first-layer paths: users without any interaction and items nobody rated inside the batch (zero input vectors:
    the l2 normaliser falls back to its epsilon), every sample the same user, every sample the same item — against the oracle."""
    from drecpy_amd.engine_dmf import DmfEngine
    rng = np.random.default_rng(23)
    U, N = 40, 50
    u = rng.integers(0, U - 5, size=400)                 # users U-5 .. U-1 have no row
    i = rng.integers(0, N - 6, size=400)                 # items N-6 .. N-1 have no column
    _, first = np.unique(u * N + i, return_index=True)
    u, i = u[np.sort(first)], i[np.sort(first)]
    v = rng.integers(1, 6, size=len(u)).astype(np.float64)
    csr, csc = do.interaction_csr(u, i, v, U, N), do.interaction_csr(i, u, v, N, U)
    dense = np.zeros((U, N))
    dense[u, i] = v
    p = dm.init_params(rng, U, N, (16, 8), (16, 8), np.float64)
    eng = DmfEngine(U, N, (16, 8), (16, 8), True)
    eng.set_interactions(csr, csc)
    eng.first_layer_update = update
    eng.set_params(p)
    eng.lr, eng.reg = 2e-3, 1e-3
    st = dm.adam_state(p)
    B = 24
    batches = [(rng.integers(0, U, size=B), rng.integers(0, N, size=B)),                       # mixes empty and non-empty ids
               (np.full(B, U - 1), rng.integers(0, N, size=B)),                                # one user, and an empty one
               (rng.integers(0, U, size=B), np.full(B, N - 1)),                                # one item, and an empty one
               (np.full(B, 3), np.full(B, 7)),                                                 # one pair repeated
               (rng.integers(0, U, size=B), rng.integers(0, N, size=B))]
    for step, (uids, iids) in enumerate(batches):
        y = rng.random(B)
        lo = dm.step(p, st, step, dense[uids], dense[:, iids].T.copy(), y, 2e-3, 1e-3, 2, 2, True)
        lg = eng.step(step, uids, iids, y, want_loss=True)
        assert abs(lg - lo) / abs(lo) < 1e-4, (step, lg, lo)
    g = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g[k], p[k], rtol=0, atol=3e-5, err_msg=k)


# ---- DMF.fit(device_sampler=True): batches drawn and prepared on the device ----------------------------------------------------------
@pytest.mark.parametrize('B,U,N', [(64, 70, 90), (4096, 600, 300), (1000, 5000, 20)])
def test_distinct_ids_on_the_device_equal_the_host_helper(B, U, N):
    """drx_dmf_batch_distinct_device (one stable sort of the 2B (id, sample) pairs + one workgroup numbering the runs) against
    drx_batch_distinct on the host: distinct ids ascending, every sample's index among them, the samples per distinct id ascending,
    the counts, the batch mean of y."""
    import torch
    from drecpy_amd.engine_dmf import DmfEngine
    rng = np.random.default_rng(B)
    csr, csc, _ = _problem(rng, U, N, 4 * (U + N))
    eng = DmfEngine(U, N, (16, 8), (16, 8), True)
    eng.set_interactions(csr, csc)
    u = rng.integers(0, U, size=B).astype(np.int32)
    i = np.minimum(rng.pareto(1.1, size=B).astype(np.int64), N - 1).astype(np.int32)       # a few hot items, many repeats
    y = rng.random(B).astype(np.float32)
    host = eng.prepare_batch(u, i, y)
    offs, lens, _ = eng._batch_layout(B)
    at = dict(zip(eng._BATCH_ARRAYS, offs))
    view = lambda name, n: host['buf'][at[name]:at[name] + 4 * n].view(np.int32)
    dev = eng.prepare_batch_device(B, 0, 0, triples=[torch.as_tensor(a).cuda() for a in (u, i, y)])['device']
    torch.cuda.synchronize()
    nd = dev['nd'].cpu().numpy()
    assert (int(nd[0]), int(nd[1])) == (host['n_du'], host['n_di'])
    a = dev['arr'].cpu().numpy()
    for r, (name, n) in enumerate((('du', nd[0]), ('di', nd[1]), ('inv_u', B), ('inv_i', B), ('gptr_u', nd[0] + 1), ('gptr_i', nd[1] + 1),
                                   ('grows_u', B), ('grows_i', B))):
        assert np.array_equal(a[r][:n], view(name, n)), name
    assert abs(float(dev['y_mean'].item()) - float(y.astype(np.float64).mean())) < 1e-6


@pytest.mark.parametrize('seg_len', [None, 8])
@pytest.mark.parametrize('modified', [False, True])
def test_a_step_on_a_device_prepared_batch_equals_the_host_prepared_one(modified, seg_len):
    """The same triples through prepare_batch (host arrays, one upload) and through prepare_batch_device (the distinct counts and the
    batch mean stay on the device: DrxDmfArgs.nd_dev / y_mean_dev): identical parameters after three steps, bit for bit.
    seg_len = 8: rows and columns longer than 8 non-zeros are gathered in segments — the work list with its partial rows built by
    drx_dmf_work_order on the host and by drx_dmf_work_order_device on the device (the lists differ in order, the sums do not)."""
    import torch
    from drecpy_amd.engine_dmf import DmfEngine
    from drecpy_amd.Recommender import Variable
    rng = np.random.default_rng(17)
    U, N, B = 120, 80, 200
    csr, csc, _ = _problem(rng, U, N, 2500)
    p = dm.init_params(rng, U, N, (32, 16), (24, 16), np.float64)
    engs = []
    for _ in range(2):
        e = DmfEngine(U, N, (32, 16), (24, 16), True)
        e.set_interactions(csr, csc)
        if modified:
            e.bind_prediction_scale(Variable([1.0]), broadcast_targets=True)
        e.set_params(p)
        e.lr, e.reg = 2e-3, 1e-3
        if seg_len is not None:
            e._seg_len = seg_len
            assert e._device_seg_len() == seg_len
        engs.append(e)
    for step in range(3):
        u = rng.integers(0, U, size=B).astype(np.int32)
        i = rng.integers(0, N // 4, size=B).astype(np.int32)
        y = rng.random(B).astype(np.float32)
        la = engs[0].step(step, u, i, y, want_loss=True)
        lb = engs[1].step(step, engs[1].prepare_batch_device(B, 0, 0, triples=[torch.as_tensor(a).cuda() for a in (u, i, y)]), want_loss=True)
        assert la == lb, (step, la, lb)
    if seg_len is not None:                          # (the device list really held segments)
        nw = next(iter(engs[1]._dev_ring.values()))['nw'].cpu().numpy()
        assert nw[1] > 0 and nw[0] > nw[1]
    ga, gb = engs[0].get_params(), engs[1].get_params()
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k


@pytest.mark.parametrize('modified', [False, True])
def test_the_cached_argument_structs_of_device_batches_change_nothing(modified, monkeypatch):
    """Steps on device-prepared batches reuse the argument structs an earlier step built for the same ring slot (only the stamp and the
    learning rates are refreshed: DmfEngine._step_device_cached): the same parameters, bit for bit, as with the cache off — over more steps
    than the ring has slots, with a learning-rate change and a set_params (which replaces nothing the cache points to) in between."""
    import torch
    from drecpy_amd.engine_dmf import DmfEngine
    from drecpy_amd.Recommender import Variable
    rng = np.random.default_rng(23)
    U, N, B = 150, 90, 180
    csr, csc, _ = _problem(rng, U, N, 2800)
    p = dm.init_params(rng, U, N, (32, 16), (24, 16), np.float64)
    engs = []
    for _ in range(2):
        e = DmfEngine(U, N, (32, 16), (24, 16), True)
        e.set_interactions(csr, csc)
        if modified:
            e.bind_prediction_scale(Variable([1.0]), broadcast_targets=True)
        e.set_params(p)
        e.lr, e.reg = 2e-3, 1e-3
        engs.append(e)
    monkeypatch.setattr(engs[1], '_step_device_cached', lambda *a, **k: False)          # the general path every time
    hits = [0]
    inner = engs[0]._step_device_cached

    def counted(*a, **k):
        r = inner(*a, **k)
        hits[0] += bool(r)
        return r
    monkeypatch.setattr(engs[0], '_step_device_cached', counted)
    for step in range(11):
        u = rng.integers(0, U, size=B).astype(np.int32)
        i = rng.integers(0, N // 3, size=B).astype(np.int32)
        y = rng.random(B).astype(np.float32)
        if step == 7:
            for e in engs:
                e.lr = 5e-3
        for e in engs:
            e.step(step, e.prepare_batch_device(B, 0, 0, triples=[torch.as_tensor(a).cuda() for a in (u, i, y)]))
    assert hits[0] >= 4, hits                       # (3 slots: filled on their second visit, reused from the third)
    ga, gb = engs[0].get_params(), engs[1].get_params()
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k
    engs[0].reg = 5e-3                              # a scalar the structs hold: the cache must notice
    engs[1].reg = 5e-3
    for step in range(11, 14):
        u = rng.integers(0, U, size=B).astype(np.int32)
        i = rng.integers(0, N // 3, size=B).astype(np.int32)
        y = rng.random(B).astype(np.float32)
        for e in engs:
            e.step(step, e.prepare_batch_device(B, 0, 0, triples=[torch.as_tensor(a).cuda() for a in (u, i, y)]))
    ga, gb = engs[0].get_params(), engs[1].get_params()
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k


def test_dmf_fit_with_the_device_sampler():
    """DMF.fit(device_sampler=True): the device PointSampler's triples carry the reference sampler's distribution AND values
    (two-sample chi-square against the reference-exact stream: users of positives, cells of positives, users / items of
    negatives; the standardised targets of the positives), and the fit runs and learns (the loss falls)."""
    import torch
    from helpers import load_frames
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import DMF
    from drecpy_amd.Sampler import PointSampler
    frame = {k: v.copy() for k, v in load_frames()['pt_int_dense'].items()}     # 64 users x 40 items, values 0..5: pairs below the threshold
    ds = InteractionDataset.read_df(frame, verbose=False)
    model = DMF(user_factors=[16, 8], item_factors=[16, 8], seed=5, verbose=False)
    model.fit(ds, epochs=60, batch_size=256, learning_rate=5e-3, reg_rate=1e-4, neg_ratio=3, device_sampler=True)
    assert model._sampler_kind.startswith('device')
    e = model._engine
    n = 60000
    prep = e.prepare_batch_device(n, 3, 4242)
    torch.cuda.synchronize()
    u, i = (t.cpu().numpy().astype(np.int64) for t in prep['device']['ids'])
    y = prep['device']['y'].cpu().numpy()
    ru, ri, rv, rneg = PointSampler(ds, 3, 1e-3, 11).sample_arrays(n)
    ru, ri, rneg = np.asarray(ru, np.int64), np.asarray(ri, np.int64), np.asarray(rneg).astype(bool)
    ry = np.asarray(model._standardize_value(np.asarray(rv, np.float64)), np.float64)
    U, N = model.n_users, model.n_items
    # negatives are the draws of pairs ABSENT from the frame; a positive's target is > 0 here (values 1..5 over a range of 5)
    neg_d = y == 0
    assert abs(neg_d.mean() - rneg.mean()) < 0.01

    def close(x, ycol, bins, what):
        hx, hy = np.bincount(x, minlength=bins).astype(float), np.bincount(ycol, minlength=bins).astype(float)
        hx = hx * (hy.sum() / hx.sum())
        live = (hx + hy) > 0
        chi = float((((hx - hy) ** 2) / (hx + hy))[live].sum() / max(1, live.sum() - 1))
        assert chi < 1.6, (what, chi)
    close(u[~neg_d], ru[~rneg], U, 'users of positives')
    close(u[~neg_d] * N + i[~neg_d], ru[~rneg] * N + ri[~rneg], U * N, 'cells of positives')
    close(u[neg_d], ru[rneg], U, 'users of negatives')
    close(i[neg_d], ri[rneg], N, 'items of negatives')
    close(np.rint(y[~neg_d] * 5).astype(np.int64), np.rint(ry[~rneg] * 5).astype(np.int64), 6, 'targets of positives')
    # the fit learned something: the model separates positives from negatives of a fresh device batch better than chance
    pred = e.predict(u[:4000], i[:4000]).cpu().numpy()
    assert pred[~neg_d[:4000]].mean() > pred[neg_d[:4000]].mean()


def test_device_sampled_negatives_carry_the_standardised_zero():
    """dmf.py:68 with recommender_abc.py:463-465: with use_nce every target — a negative's interaction value 0 included — is
    (v - min) / (max - min).  A frame whose smallest rating is 2 (range 2..7) gives negatives the target -0.4, not 0: the device
    sampler's targets equal the host stream's on both kinds of triple; with use_nce off both are raw (negatives 0)."""
    import torch
    from helpers import load_frames
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import DMF
    from drecpy_amd.Sampler import PointSampler
    frame = {k: v.copy() for k, v in load_frames()['pt_int_dense'].items()}
    frame['interaction'] = np.asarray(frame['interaction'], np.float64) + 2.0            # 2..7: min_interaction = 2
    ds = InteractionDataset.read_df(frame, verbose=False)
    for use_nce in (True, False):
        model = DMF(user_factors=[16, 8], item_factors=[16, 8], seed=5, verbose=False, use_nce=use_nce)
        model.fit(ds, epochs=2, batch_size=64, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=3, device_sampler=True)
        assert model.min_interaction == 2.0 and model.max_interaction == 7.0
        prep = model._engine.prepare_batch_device(20000, 3, 77)
        torch.cuda.synchronize()
        y = prep['device']['y'].cpu().numpy()
        ru, ri, rv, rneg = PointSampler(ds, 3, model.interaction_threshold, 11).sample_arrays(20000)
        rv, rneg = np.asarray(rv, np.float64), np.asarray(rneg).astype(bool)
        ry = np.asarray(model._standardize_value(rv), np.float32) if use_nce else rv.astype(np.float32)
        want_neg = np.float32((0.0 - 2.0) / 5.0) if use_nce else np.float32(0.0)
        assert np.all(ry[rneg] == want_neg)                                  # the reference stream's own negatives
        neg_d = y == want_neg
        assert abs(neg_d.mean() - rneg.mean()) < 0.015, (use_nce, neg_d.mean(), rneg.mean())
        # every other target is a positive's: the same set of values as the host stream's positives
        assert set(np.unique(y[~neg_d]).tolist()) <= set(np.unique(ry[~rneg]).tolist()), use_nce
        assert y[~neg_d].min() >= (0.0 if use_nce else 2.0)


def test_dmf_of_the_references_consistency_script_fits_and_ranks():
    """examples/consistency_eval/dmf.py:20 of the reference: DMF(user_factors=[128, 64], item_factors=[128, 64]) — fit, predict and the
    all-pairs scorer (64 final factors) through the public classes; the fused step against the oracle is test_dmf_steps_match_oracle."""
    from helpers import load_frames
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import DMF
    frame = {k: v.copy() for k, v in load_frames()['pt_int_dense'].items()}
    ds = InteractionDataset.read_df(frame, verbose=False)
    model = DMF(user_factors=[128, 64], item_factors=[128, 64], seed=10, verbose=False)
    model.fit(ds, epochs=30, batch_size=64, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5)
    assert model._engine.W == 128 and model._engine.first_layer_update == 'scatter'
    u, i = int(frame['user'][0]), int(frame['item'][0])
    p = model.predict(u, i)
    assert np.isfinite(p)
    rec = model.recommend(u, n=5)
    assert len(rec) == 5
    # the scorer's rows are 128 floats wide here (64 factors used): equal to per-pair predictions to bf16 accuracy
    e = model._engine
    sc = e.score_matrix_bf16(np.arange(4)).cpu().numpy()
    pr = e.predict(np.repeat(np.arange(4), e.N), np.tile(np.arange(e.N), 4)).cpu().numpy().reshape(4, e.N)
    assert np.max(np.abs(sc - pr)) < 1.5e-2


@pytest.mark.parametrize('seg_len', [0, 8, 64])
def test_the_device_work_list_has_the_host_lists_entries_and_classes(seg_len):
    """drx_dmf_work_order_device against drx_dmf_work_order: the same entries (work index | segment << 24) as a SET, degree classes
    (bit length of the degree) non-increasing along the list, every id's segments consecutive and ascending, the partial rows of
    different ids disjoint and as many as the host counts (inside a class the two lists may differ in order: no result depends on it)."""
    import ctypes as C
    import torch
    from drecpy_amd import _lib
    L_ = _lib.lib()
    rng = np.random.default_rng(31 + seg_len)
    U, N, B = 300, 200, 256
    csr, csc, _ = _problem(rng, U, N, 9000)
    du = np.sort(rng.choice(U, size=140, replace=False)).astype(np.int32)
    di = np.sort(rng.choice(N, size=90, replace=False)).astype(np.int32)
    ipu, ipi = np.asarray(csr[0], np.int64), np.asarray(csc[0], np.int64)
    off_u = np.concatenate([[0], np.cumsum(ipu[du + 1] - ipu[du])]).astype(np.int32)
    off_i = np.concatenate([[0], np.cumsum(ipi[di + 1] - ipi[di])]).astype(np.int32)
    cap = 2 * B + 4096
    want, zs_h, n_part = np.zeros(cap, np.int32), np.zeros(len(du) + len(di), np.int32), C.c_int32(0)
    n_h = L_.drx_dmf_work_order(off_u.ctypes.data, len(du), off_i.ctypes.data, len(di), seg_len, want.ctypes.data, cap, zs_h.ctypes.data, C.byref(n_part))
    assert n_h >= len(du) + len(di)
    dev = torch.device('cuda:0')
    t = lambda a: torch.as_tensor(a).to(dev)
    d_ipu, d_ipi, d_du, d_di = t(ipu), t(ipi), t(du), t(di)
    nd = t(np.array([len(du), len(di)], np.int32))
    order, zseg, out2 = torch.full((cap,), -1, dtype=torch.int32, device=dev), torch.zeros(2 * B, dtype=torch.int32, device=dev), torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.check(L_.drx_dmf_work_order_device(d_ipu.data_ptr(), d_ipi.data_ptr(), d_du.data_ptr(), d_di.data_ptr(), nd.data_ptr(), seg_len,
                                            order.data_ptr(), cap, zseg.data_ptr(), out2.data_ptr(), _lib.stream_ptr(dev)), 'drx_dmf_work_order_device')
    torch.cuda.synchronize()
    n_d, parts_d = out2.cpu().numpy().tolist()
    got, zs_d = order.cpu().numpy()[:n_d], zseg.cpu().numpy()[:len(du) + len(di)]
    assert n_d == n_h and parts_d == n_part.value and sorted(got.tolist()) == sorted(want[:n_h].tolist())
    deg = np.concatenate([np.diff(off_u), np.diff(off_i)])
    cls = np.array([int(d).bit_length() for d in deg])
    along = cls[got & 0xFFFFFF]
    assert np.all(np.diff(along) <= 0)                                   # longest classes first
    for i in np.unique(got & 0xFFFFFF):                                  # an id's segments: consecutive, ascending from 0
        at = np.flatnonzero((got & 0xFFFFFF) == i)
        assert np.array_equal(at, np.arange(at[0], at[0] + len(at))) and np.array_equal(got[at] >> 24, np.arange(len(at)))
        assert (zs_d[i] & 255) == len(at) - 1 == (zs_h[i] & 255)
    taken = np.zeros(max(1, parts_d), bool)
    for i in np.flatnonzero(zs_d & 255):
        lo, n = zs_d[i] >> 8, zs_d[i] & 255
        assert not taken[lo:lo + n].any()
        taken[lo:lo + n] = True
    assert taken.sum() == parts_d


@pytest.mark.parametrize('modified', [False, True])
def test_the_cached_argument_structs_of_host_batches_change_nothing(modified):
    """Steps on host-prepared batches reuse the argument structs of the slot of the upload ring they land in (four slots; a slot's device
    addresses are the same from visit to visit): the same parameters, bit for bit, as with the cache off — over more steps than the ring
    has slots, with batches of different distinct-id counts, a learning-rate change, and a loss read in between (the general path)."""
    from drecpy_amd.engine_dmf import DmfEngine
    from drecpy_amd.Recommender import Variable
    rng = np.random.default_rng(29)
    U, N, B = 150, 90, 180
    csr, csc, _ = _problem(rng, U, N, 2800)
    p = dm.init_params(rng, U, N, (32, 16), (24, 16), np.float64)
    engs = []
    for cached in (True, False):
        e = DmfEngine(U, N, (32, 16), (24, 16), True)
        e.set_interactions(csr, csc)
        if modified:
            e.bind_prediction_scale(Variable([1.0]), broadcast_targets=True)
        e.set_params(p)
        e.lr, e.reg = 2e-3, 1e-3
        e.host_step_cache = cached
        e._seg_len = 8                                   # (segments in play: n_work and the partial rows differ from batch to batch)
        engs.append(e)
    for step in range(14):
        hi = U if step % 3 else U // 5                    # some batches with few distinct users
        u = rng.integers(0, hi, size=B).astype(np.int32)
        i = rng.integers(0, N // 2, size=B).astype(np.int32)
        y = rng.random(B).astype(np.float32)
        if step == 8:
            for e in engs:
                e.lr = 5e-3
        want_loss = step == 10
        la = [e.step(step, u, i, y, want_loss=want_loss) for e in engs]
        assert la[0] == la[1]
    assert sum(c is not None for c in engs[0]._stage['cache']) == 4 and all(c is None for c in engs[1]._stage['cache'])
    ga, gb = engs[0].get_params(), engs[1].get_params()
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k
