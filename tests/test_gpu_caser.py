"""GPU parity of the Caser step (drx_caser_* + drx_scatter_rows + drx_adam_*) against oracle/caser_oracle.py."""
import numpy as np
import pytest

from oracle import caser_oracle as ca

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / np.maximum(np.abs(b), 1e-12)))


@pytest.mark.parametrize('d,L,n_v,n_h,T,neg,B,drop', [(50, 5, 4, 16, 3, 3, 64, False), (16, 3, 2, 8, 2, 2, 37, True),
                                                      (64, 5, 4, 16, 3, 3, 300, True)])
def test_caser_steps_match_oracle(d, L, n_v, n_h, T, neg, B, drop):
    from drecpy_amd.engine_caser import CaserEngine
    rng = np.random.default_rng(d + B)
    U, N = 40, 150
    p = ca.init_params(rng, U, N, L, d, n_v, n_h, np.float64)
    eng = CaserEngine(U, N, L, T, neg, d, n_v, n_h)
    eng.set_params(p)
    eng.lr, eng.reg = 5e-3, 1e-4
    st = ca.adam_state(p)
    g0 = eng.get_params()
    for k in p:
        np.testing.assert_allclose(g0[k], p[k].astype(np.float32), rtol=0, atol=0)
    nx = n_v + L * n_h
    for step in range(6):
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
