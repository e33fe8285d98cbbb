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


@pytest.mark.parametrize('uf,itf,l2n,B', [((64, 32), (64, 32), True, 64), ((16,), (24, 16), True, 33), ((32, 20, 8), (12, 8), False, 50)])
def test_dmf_steps_match_oracle(uf, itf, l2n, B):
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
    _lib.check(_lib.lib().drx_score_pairs_bf16(_lib.ptr(a), 70, _lib.ptr(b), 45, 64, 32, _lib.ptr(out), _lib.stream_ptr()), 'score')
    want = torch.clamp(a[:, :32] @ b[:, :32].t(), min=1e-6)
    assert torch.equal(out, want)
