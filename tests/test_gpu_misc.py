"""GPU parity for the integer kernels: id map (bit-exact vs the reference-generated golden codes) and top-k."""
import numpy as np
import pytest

from oracle import cdae_oracle as co
from oracle import data_oracle as do
from helpers import load_frames, load_json

pytestmark = pytest.mark.gpu


def test_idmap_device_matches_reference_codes():
    from drecpy_amd.Dataset.interaction_dataset import _first_appearance_device
    g = load_json('idmap.json')
    frames = load_frames()
    for k in ('pt_str_gapped', 'pt_int_dense', 'ls_int_ts'):
        codes, cats = _first_appearance_device(frames[k]['item'].astype(np.int64))     # int raw ids
        assert codes.tolist() == g[k]['iid']
        assert len(cats) == g[k]['n_items']
        want_codes, want_cats = do.first_appearance_codes(frames[k]['item'].tolist())
        assert cats.tolist() == list(want_cats)
    # reference resource file with int ids (tests/Dataset/resources/test_int_ids.csv)
    e = g['test_int_ids.csv']
    codes, _ = _first_appearance_device(np.array(e['user'], dtype=np.int64))
    assert codes.tolist() == e['uid']
    # large random case with heavy duplication and negative ids
    rng = np.random.default_rng(0)
    raw = rng.integers(-5000, 5000, size=300_000).astype(np.int64) * 1_000_003
    codes, cats = _first_appearance_device(raw)
    want, wcats = do.first_appearance_codes(raw.tolist())
    assert np.array_equal(codes, want) and cats.tolist() == wcats


def test_dataset_assign_ids_on_gpu_matches_golden():
    from drecpy_amd.Dataset import InteractionDataset
    g = load_json('idmap.json')
    for k, f in load_frames().items():
        ds = InteractionDataset.read_df(dict(f), verbose=False)
        ds.assign_internal_ids()
        assert ds._cols['uid'].tolist() == g[k]['uid'] and ds._cols['iid'].tolist() == g[k]['iid']
        for raw, want in g[k]['probe_user_to_uid']:
            assert ds.user_to_uid(raw) == want
        for iid, want in g[k]['probe_iid_to_item']:
            assert ds.iid_to_item(iid) == want


@pytest.mark.parametrize('n,k', [(40, 5), (1682, 10), (3706, 100), (5000, 5000), (16384, 7), (16385, 50), (100000, 1000),
                                 (1_000_000, 100), (40_000, 20_000), (20_000, 30_000),      # long rows: radix select (+ the k > 16384 ordering)
                                 # r06: k <= 64 of rows of <= 4096 scores: selection by one wave per row (16 / 32 / 64 keys per lane), more
                                 # candidates asked for than the mask leaves, a row of one score
                                 (3706, 10), (4096, 64), (1024, 1), (1025, 33), (2048, 64), (2049, 3), (63, 64), (5, 1), (700, 40), (130, 60), (3706, 65)])
def test_topk_matches_heapq(n, k):
    import torch
    from drecpy_amd.engine import CdaeEngine, pack_mask_bits
    eng = CdaeEngine(4, 8, 4)
    rng = np.random.default_rng(n)
    R = 5 if n <= 20000 else 2
    scores = rng.random((R, n)).astype(np.float32)
    scores[:, ::7] = scores[:, 3:4]                    # many exact ties -> larger index must win
    mask = rng.random((R, n)) < 0.7
    mask[0] = True
    idx, val = eng.topk(torch.as_tensor(scores).cuda(), k, torch.as_tensor(pack_mask_bits(mask).view(np.int32)).cuda())
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    for r in range(R):
        want = co.rank_row(scores[r], np.flatnonzero(mask[r]), k)
        got = [(float(v), int(i)) for v, i in zip(val[r], idx[r]) if i >= 0]
        assert got == [(float(np.float32(v)), i) for v, i in want]
        assert (idx[r] >= 0).sum() == min(k, int(mask[r].sum()))


@pytest.mark.parametrize('n,bits', [(1, 1), (63, 9), (64, 10), (4096, 18), (4097, 18), (100_003, 25), (1_440_000, 25), (300_000, 27), (70_000, 32),
                                    (200_000, 12), (500_000, 16), (33_000, 8)])      # narrow keys over many tiles: one- and two-pass plans
def test_sort_pairs_is_a_stable_sort(n, bits):
    """drx_sort_pairs (the inverted-index builder of the sparse steps) against torch's stable sort: heavy duplicates (Zipf-like
    keys), padding keys 0xFFFFFFFF, sizes around tile and wave borders, only the low `bits` bits order the pairs."""
    import ctypes as C
    import torch
    from drecpy_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device='cuda'); g.manual_seed(n + bits)
    hi = (1 << bits) - 1
    u = torch.rand(n, generator=g, device='cuda', dtype=torch.float64)
    keys64 = (u ** 6 * hi).to(torch.int64)                                 # skewed: many duplicates of small keys
    if bits == 25 or bits == 32:
        pad = torch.rand(n, generator=g, device='cuda') < 0.15
        keys64 = torch.where(pad, torch.full_like(keys64, 0xFFFFFFFF), keys64)
    keys64 = keys64 & 0xFFFFFFFF
    keys = (keys64 - ((keys64 >> 31) << 32)).to(torch.int32)               # uint32 bit pattern in an int32 tensor
    vals = torch.arange(n, device='cuda', dtype=torch.int32)
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    need = L.drx_sort_pairs_temp_bytes(n, bits)
    tmp = torch.empty(need, dtype=torch.uint8, device='cuda')
    _lib.check(L.drx_sort_pairs(_lib.ptr(keys), _lib.ptr(ko), _lib.ptr(vals), _lib.ptr(vo), n, bits, _lib.ptr(tmp), need,
                                _lib.stream_ptr(torch.device('cuda'))), 'drx_sort_pairs')
    torch.cuda.synchronize()
    low = keys64 & hi                                                       # only the low `bits` bits order the pairs
    want = torch.sort(low, stable=True)
    assert torch.equal(vo.long(), want.indices)
    assert torch.equal(ko.long() & 0xFFFFFFFF, keys64[want.indices])


@pytest.mark.parametrize('T,n_rows,ld,indexed', [(1, 5, 4, False), (700, 300, 52, False), (6144, 3706, 100, False),
                                                  (2560, 3706, 52, True), (40_000, 3706, 64, True), (200_000, 50_000, 128, True),
                                                  (70, 9, 300, True)])
def test_scatter_rows_both_paths_against_numpy(T, n_rows, ld, indexed):
    """drx_scatter_rows (the gradient of an embedding lookup) against a float64 statement: small problems take the bit-mask path
    (a bit per destination row and touch, rows walk their bits in touch order), larger ones the stable sort + segmented
    reduction; padding keys are ignored, rows nobody names keep what they held."""
    import torch
    from drecpy_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(T + n_rows)
    n_src = max(T // 3, 1) if indexed else T
    keys = rng.integers(0, n_rows, size=T).astype(np.int64)
    keys[rng.random(T) < 0.5] = rng.integers(0, min(n_rows, 7))                 # a few hot rows
    pad = rng.random(T) < 0.1
    src = rng.standard_normal((n_src, ld)).astype(np.float32)
    src_s = rng.standard_normal(n_src).astype(np.float32)
    idx = rng.integers(0, n_src, size=T).astype(np.int64) if indexed else np.arange(T)
    coef = rng.standard_normal(T).astype(np.float32) if indexed else np.ones(T, np.float32)
    want, want_s = np.full((n_rows, ld), 7.0), np.full(n_rows, 7.0)             # 7 = "kept what it held"
    named = np.unique(keys[~pad])
    want[named], want_s[named] = 0.0, 0.0
    np.add.at(want, keys[~pad], coef[~pad, None].astype(np.float64) * src[idx[~pad]])
    np.add.at(want_s, keys[~pad], coef[~pad].astype(np.float64) * src_s[idx[~pad]])
    dev = torch.device('cuda')
    k32 = np.where(pad, 0xFFFFFFFF, keys).astype(np.uint32).view(np.int32)
    t = lambda a: torch.as_tensor(a).to(dev)
    d_keys, d_src, d_ss = t(k32), t(src), t(src_s)
    d_idx = t(idx.astype(np.int32)) if indexed else None
    d_coef = t(coef) if indexed else None
    out = torch.full((n_rows, ld), 7.0, dtype=torch.float32, device=dev)
    out_s = torch.full((n_rows,), 7.0, dtype=torch.float32, device=dev)
    need = L.drx_scatter_scratch_bytes(ld, T, n_rows)
    scratch = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.check(L.drx_scatter_rows(_lib.ptr(d_keys), T, _lib.ptr(d_src), _lib.ptr(d_idx), _lib.ptr(d_coef), _lib.ptr(d_ss), ld, n_rows,
                                  _lib.ptr(out), _lib.ptr(out_s), _lib.ptr(scratch), need, _lib.stream_ptr(dev)), 'drx_scatter_rows')
    torch.cuda.synchronize()
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(out_s.cpu().numpy(), want_s, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want_s).max())))


def test_run_ahead_streams_come_from_one_probed_pool():
    """engine.run_ahead_stream: the k-th run-ahead stream of the PROCESS (DMF, Caser, the device samplers and the row layout's side stream
    share them — a process that uses few streams keeps each on a hardware queue of its own), probed when created: the round trip of the
    run-ahead pattern with the training stream (_ping_pong_us) is that of a stream with its own queue."""
    import torch
    from drecpy_amd import engine
    dev = torch.device('cuda:0')
    a, b = engine.run_ahead_stream(dev, 0), engine.run_ahead_stream(dev, 1)
    assert a is engine.run_ahead_stream(dev, 0) and b is engine.run_ahead_stream(dev, 1) and a.cuda_stream != b.cuda_stream
    main = torch.cuda.current_stream(dev)
    assert a.cuda_stream != main.cuda_stream
    us = min(engine._ping_pong_us(main, a), engine._ping_pong_us(main, a))
    assert us < 2 * engine.STREAM_PROBE_US, us            # (a stream on the training stream's queue: 90 - 170 us on the boxes measured)
    x = torch.zeros(8, device=dev)
    with engine._on_stream(a):
        x.add_(1)
    torch.cuda.synchronize()
    assert float(x.sum()) == 8.0 and torch.cuda.current_stream(dev).cuda_stream == main.cuda_stream
