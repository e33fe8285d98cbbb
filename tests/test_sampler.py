"""Host-side sampler / RNG of libdrx.so against stdlib random (CPython's MT19937) and the golden streams recorded
from the reference (tests/golden/point_sampler.json).  No GPU needed: these entry points are host code."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from helpers import load_frames, load_json


def _ds(frame):
    from drecpy_amd.Dataset import InteractionDataset
    ds = InteractionDataset.read_df({k: v for k, v in frame.items()}, verbose=False)
    ds.assign_internal_ids()
    return ds


def test_rng_matches_cpython():
    from drecpy_amd import _lib
    L = _lib.lib()
    for seed in (0, 1, 10, 23, 2 ** 31, 2 ** 40 + 17, -5):
        r = L.drx_rng_create(seed)
        py = random.Random(seed)
        for _ in range(700):                       # crosses the 624-word regeneration boundary
            assert L.drx_rng_random(r) == py.random()
        for hi in (0, 1, 2, 5, 942, 1681, 10 ** 6, 2 ** 31 - 1, 2 ** 32, 2 ** 40 + 3):
            for _ in range(20):
                assert L.drx_rng_randint(r, 0, hi) == py.randint(0, hi)
        L.drx_rng_destroy(r)


def test_point_sampler_streams_match_reference():
    from drecpy_amd.Sampler import PointSampler
    ps = load_json('point_sampler.json')
    frames = load_frames()
    checked = 0
    for key, want in ps.items():
        k = key.split('|')[0]
        if k not in frames:
            continue
        _, neg, thr, seed = key.split('|')
        thr = None if thr == 'None' else float(thr)
        s = PointSampler(_ds(frames[k]), int(neg), thr, int(seed))
        got = s.sample(len(want) // 2) + s.sample(len(want) - len(want) // 2)
        assert [[int(a), int(b), int(c)] for a, b, c in got] == want, key
        checked += len(want)
    assert checked >= 1500


def test_point_sampler_resource_csv_stream():
    # stream captured from the reference on tests/Dataset/resources/test.csv
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Sampler import PointSampler
    g = load_json('idmap.json')['test.csv']
    ds = InteractionDataset.read_df({'user': np.array(g['user'], dtype=object), 'item': np.array(g['item'], dtype=object),
                                     'interaction': np.array(g['interaction'])}, verbose=False)
    ds.assign_internal_ids()
    assert ds._cols['uid'].tolist() == g['uid'] and ds._cols['iid'].tolist() == g['iid']
    s = PointSampler(ds, 5, 0.001, 10)
    got = [[int(a), int(b), int(c)] for a, b, c in s.sample(64)]
    assert got == load_json('point_sampler.json')['test.csv|5|0.001|10']
    # reference known answers (tests/Dataset/test_mem_dataset.py:678-697): seed 23 -> (1,0),(0,2),(1,3),(0,1)
    s = PointSampler(ds, 5, None, 23)
    assert [s.sample_negative()[:2] for _ in range(4)] == [(1, 0), (0, 2), (1, 3), (0, 1)]
    gen = ds.null_interaction_pair_generator(seed=23)
    assert [next(gen) for _ in range(4)] == [(1, 0), (0, 2), (1, 3), (0, 1)]
    s = PointSampler(ds, 5, None, 23)
    want = load_json('point_sampler.json')['test.csv|select_random_gen|seed23']
    assert [[int(x) for x in s.sample_positive()] for _ in range(16)] == want
    gen = ds.select_random_generator(seed=23)
    assert [[r['uid'], r['iid'], int(r['interaction'])] for r in (next(gen) for _ in range(16))] == want


def test_corruption_stream_matches_python_rng():
    from drecpy_amd import _lib
    from oracle import data_oracle as do
    L = _lib.lib()
    rng = np.random.default_rng(0)
    U, N, B, q = 30, 47, 12, 0.2
    indptr = np.zeros(U + 1, dtype=np.int64)
    idx = []
    for u in range(U):
        c = np.sort(rng.choice(N, size=rng.integers(0, 9), replace=False))
        idx.append(c); indptr[u + 1] = indptr[u] + len(c)
    indices = np.concatenate(idx).astype(np.int32)
    uids = rng.integers(0, U, size=B).astype(np.int32)
    r = L.drx_rng_create(10)
    keep_off = np.zeros(B + 1, dtype=np.int32)
    keep = np.zeros(int(indptr[-1]) * 2 + 8, dtype=np.uint8)
    assert L.drx_rng_corruption_keep(r, indptr.ctypes.data, indices.ctypes.data, N, uids.ctypes.data, B, q,
                                     keep_off.ctypes.data, keep.ctypes.data, len(keep)) == 0
    want = do.corruption_keep_mask(random.Random(10), B, N, q)       # cdae.py:63 stream, N draws per row
    for b, u in enumerate(uids):
        cols = indices[indptr[u]:indptr[u + 1]]
        assert keep[keep_off[b]:keep_off[b + 1]].astype(bool).tolist() == want[b, cols].tolist()
    # the next draw continues the same MT stream
    py = random.Random(10)
    for _ in range(B * N):
        py.random()
    assert L.drx_rng_random(r) == py.random()
    L.drx_rng_destroy(r)


def test_two_generators_leapfrogging_the_corruption_stream_equal_one():
    """What CDAE.fit() does on its two worker threads: batches drawn alternately from two generators of the same seed, each
    advanced (drx_rng_discard) to the word where its batch begins — same masks as one generator drawing every batch, for a
    stream long enough to cross many 624-word state blocks."""
    from drecpy_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    U, N, B, q = 40, 1000, 16, 0.2
    indptr = np.zeros(U + 1, dtype=np.int64)
    idx = []
    for u in range(U):
        c = np.sort(rng.choice(N, size=rng.integers(1, 30), replace=False))
        idx.append(c); indptr[u + 1] = indptr[u] + len(c)
    indices = np.concatenate(idx).astype(np.int32)
    batches = [rng.integers(0, U, size=B).astype(np.int32) for _ in range(7)]

    def draw(r, uids):
        keep_off = np.zeros(B + 1, dtype=np.int32)
        keep = np.zeros(int(indptr[-1]) * 2 + 8, dtype=np.uint8)
        assert L.drx_rng_corruption_keep(r, indptr.ctypes.data, indices.ctypes.data, N, uids.ctypes.data, B, q,
                                         keep_off.ctypes.data, keep.ctypes.data, len(keep)) == 0
        return keep[:keep_off[-1]].copy()

    one = L.drx_rng_create(77)
    want = [draw(one, u) for u in batches]
    two, at = [L.drx_rng_create(77), L.drx_rng_create(77)], [0, 0]
    for t, u in enumerate(batches):
        g, begin = t % 2, t * 2 * N * B
        L.drx_rng_discard(two[g], begin - at[g])
        assert np.array_equal(draw(two[g], u), want[t])
        at[g] = begin + 2 * N * B
    py = random.Random(77)                                             # and both agree with CPython on where the stream is
    for _ in range(len(batches) * N * B):
        py.random()
    assert L.drx_rng_random(one) == py.random()
    for r in [one] + two:
        L.drx_rng_destroy(r)


def test_native_draw_ahead_workers_reproduce_the_sequential_streams():
    """drx_drawahead_*: the two native worker threads of reference-mode fit() (sampler triples in ticket order, each worker's
    corruption generator advanced past the other's batches) against one sampler + one generator drawing batch after batch."""
    from drecpy_amd import _lib
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Sampler import PointSampler
    L = _lib.lib()
    rng = np.random.default_rng(5)
    U, N, B, q, T = 60, 500, 32, 0.2, 11
    rows = {(int(u), int(i)) for u, i in zip(rng.integers(0, U, 1500), rng.integers(0, N, 1500))}
    frame = {'user': [r[0] for r in rows], 'item': [r[1] for r in rows], 'interaction': [1 + (r[0] + r[1]) % 5 for r in rows]}
    ds = InteractionDataset.read_df(frame, verbose=False)
    ds.assign_internal_ids()
    indptr, indices = ds.positives_csr(1e-3)
    n_items, max_deg = ds.count_unique('iid'), int(np.diff(indptr).max())

    def buffers():
        return (np.empty(B, np.int32), np.empty(B, np.int32), np.empty(B, np.float64), np.empty(B, np.uint8), np.empty(B + 1, np.int32),
                np.empty(B * max_deg, np.uint8))

    # sequential statement: one sampler, one generator
    s1, g1, want = PointSampler(ds, 5, 1e-3, 10), L.drx_rng_create(10), []
    for t in range(T):
        u, i, v, ng = s1.sample_arrays(B)
        ko, kp = np.zeros(B + 1, np.int32), np.zeros(B * max_deg, np.uint8)
        assert L.drx_rng_corruption_keep(g1, indptr.ctypes.data, indices.ctypes.data, n_items, u.ctypes.data, B, q, ko.ctypes.data,
                                         kp.ctypes.data, len(kp)) == 0
        want.append((u.copy(), i.copy(), v.copy(), ng.copy(), ko.copy(), kp[:ko[-1]].copy()))
    # the workers: up to four jobs in flight, submitted in ticket order, finished in ticket order
    s2, gens, at = PointSampler(ds, 5, 1e-3, 10), [L.drx_rng_create(10), L.drx_rng_create(10)], [0, 0]
    h = L.drx_drawahead_create(s2._host._h, gens[0], gens[1], indptr.ctypes.data, indices.ctypes.data, n_items)
    assert h
    flight, got = [], []

    def finish():
        g, job, buf = flight.pop(0)
        assert L.drx_drawahead_wait(h, g, job) == 0
        got.append((buf[0].copy(), buf[1].copy(), buf[2].copy(), buf[3].copy(), buf[4].copy(), buf[5][:buf[4][-1]].copy()))

    for t in range(T):
        g, begin, buf = t % 2, t * 2 * n_items * B, buffers()
        job = L.drx_drawahead_submit(h, g, t, begin - at[g], B, q, *[b.ctypes.data for b in buf], len(buf[5]))
        assert job >= 0
        at[g] = begin + 2 * n_items * B
        flight.append((g, job, buf))
        if len(flight) == 4:
            finish()
    while flight:
        finish()
    L.drx_drawahead_destroy(h)
    for a, b in zip(want, got):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    for r in [g1] + gens:
        L.drx_rng_destroy(r)


def test_library_exports_every_declared_symbol():
    import re, os
    from drecpy_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'drx.h')).read()
    declared = set(re.findall(r'\b(drx_[a-z0-9_]+)\s*\(', hdr))
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_list_sampler_streams_match_reference():
    """Caser's configuration of ListSampler (caser.py:72-75) against streams recorded from the reference."""
    from drecpy_amd.Sampler import ListSampler
    ls = load_json('list_sampler.json')
    ds = _ds(load_frames()['ls_int_ts'])
    for key, want in ls.items():
        L, T, neg, thr, seed = key.split('|')
        s = ListSampler(ds, ['uid'], neg_ratio=int(neg), n_targets=int(T), interaction_threshold=float(thr),
                        negative_ids_col='iid', min_positive_records=int(L), max_positive_records=int(L),
                        sort_column='timestamp', seed=int(seed))
        got = s.sample_group_records(len(want))
        for (b, a, ng), w in zip(got, want):
            assert [int(r['rid']) for r in b] == w['before_rid'] and [int(r['rid']) for r in a] == w['after_rid']
            assert [int(r['iid']) for r in b] == w['before'] and [int(r['iid']) for r in a] == w['after']
            assert int(b[0]['uid']) == w['uid'] and [int(x) for x in ng] == w['neg'], key


@pytest.mark.parametrize('cfg', [dict(n_targets=3, min_positive_records=5, max_positive_records=5, neg_ratio=3, sort_column='timestamp'),
                                 dict(n_targets=2, min_positive_records=1, max_positive_records=None, neg_ratio=2, sort_column=None),
                                 dict(n_targets=None, min_positive_records=2, max_positive_records=4, neg_ratio=1, sort_column='timestamp'),
                                 dict(n_targets=1, min_positive_records=3, max_positive_records=6, neg_ratio=9, sort_column='timestamp',
                                      interaction_threshold=3)])
def test_native_list_sampler_equals_the_python_loop(cfg):
    """drx_list_sampler_* (C++: MT19937 stream + CPython's set iteration order for the eligible negatives) against the
    Python loop on the same dataset and seed — including users who hold more than a quarter of all items, for whom
    set.difference rebuilds the set by insertion and the tuple order is NOT ascending."""
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Sampler import ListSampler
    rng = np.random.default_rng(3)
    U, N = 60, 90
    rows = []
    for u in range(U):
        deg = int(rng.integers(1, 12)) if u % 7 else int(rng.integers(30, 80))        # every 7th user is a heavy one
        for i in rng.choice(N, size=deg, replace=False):
            rows.append((u + 100, int(i) * 3 + 7, int(rng.integers(1, 6)), int(rng.integers(0, 10 ** 6))))
    rows = [rows[j] for j in rng.permutation(len(rows))]
    cols = list(zip(*rows))
    ds = InteractionDataset.read_df({'user': np.array(cols[0]), 'item': np.array(cols[1]), 'interaction': np.array(cols[2]),
                                     'timestamp': np.array(cols[3])}, verbose=False)
    ds.assign_internal_ids()
    a = ListSampler(ds, ['uid'], seed=41, **cfg)
    b = ListSampler(ds, ['uid'], seed=41, **cfg)
    assert a._native is not None
    b._native = None                                                                  # force the Python loop
    for n in (1, 7, 64, 200):
        ra, rb = a.sample_group_records(n), b.sample_group_records(n)
        assert len(ra) == len(rb) == n
        for xa, xb in zip(ra, rb):
            if cfg['n_targets'] is None:
                assert [r['rid'] for r in xa] == [r['rid'] for r in xb]
            else:
                assert [r['rid'] for r in xa[0]] == [r['rid'] for r in xb[0]]
                assert [r['rid'] for r in xa[1]] == [r['rid'] for r in xb[1]]
                assert [int(v) for v in xa[2]] == [int(v) for v in xb[2]]
                assert all(k == kb and va == vb for (k, va), (kb, vb) in zip(sorted(xa[0][0].items()), sorted(xb[0][0].items())))


def test_native_list_sampler_large_batches_equal_small_ones():
    """A batch of >= 2048 windows finishes its second phase (indices -> ids, row copies) on helper threads; the stream is sequential, so
    one call for 3000 windows must equal six calls for 500 each (which stay on the calling thread) bit for bit — heavy users (ids
    looked up in a set order) included."""
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Sampler import ListSampler
    rng = np.random.default_rng(11)
    U, N = 80, 120
    rows = []
    for u in range(U):
        deg = int(rng.integers(9, 25)) if u % 9 else int(rng.integers(40, 100))
        for i in rng.choice(N, size=deg, replace=False):
            rows.append((u, int(i), 1, int(rng.integers(0, 10 ** 6))))
    cols = list(zip(*rows))
    ds = InteractionDataset.read_df({'user': np.array(cols[0]), 'item': np.array(cols[1]), 'interaction': np.array(cols[2]),
                                     'timestamp': np.array(cols[3])}, verbose=False)
    ds.assign_internal_ids()
    cfg = dict(n_targets=3, min_positive_records=5, max_positive_records=5, neg_ratio=3, sort_column='timestamp', negative_ids_col='iid')
    a, b = ListSampler(ds, ['uid'], seed=5, **cfg), ListSampler(ds, ['uid'], seed=5, **cfg)
    assert a._native is not None
    for _ in range(2):                                   # (the second round reuses the helper threads)
        big = a.sample_group_arrays(3000)
        parts = [b.sample_group_arrays(500) for _ in range(6)]
        assert np.array_equal(big[0], np.concatenate([p[0] for p in parts]))                     # groups
        for k in (2, 4, 6):                                                                      # input rows, target rows, negative ids
            assert np.array_equal(big[k], np.concatenate([p[k] for p in parts])), k


def test_native_list_sampler_large_batch_in_a_forked_child():
    """The helper threads of a large batch belong to the process that made them: a forked child draws its own large batch with new
    ones instead of waiting for threads that do not exist there (a child process of its own, with a timeout)."""
    import subprocess
    import sys
    code = """
import os, sys
sys.path.insert(0, %r)
import numpy as np
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Sampler import ListSampler
rng = np.random.default_rng(1)
rows = [(u, int(i), 1, int(rng.integers(0, 10 ** 6))) for u in range(80) for i in rng.choice(120, size=20, replace=False)]
c = list(zip(*rows))
ds = InteractionDataset.read_df({'user': np.array(c[0]), 'item': np.array(c[1]), 'interaction': np.array(c[2]), 'timestamp': np.array(c[3])},
                                verbose=False)
ds.assign_internal_ids()
s = ListSampler(ds, ['uid'], seed=5, n_targets=3, min_positive_records=5, max_positive_records=5, neg_ratio=3, sort_column='timestamp',
                negative_ids_col='iid')
s.sample_group_arrays(3000)
pid = os.fork()
if pid == 0:
    out = s.sample_group_arrays(3000)
    os._exit(0 if len(out[0]) == 3000 else 1)
_, st = os.waitpid(pid, 0)
sys.exit(os.WEXITSTATUS(st))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    # ... and a child that only DROPS the sampler it inherited (drx_list_sampler_destroy joined threads that do not exist
    # in the child); the child leaves through sys.exit, so every finaliser runs
    drop = code.replace("    out = s.sample_group_arrays(3000)\n    os._exit(0 if len(out[0]) == 3000 else 1)\n",
                        "    del s\n    import gc\n    gc.collect()\n    sys.exit(0)\n")
    assert 'del s' in drop
    out = subprocess.run([sys.executable, '-c', drop], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]


def test_native_list_sampler_gives_up_like_the_python_loop():
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Sampler import ListSampler
    ds = InteractionDataset.read_df({'user': np.array([1, 1, 2]), 'item': np.array([5, 6, 5]), 'interaction': np.array([1, 1, 1])},
                                    verbose=False)
    ds.assign_internal_ids()
    s = ListSampler(ds, ['uid'], n_targets=2, min_positive_records=4, seed=1)
    assert s._native is not None
    with pytest.raises(Exception, match='max consecutive tries reached'):
        s.sample_group_records(3)


def test_c_program_links_and_runs_against_the_abi(tmp_path):
    """include/drx.h is a C header and libdrx.so a plain shared library: a C program (no Python, no torch, no GPU) builds with
    gcc, links, and gets the CPython-exact random stream."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'drecpy_amd', 'libdrx.so')
    if shutil.which('gcc') is None or not os.path.exists(lib):
        pytest.skip('gcc or libdrx.so not available')
    exe = str(tmp_path / 'abi_smoke')
    subprocess.run(['gcc', '-std=c11', '-Wall', '-Werror', '-I', os.path.join(root, 'include'), os.path.join(root, 'tests', 'c', 'abi_smoke.c'),
                    '-o', exe, '-L', os.path.dirname(lib), '-ldrx', '-Wl,-rpath,' + os.path.dirname(lib)], check=True)
    env = dict(os.environ)
    try:
        import torch
        env['LD_LIBRARY_PATH'] = os.path.join(os.path.dirname(torch.__file__), 'lib') + ':/opt/rocm/lib:' + env.get('LD_LIBRARY_PATH', '')
    except ImportError:
        pass
    out = subprocess.run([exe], check=True, capture_output=True, text=True, env=env).stdout.split('\n')
    r = random.Random(10)
    a, b, c = r.random(), r.random(), r.randint(0, 9)
    got = out[0].split()
    assert float(got[0]) == a and float(got[1]) == b and int(got[2]) == c          # %.17g round-trips a double exactly
    assert out[1] == 'ok'


def test_point_sampler_membership_by_rows_when_the_pair_bitmap_would_be_too_large():
    """Above 2^28 (user, item) cells the native sampler looks a pair up in its user's sorted item row instead of a bitmap: the stream
    must still be the oracle's (mem_dataset.py:154-163)."""
    from oracle import data_oracle as do
    from drecpy_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(4)
    U, N, n = 70001, 4001, 3000                        # 2.8e8 cells
    uid = np.concatenate([rng.integers(0, 50, n), [U - 1]]).astype(np.int32)      # dense among the first users: rejections happen
    iid = np.concatenate([rng.integers(0, 60, n), [N - 1]]).astype(np.int32)
    val = rng.integers(0, 6, n + 1).astype(np.float64)
    h = L.drx_sampler_create(uid.ctypes.data, iid.ctypes.data, val.ctypes.data, len(uid), 3, 1, 1.0, 11)
    assert h
    u = np.empty(400, np.int32); i = np.empty(400, np.int32); v = np.empty(400, np.float64); ng = np.empty(400, np.uint8)
    assert L.drx_sampler_draw(h, 0, 400, u.ctypes.data, i.ctypes.data, v.ctypes.data, ng.ctypes.data) == 0
    L.drx_sampler_destroy(h)
    want = do.PointSamplerOracle(uid, iid, val, 3, 1.0, 11).sample(400)
    assert [(int(a), int(b), float(c) if not g else 0.0) for a, b, c, g in zip(u, i, v, ng)] == [(a, b, float(c)) for a, b, c in want]
    seen = set(zip(uid.tolist(), iid.tolist()))
    assert all((int(a), int(b)) not in seen for a, b, g in zip(u, i, ng) if g)


@pytest.mark.parametrize('n,B', [(6040, 4096), (3706, 256), (100000, 4096), (50, 200), (7, 1)])
def test_batch_distinct_matches_numpy(n, B):
    """drx_batch_distinct (the DMF step's host bookkeeping) against np.unique; the caller's scratch comes back clean."""
    from drecpy_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(n + B)
    ids = rng.integers(0, n, B).astype(np.int32)
    ip = np.zeros(n + 1, np.int64)
    ip[1:] = np.cumsum(rng.integers(0, 30, n))
    sc = np.full(n, -1, np.int32)
    d, inv, gptr, grows, off = (np.empty(B, np.int32), np.empty(B, np.int32), np.empty(B + 1, np.int32), np.empty(B, np.int32),
                                np.empty(B + 1, np.int32))
    nd = L.drx_batch_distinct(ids.ctypes.data, B, n, ip.ctypes.data, sc.ctypes.data, d.ctypes.data, inv.ctypes.data, gptr.ctypes.data,
                              grows.ctypes.data, off.ctypes.data)
    ud, uinv = np.unique(ids, return_inverse=True)
    assert nd == len(ud) and np.array_equal(d[:nd], ud) and np.array_equal(inv, uinv) and (sc == -1).all()
    for j in range(nd):
        assert np.array_equal(grows[gptr[j]:gptr[j + 1]], np.flatnonzero(ids == ud[j]))
    assert off[0] == 0 and np.array_equal(off[1:nd + 1], np.cumsum(ip[ud + 1] - ip[ud]))
    bad = ids.copy()
    bad[B // 2] = n                                    # an id outside the table: refused, scratch untouched
    assert L.drx_batch_distinct(bad.ctypes.data, B, n, ip.ctypes.data, sc.ctypes.data, d.ctypes.data, inv.ctypes.data, gptr.ctypes.data,
                                grows.ctypes.data, off.ctypes.data) == -1          # DRX_EINVAL
    assert (sc == -1).all()


@pytest.mark.parametrize('n,T', [(3706, 20480), (6040, 512), (5, 0), (9, 40)])
def test_batch_csr_is_a_stable_counting_sort(n, T):
    """drx_batch_csr (the Caser step's host bookkeeping): lookups grouped by row, a row's lookups in batch order."""
    from drecpy_amd import _lib
    L = _lib.lib()
    keys = np.random.default_rng(n + T).integers(0, n, T).astype(np.int32)
    ptr, order = np.empty(n + 1, np.int32), np.empty(max(T, 1), np.int32)
    assert L.drx_batch_csr(keys.ctypes.data, T, n, ptr.ctypes.data, order.ctypes.data) == 0
    assert np.array_equal(ptr, np.concatenate([[0], np.cumsum(np.bincount(keys, minlength=n))]))
    assert np.array_equal(order[:T], np.argsort(keys, kind='stable'))
    if T:
        keys[T // 2] = n
        assert L.drx_batch_csr(keys.ctypes.data, T, n, ptr.ctypes.data, order.ctypes.data) == -1      # DRX_EINVAL


def test_counter_based_list_sampler_restatement_draws_valid_windows():
    """oracle/data_oracle.py::list_sample_counter (the CPU statement of the device list sampler's throughput mode): every draw is a
    run of L + T consecutive records of one group in timestamp order, its negatives are distinct ids the group does not hold."""
    from helpers import load_frames
    from oracle import data_oracle as do
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Sampler import ListSampler
    frame = {k: v.copy() for k, v in load_frames()['ls_int_ts'].items()}
    ds = InteractionDataset.read_df(frame, verbose=False)
    ds.assign_internal_ids()
    L, T, neg = 4, 2, 3
    s = ListSampler(ds, ['uid'], neg_ratio=neg, n_targets=T, interaction_threshold=1e-3, negative_ids_col='iid', min_positive_records=L,
                    max_positive_records=L, sort_column='timestamp', seed=10)
    tw = s.twin_host_arrays()
    g, b, a = do.list_sample_counter(tw, 150, L, T, neg, 4242)
    uid, iid, ts = ds._cols['uid'], ds._cols['iid'], ds._cols['timestamp']
    seen_groups = set()
    for d in range(150):
        rows = np.flatnonzero((uid == g[d]) & (ds._cols['interaction'] >= 1e-3))
        seq = iid[rows[np.argsort(ts[rows], kind='stable')]]
        w = np.concatenate([b[d], a[d, :T]])
        assert any(np.array_equal(seq[i:i + L + T], w) for i in range(len(seq) - L - T + 1)), d
        negs = a[d, T:].tolist()
        assert len(set(negs)) == T * neg and not (set(negs) & set(seq.tolist()))
        seen_groups.add(int(g[d]))
    assert len(seen_groups) > 20            # (uniform over the eligible groups, not stuck on one)
    g2, b2, a2 = do.list_sample_counter(tw, 150, L, T, neg, 4242)
    assert np.array_equal(a, a2) and np.array_equal(b, b2)


def test_counter_based_list_sampler_has_the_reference_samplers_distribution():
    """The throughput-mode list sampler is a named deviation in HOW it draws (a counter-based generator, not the MT19937 stream), not in
    WHAT: over many draws its groups, window starts and negative ids are distributed like the reference-exact sampler's (the C++ twin of
    list_sampler.py, itself pinned to vectors recorded from the reference)."""
    from helpers import load_frames
    from oracle import data_oracle as do
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Sampler import ListSampler
    frame = {k: v.copy() for k, v in load_frames()['ls_int_ts'].items()}
    ds = InteractionDataset.read_df(frame, verbose=False)
    ds.assign_internal_ids()
    L, T, neg, n = 3, 1, 2, 12000
    mk = lambda: ListSampler(ds, ['uid'], neg_ratio=neg, n_targets=T, interaction_threshold=1e-3, negative_ids_col='iid',
                             min_positive_records=L, max_positive_records=L, sort_column='timestamp', seed=5)
    s = mk()
    tw = s.twin_host_arrays()
    g_c, b_c, a_c = do.list_sample_counter(tw, n, L, T, neg, 777)
    grp, in_off, in_rows, tg_off, tg_rows, ng_off, negs = mk().sample_group_arrays(n)
    iid = ds._cols['iid']
    n_groups, n_ids = len(tw['group_value']), tw['n_ids']
    elig = set(tw['group_value'][tw['eligible']].tolist())
    assert set(np.unique(grp).tolist()) <= elig and set(np.unique(g_c).tolist()) <= elig

    def close(x, y, bins, what):
        hx, hy = np.bincount(x, minlength=bins).astype(float), np.bincount(y, minlength=bins).astype(float)
        live = (hx + hy) > 0
        # two-sample chi-square statistic per degree of freedom: ~1 for samples of one distribution
        chi = float((((hx - hy) ** 2) / (hx + hy))[live].sum() / max(1, live.sum() - 1))
        assert chi < 1.6, (what, chi)
    close(np.asarray(grp, np.int64), g_c.astype(np.int64), int(max(grp.max(), g_c.max())) + 1, 'groups')
    first_ref = np.asarray(iid[in_rows], np.int64).reshape(n, L)[:, 0]
    close(first_ref, b_c[:, 0].astype(np.int64), n_ids, 'first input id (group x window start)')
    close(np.asarray(negs, np.int64), a_c[:, T:].reshape(-1).astype(np.int64), n_ids, 'negative ids')


def test_dmf_work_order_puts_the_longest_rows_first_and_cuts_them():
    """drx_dmf_work_order (include/drx.h DrxDmfArgs::work_order / zseg): the gather's work items — every distinct id, classes of the
    degree's bit length descending, stable inside a class, a row of more than seg_len non-zeros as ceil(deg / seg_len) entries
    (segment << 24), its partial rows numbered consecutively in work-index order."""
    import ctypes as C
    from drecpy_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for n_u, n_i, seg in ((50, 70, 1024), (1, 0, 0), (0, 3, 16), (4096, 3000, 1024), (40, 40, 0)):
        du, di = rng.integers(0, 3000, n_u), rng.integers(0, 7000, n_i)
        ou, oi = np.zeros(n_u + 1, np.int32), np.zeros(n_i + 1, np.int32)
        ou[1:], oi[1:] = np.cumsum(du), np.cumsum(di)
        deg = np.concatenate([du, di])
        nseg = np.where((seg > 0) & (deg > seg), -(-deg // max(seg, 1)), 1) if seg else np.ones(len(deg), np.int64)
        if nseg.max(initial=1) > 255:
            continue
        cap = int(nseg.sum())
        out, zseg, n_part = np.full(cap, -1, np.int32), np.full(n_u + n_i, -1, np.int32), C.c_int32(-1)
        n = L.drx_dmf_work_order(ou.ctypes.data, n_u, oi.ctypes.data, n_i, seg, out.ctypes.data, cap, zseg.ctypes.data, C.byref(n_part))
        assert n == cap and n_part.value == int((nseg - 1).sum())
        ids, segs = out & 0xFFFFFF, out >> 24
        cls = [int(deg[i]).bit_length() for i in ids]
        assert all(cls[i] >= cls[i + 1] for i in range(len(cls) - 1))
        for i in range(n_u + n_i):                          # every id: its segments 0 .. nseg - 1, in order, adjacent
            at = np.flatnonzero(ids == i)
            assert segs[at].tolist() == list(range(int(nseg[i]))) and (len(at) == 1 or np.all(np.diff(at) == 1))
        first = 0
        for i in range(n_u + n_i):
            assert zseg[i] == (((first << 8) | int(nseg[i] - 1)) if nseg[i] > 1 else 0)
            first += int(nseg[i] - 1)
        if cap > n_u + n_i:                                  # a list that is too short is refused, not truncated
            assert L.drx_dmf_work_order(ou.ctypes.data, n_u, oi.ctypes.data, n_i, seg, out.ctypes.data, cap - 1, zseg.ctypes.data, C.byref(n_part)) < 0
