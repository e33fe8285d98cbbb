"""Host-side dataset logic: query language, unique/first-appearance order, drop/select bookkeeping, CSR vectors,
MovieLens wire formats (tab, '::', csv)."""
import os

import numpy as np
import pytest

from helpers import load_frames, load_json


def _ds(frame):
    from drecpy_amd.Dataset import InteractionDataset
    return InteractionDataset.read_df({k: v.copy() for k, v in frame.items()}, verbose=False)


def test_query_language_and_bookkeeping():
    f = load_frames()['ls_int_ts']
    ds = _ds(f)
    n = len(ds)
    assert n == len(f['user']) and ds.max('interaction') == f['interaction'].max()
    sub = ds.select('interaction >= 3, timestamp < 500000')
    want = (f['interaction'] >= 3) & (f['timestamp'] < 500000)
    assert len(sub) == int(want.sum()) and len(ds) == n                      # select copies by default
    assert sub.values_list('rid', to_list=True) == np.flatnonzero(want).tolist()
    one = ds.select_one('user == 5', ['item', 'interaction'])
    first = np.flatnonzero(f['user'] == 5)[0]
    assert one == {'item': f['item'][first], 'interaction': f['interaction'][first]}
    assert ds.select_one('user == -1') is None and not ds.exists('user == -1') and ds.exists('user == 5')
    with pytest.raises(Exception):
        ds.select('interaction>=3')                                           # missing spaces
    with pytest.raises(AssertionError):
        ds.select('nope == 3')
    u = ds.unique('user')
    _, firsts = np.unique(f['user'], return_index=True)
    assert u.values_list('user', to_list=True) == f['user'][np.sort(firsts)].tolist()     # first-appearance order
    assert ds.count_unique(['user', 'item']) == len(set(zip(f['user'].tolist(), f['item'].tolist())))
    kept = ds.drop([0, 1, 2], keep=True)
    assert kept.values_list('rid', to_list=True) == [0, 1, 2] and len(ds.drop([0, 1, 2])) == n - 3
    ds.assign_internal_ids()
    uid = ds.user_to_uid(f['user'][10])
    assert ds.uid_to_user(uid) == f['user'][10] and ds.user_to_uid(10 ** 9) is None and ds.iid_to_item(10 ** 9) is None
    assert len(ds.select(f'uid == {uid}')) == int((f['user'] == f['user'][10]).sum())
    ds.apply('interaction', lambda x: 1 if x > 2 else 0)
    assert set(np.unique(ds._cols['interaction']).tolist()) <= {0, 1}
    with pytest.raises(Exception):
        ds.apply('user', lambda x: x)


def test_string_ids_and_interaction_vectors():
    iv = load_json('interaction_vecs.json')['pt_str_gapped']
    ds = _ds(load_frames()['pt_str_gapped'])
    ds.assign_internal_ids()
    assert ds.user_to_uid('user_0000') is not None and ds.user_to_uid('nobody') is None
    for uid, e in iv['user'].items():
        v = ds.select_user_interaction_vec(int(uid))
        want = np.zeros(e['n']); want[e['idx']] += np.array(e['val'])
        np.testing.assert_array_equal(np.asarray(v.todense()).ravel(), want)
    ip, idx = ds.positives_csr(3)
    dense = np.zeros((len(ip) - 1, iv['user']['0']['n']))
    np.add.at(dense, (ds._cols['uid'].astype(int), ds._cols['iid'].astype(int)), ds._cols['interaction'])
    for u in (0, 5, 17):
        assert idx[ip[u]:ip[u + 1]].tolist() == np.flatnonzero(dense[u] >= 3).tolist()


def test_movielens_wire_formats(tmp_path):
    from drecpy_amd.Dataset import load_movielens
    rows = [(1, 10, 5, 100), (2, 10, 3, 101), (1, 11, 4, 102)]
    d100 = tmp_path / 'ml-100k'; d100.mkdir()
    (d100 / 'u.data').write_text(''.join(f'{u}\t{i}\t{r}\t{t}\n' for u, i, r, t in rows))
    d1m = tmp_path / 'ml-1m'; d1m.mkdir()
    (d1m / 'ratings.dat').write_text(''.join(f'{u}::{i}::{r}::{t}\n' for u, i, r, t in rows))
    d20 = tmp_path / 'ml-20m'; d20.mkdir()
    (d20 / 'ratings.csv').write_text('userId,movieId,rating,timestamp\n' + ''.join(f'{u},{i},{r}.5,{t}\n' for u, i, r, t in rows))
    for name, folder, bump in (('ml-100k', d100, 0), ('ml-1m', d1m, 0), ('ml-20m', d20, 0.5)):
        ds = load_movielens(name, str(folder))
        assert ds.values_list(['user', 'item', 'interaction', 'timestamp'], to_list=True) == \
            [[u, i, r + bump, t] for u, i, r, t in rows]
    with pytest.raises(FileNotFoundError):
        load_movielens('ml-10m', str(tmp_path))
