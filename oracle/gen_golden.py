"""Generates tests/golden/*.json|npz by RUNNING THE REFERENCE (dev container only).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Imports the reference's Dataset / Sampler /
Evaluation code through the recipe of oracle/ref_import.py (TensorFlow stubbed — none of the code
exercised here touches TensorFlow) and records inputs + the reference's outputs.  The fixtures are
data only; no reference source is copied.  Run:  python oracle/gen_golden.py

What is pinned:
  idmap.json ............ assign_internal_ids            (mem_dataset.py:309-330)
  point_sampler.json .... PointSampler.sample streams    (point_sampler.py:44-61), plus the reference's own
                          known answers null_interaction_pair_generator(seed=23) -> (1,0),(0,2),(1,3),(0,1)
                          (tests/Dataset/test_mem_dataset.py:678-697)
  list_sampler.json ..... ListSampler.sample_group_records as Caser configures it (caser.py:72-75)
  interaction_vecs.json . select_user_interaction_vec / select_item_interaction_vec rows
  frames.npz ............ the seeded synthetic input frames used above
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
RES = '/root/reference/tests/Dataset/resources'


def _py(x):
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, (list, tuple)):
        return [_py(v) for v in x]
    return x


def synth_frame(seed, n_users, n_items, n_rows, string_users, gapped_items, with_ts, dup_frac=0.0):
    r = np.random.RandomState(seed)
    u = r.randint(0, n_users, size=n_rows)
    pop = 1.0 / np.arange(1, n_items + 1)
    pop /= pop.sum()
    i = r.choice(n_items, size=n_rows, p=pop)
    if dup_frac == 0.0:                      # unique (u,i) pairs, like MovieLens
        _, first = np.unique(u.astype(np.int64) * n_items + i, return_index=True)
        first.sort()
        u, i = u[first], i[first]
    val = r.randint(0, 6, size=len(u)).astype(np.int64)      # 0..5, zeros exercise the threshold
    users = np.array([f'user_{x:04d}' for x in u]) if string_users else (u + 1).astype(np.int64)
    items = (i * 3 + 7).astype(np.int64) if gapped_items else (i + 1).astype(np.int64)
    f = {'user': users, 'item': items, 'interaction': val}
    if with_ts:
        f['timestamp'] = r.randint(0, 10 ** 6, size=len(u)).astype(np.int64)
    return f


def main():
    warnings.filterwarnings('ignore')
    from ref_import import import_reference
    import_reference()
    import pandas as pd
    from DRecPy.Dataset import InteractionDataset
    from DRecPy.Sampler import PointSampler, ListSampler

    os.makedirs(OUT, exist_ok=True)
    frames = {
        'pt_str_gapped': synth_frame(1, 300, 500, 9000, True, True, False),
        'pt_int_dense': synth_frame(2, 64, 40, 1500, False, False, False),     # dense: many rejections
        'ls_int_ts': synth_frame(3, 120, 400, 7000, False, True, True),
    }
    np.savez_compressed(os.path.join(OUT, 'frames.npz'),
                        **{f'{k}__{c}': v for k, f in frames.items() for c, v in f.items()})

    def ds_of(frame):
        ds = InteractionDataset.read_df(pd.DataFrame(frame), verbose=False)
        ds.assign_internal_ids()
        return ds

    # ---- id maps --------------------------------------------------------------------------
    idmap = {}
    for name in ('test.csv', 'test_floats.csv', 'test_int_ids.csv'):
        ds = InteractionDataset(os.path.join(RES, name), columns=['user', 'item', 'interaction'],
                                has_header=True, verbose=False)
        ds.assign_internal_ids()
        rows = ds.values_list(['user', 'item', 'interaction', 'uid', 'iid'], to_list=True)
        idmap[name] = {'user': [_py(r[0]) for r in rows], 'item': [_py(r[1]) for r in rows],
                       'interaction': [_py(r[2]) for r in rows],
                       'uid': [_py(r[3]) for r in rows], 'iid': [_py(r[4]) for r in rows],
                       'user_to_uid': {str(k): _py(v) for k, v in ds._user_mapping.items()},
                       'item_to_iid': {str(k): _py(v) for k, v in ds._item_mapping.items()}}
    for k, f in frames.items():
        ds = ds_of(f)
        idmap[k] = {'uid': [int(x) for x in ds._df['uid'].values],
                    'iid': [int(x) for x in ds._df['iid'].values],
                    'n_users': int(ds.count_unique('uid')), 'n_items': int(ds.count_unique('iid')),
                    'probe_user_to_uid': [[_py(u), _py(ds.user_to_uid(u))] for u in f['user'][:20].tolist()]
                    + [['no_such_user' if f['user'].dtype.kind == 'U' else -5,
                        _py(ds.user_to_uid('no_such_user' if f['user'].dtype.kind == 'U' else -5))]],
                    'probe_iid_to_item': [[i, _py(ds.iid_to_item(i))] for i in range(10)]}
    json.dump(idmap, open(os.path.join(OUT, 'idmap.json'), 'w'))

    # ---- PointSampler streams ----------------------------------------------------------------
    ps = {}
    ds = InteractionDataset(os.path.join(RES, 'test.csv'), columns=['user', 'item', 'interaction'],
                            has_header=True, verbose=False)
    ds.assign_internal_ids()
    ps['test.csv|5|0.001|10'] = [_py(list(t)) for t in PointSampler(ds, 5, 0.001, 10).sample(64)]
    g = ds.null_interaction_pair_generator(seed=23)
    ps['test.csv|null_pair_gen|seed23'] = [_py(list(next(g))) for _ in range(16)]
    g = ds.select_random_generator(seed=23)
    ps['test.csv|select_random_gen|seed23'] = [[_py(r['uid']), _py(r['iid']), _py(r['interaction'])]
                                               for r in (next(g) for _ in range(16))]
    for k, cfgs in (('pt_str_gapped', [(5, 0.001, 10, 600), (1, 3, 7, 300), (5, None, 123, 300)]),
                    ('pt_int_dense', [(5, 0.001, 10, 400), (2, 4, 0, 300)])):
        ds = ds_of(frames[k])
        for neg, thr, seed, n in cfgs:
            s = PointSampler(ds, neg, thr, seed)
            a = s.sample(n // 2)
            a += s.sample(n - n // 2)            # two calls: generators keep their state
            ps[f'{k}|{neg}|{thr}|{seed}'] = [_py(list(t)) for t in a]
    json.dump(ps, open(os.path.join(OUT, 'point_sampler.json'), 'w'))

    # ---- ListSampler streams (Caser configuration, caser.py:72-75) --------------------------------
    ls = {}
    ds = ds_of(frames['ls_int_ts'])
    for (L, T, neg, thr, seed, n) in ((5, 3, 2, 0.001, 10, 150), (3, 2, 3, 2, 5, 100)):
        s = ListSampler(ds, ['uid'], neg_ratio=neg, n_targets=T, interaction_threshold=thr,
                        negative_ids_col='iid', min_positive_records=L, max_positive_records=L,
                        sort_column='timestamp', seed=seed)
        recs = s.sample_group_records(n)
        ls[f'{L}|{T}|{neg}|{thr}|{seed}'] = [
            {'uid': _py(b[0]['uid']), 'before': [_py(r['iid']) for r in b], 'after': [_py(r['iid']) for r in a],
             'before_rid': [_py(r['rid']) for r in b], 'after_rid': [_py(r['rid']) for r in a],
             'neg': [_py(x) for x in ng]}
            for b, a, ng in recs]
    json.dump(ls, open(os.path.join(OUT, 'list_sampler.json'), 'w'))

    # ---- interaction vectors -------------------------------------------------------------------
    iv = {}
    for k in ('pt_str_gapped', 'pt_int_dense'):
        ds = ds_of(frames[k])
        iv[k] = {'user': {}, 'item': {}}
        for uid in (0, 1, 5, 17, 63):
            v = ds.select_user_interaction_vec(uid)
            iv[k]['user'][str(uid)] = {'idx': [int(x) for x in v.indices.tolist()],
                                       'val': [float(x) for x in v.data.tolist()], 'n': int(v.shape[1])}
        for iid in (0, 2, 11, 39):
            v = ds.select_item_interaction_vec(iid).tocsr()
            v.sort_indices()
            iv[k]['item'][str(iid)] = {'idx': [int(x) for x in v.indices.tolist()],
                                       'val': [float(x) for x in v.data.tolist()], 'n': int(v.shape[1])}
    json.dump(iv, open(os.path.join(OUT, 'interaction_vecs.json'), 'w'))
    # ---- ranking_evaluation protocol + metrics ---------------------------------------------
    import random as _random
    from heapq import nlargest
    from DRecPy.Evaluation.Processes import ranking_evaluation
    from DRecPy.Evaluation.Metrics import HitRatio, NDCG, Precision, Recall, ReciprocalRank, AveragePrecision, FScore, DCG

    r0 = _random.Random(0)       # same construction as the reference's own fixture (test_ranking_evaluation.py:12-19)
    rows = [[u, i, r0.randint(-1, 5)] for u in range(50) for i in range(200) if r0.randint(0, 4) == 0]
    rows = np.array(rows, dtype=np.int64)
    # deterministic hold-out: the last 5 rows of every user go to the test frame
    test_mask = np.zeros(len(rows), bool)
    for u in range(50):
        idx = np.flatnonzero(rows[:, 0] == u)
        test_mask[idx[-5:]] = True
    tr, te = rows[~test_mask], rows[test_mask]
    scores = np.random.RandomState(4).rand(50, 200)

    class FakeModel:
        # ranks by a fixed score matrix; only the attributes ranking_evaluation reads
        def __init__(self, ds):
            self.interaction_dataset = ds
            self.n_items = ds.count_unique('iid')
            self.interaction_threshold = 0.001
            self.fitted = True

        def rank(self, user_id, item_ids, novelty=True, skip_invalid_items=True, **kw):
            ds = self.interaction_dataset
            uid = ds.user_to_uid(user_id)
            iids = [ds.item_to_iid(i) for i in item_ids]
            iids = [i for i in iids if i is not None]
            if novelty:
                seen = set(ds.select(f'uid == {uid}').values_list('iid', to_list=True))
                iids = [i for i in iids if i not in seen]
            raw_u = int(ds.uid_to_user(uid))
            lst = nlargest(len(iids), [(scores[raw_u, int(ds.iid_to_item(i))], i) for i in set(iids)])
            return [(s, ds.iid_to_item(i)) for s, i in lst]

    ds_tr = InteractionDataset.read_df(pd.DataFrame(tr, columns=['user', 'item', 'interaction']), verbose=False)
    ds_te = InteractionDataset.read_df(pd.DataFrame(te, columns=['user', 'item', 'interaction']), verbose=False)
    ds_tr.assign_internal_ids()
    fm = FakeModel(ds_tr)
    cfgs = [dict(k=2), dict(k=2, n_neg_interactions=20, generate_negative_pairs=True), dict(k=2, n_neg_interactions=1),
            dict(k=[1, 5, 10]), dict(k=2, novelty=True), dict(k=2, n_pos_interactions=1),
            dict(k=[1, 5, 10], novelty=True, n_test_users=30, n_pos_interactions=1, n_neg_interactions=100,
                 generate_negative_pairs=True, seed=10),                                  # examples/cdae.py:15-17 protocol
            dict(k=3, n_neg_interactions=0.5, seed=3)]
    re_out = []
    for c in cfgs:
        res = ranking_evaluation(fm, ds_te, verbose=False, **c)
        re_out.append({'cfg': c, 'result': res})
    res_train = ranking_evaluation(fm, None, k=[1, 3], n_pos_interactions=2, n_neg_interactions=10,
                                   generate_negative_pairs=True, seed=5, verbose=False)
    mrng = np.random.RandomState(9)
    mcases = []
    for _ in range(40):
        n = int(mrng.randint(1, 12))
        recs = [int(x) for x in mrng.permutation(30)[:n]]
        rel = [int(x) for x in mrng.permutation(30)[:int(mrng.randint(1, 8))]]
        relv = {int(i): float(mrng.randint(0, 6)) for i in set(recs) | set(rel)}
        if all(v == 0 for v in relv.values()):
            relv[recs[0]] = 3.0
        kk = [None, 1, 3, 10][int(mrng.randint(0, 4))]
        vals = {'HitRatio': HitRatio()(recs, k=kk, relevant_recommendations=rel),
                'Recall': Recall()(recs, k=kk, relevant_recommendations=rel),
                'Precision': Precision()(recs, k=kk, relevant_recommendations=rel),
                'AveragePrecision': AveragePrecision()(recs, k=kk, relevant_recommendations=rel),
                'ReciprocalRank': ReciprocalRank()(recs, k=kk, relevant_recommendation=rel[0]),
                'NDCG': NDCG()(recs, k=kk, relevancies=relv), 'DCG_weak': DCG(strong_relevancy=False)(recs, k=kk, relevancies=relv)}
        p_, r_ = vals['Precision'], vals['Recall']
        vals['FScore'] = FScore()(recs, k=kk, relevant_recommendations=rel) if (p_ + r_) > 0 else None
        mcases.append({'recs': recs, 'rel': rel, 'relv': {str(a): b for a, b in relv.items()}, 'k': kk, 'vals': vals})
    # recommendation_evaluation (examples/caser.py:17-18 protocol) with a recommend()-capable fake model
    from DRecPy.Evaluation.Processes import recommendation_evaluation

    class FakeRec(FakeModel):
        def recommend(self, user_id, n=None, novelty=True, interaction_threshold=None, **kw):
            ds = self.interaction_dataset
            items = [ds.iid_to_item(i) for i in range(self.n_items)]
            ranked = self.rank(user_id, items, novelty=novelty)
            if interaction_threshold is not None:
                ranked = [x for x in ranked if x[0] >= interaction_threshold]
            return ranked[:n]
    fr = FakeRec(ds_tr)
    rec_out = []
    for c in (dict(k=[1, 5, 10], novelty=True, seed=10), dict(k=3, n_pos_interactions=2, seed=1),
              dict(k=[2, 4], novelty=False, ignore_low_predictions_threshold=0.5, n_test_users=20)):
        mets = [AveragePrecision(), Precision(), Recall()] if 'novelty' in c and c.get('seed') == 10 else None
        kw = dict(c)
        if mets is not None:
            kw['metrics'] = mets
        rec_out.append({'cfg': c, 'caser_metrics': mets is not None, 'result': recommendation_evaluation(fr, ds_te, verbose=False, **kw)})
    # leave_k_out on a frame with timestamps
    from DRecPy.Evaluation.Splits import leave_k_out
    lko = []
    fr3 = frames['ls_int_ts']
    for c in (dict(k=2, seed=10), dict(k=0.2, last_timestamps=True, seed=0), dict(k=3, min_user_interactions=40, seed=5),
              dict(k=0.1, seed=3)):
        d_tr, d_te = leave_k_out(InteractionDataset.read_df(pd.DataFrame(fr3), verbose=False), verbose=False, **c)
        lko.append({'cfg': c, 'train_rids': sorted(int(x) for x in d_tr.values_list('rid', to_list=True)),
                    'test_rids': sorted(int(x) for x in d_te.values_list('rid', to_list=True))})
    json.dump({'train': tr.tolist(), 'test': te.tolist(), 'scores_seed': 4, 'evals': re_out, 'train_eval': res_train,
               'metric_cases': mcases, 'rec_evals': rec_out, 'leave_k_out': lko},
              open(os.path.join(OUT, 'ranking_eval.json'), 'w'))
    print('golden fixtures written to', OUT)
    for fn in sorted(os.listdir(OUT)):
        print(' ', fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == '__main__':
    main()
