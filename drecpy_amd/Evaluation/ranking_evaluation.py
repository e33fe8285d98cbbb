"""HR@k / NDCG@k / Precision@k / Recall@k evaluation with the protocol of
DRecPy/Evaluation/Processes/ranking_evaluation.py:19-246 — same arguments, same per-user `random.Random(seed + i)`
streams (candidate lists are bit-identical), same metric plumbing and rounding — but the ranking itself is BATCHED on the
GPU for engine models: one forward over all evaluated users, per-user candidate bitmasks, `drx_topk`, instead of one
`model.rank()` call per user from a 4-thread pool (4.8 users/s in the reference's README:138).
Any other model object (only `rank()` required) goes through the per-user path.
"""
import logging
import random

import numpy as np

from .metrics import HitRatio, NDCG, Precision, RankingMetricABC, Recall


def _user_lists(model, user, ds_test, thr, n_pos, n_neg, train_evaluation, generate_negative_pairs, rng):
    """Candidate construction for one user (ranking_evaluation.py:163-219); returns None when the user is skipped."""
    user_ds = ds_test.select(f'user == {user}')
    pos_ds = user_ds.select(f'interaction >= {thr}')
    if n_pos is None:
        interacted = pos_ds.values_list(['item', 'interaction'])
    else:
        if len(pos_ds) < n_pos:
            return None
        interacted = rng.sample(pos_ds.values_list(['item', 'interaction']), n_pos)
    best_item = None if len(interacted) == 0 else max(interacted, key=lambda p: -p['interaction'])['item']
    interacted = [p['item'] for p in interacted]

    neg_ds = user_ds.select(f'interaction < {thr}')
    if n_neg is None:
        non_interacted = neg_ds.values_list(['item'], to_list=True)
    else:
        if isinstance(n_neg, float):
            n_neg = int(n_neg * len(interacted))
        non_interacted = rng.sample(neg_ds.values_list(['item'], to_list=True), min(n_neg, len(neg_ds)))
        if len(non_interacted) < n_neg and generate_negative_pairs:
            if train_evaluation:
                train_pos = pos_ds
            else:
                train_pos = model.interaction_dataset.select(f'user == {user}, interaction >= {thr}')
            blacklist = set(train_pos.unique('item').values_list('item', to_list=True))
            if not train_evaluation:
                blacklist = blacklist.union(set(pos_ds.unique('item').values_list('item', to_list=True)))
            if model.n_items - len(blacklist) < n_neg:
                logging.warning(f'Skipping user {user} due to not having enough negative eligible items to be sampled.')
                return None
            while len(non_interacted) < n_neg:
                new_item = rng.randint(0, model.n_items - 1)      # internal-range integer used as a raw id (quirk kept)
                if new_item not in blacklist and new_item not in non_interacted:
                    non_interacted.append(new_item)
    all_items = interacted + non_interacted
    if len(all_items) == 0:
        return None
    rng.shuffle(all_items)
    relevancies = {item: (user_ds.select_one(f'item == {item}', ['interaction'], to_list=True) or 0) for item in all_items}
    return {'user': user, 'items': all_items, 'relevant': interacted, 'best': best_item, 'relevancies': relevancies}


def _batched_rank(model, tasks, novelty):
    """All users at once on the device: forward [R,N], candidate masks, top-k with heapq.nlargest order."""
    import torch
    from ..engine import pack_mask_bits
    ds = model.interaction_dataset
    eng = model._engine
    R, N = len(tasks), model.n_items
    uids = np.array([ds.user_to_uid(t['user']) for t in tasks], dtype=np.int32)
    cand = np.zeros((R, N), dtype=bool)
    kmax = 1
    for r, t in enumerate(tasks):
        iids = [ds.item_to_iid(i) for i in t['items']]
        iids = [i for i in iids if i is not None]                   # skip_invalid_items=True
        cand[r, iids] = True
        if novelty:
            cand[r, model._all_user_items(int(uids[r]))] = False
        kmax = max(kmax, int(cand[r].sum()))
    out = []
    with model._device_lock:
        _, pred = eng.forward(uids)
        mask = torch.as_tensor(pack_mask_bits(cand).view(np.int32)).to(eng.device)
        idx, _ = eng.topk(pred, kmax, mask)
        idx = idx.cpu().numpy()
    for r in range(R):
        out.append([ds.iid_to_item(int(i)) for i in idx[r] if i >= 0])
    return out


def ranking_evaluation(model, ds_test=None, n_test_users=None, k=10, n_pos_interactions=None, n_neg_interactions=None,
                       generate_negative_pairs=False, novelty=False, seed=0, max_concurrent_threads=4, **kwds):
    assert n_test_users is None or n_test_users > 0, f'The number of test users ({n_test_users}) should be > 0.'
    assert n_pos_interactions is None or n_pos_interactions > 0, \
        f'The number of positive interactions ({n_pos_interactions}) should be None or an integer > 0.'
    assert n_neg_interactions is None or n_neg_interactions > 0, \
        f'The number of negative interactions ({n_neg_interactions}) should be None or an integer > 0.'
    if generate_negative_pairs and n_neg_interactions is None:
        raise Exception('Cannot generate negative interaction pairs when the number of negative interactions per user '
                        'is not defined. Either set generate_negative_pairs=False or define the n_neg_interactions '
                        'parameter.')
    thr = kwds.get('interaction_threshold', model.interaction_threshold)
    if type(k) is not list:
        k = [k]
    for k_ in k:
        assert k_ > 0, f'k ({k_}) should be > 0.'
    train_evaluation = False
    if ds_test is None or ds_test is model.interaction_dataset:
        train_evaluation = True
        ds_test = model.interaction_dataset
    metrics = kwds.get('metrics', [Precision(), Recall(), HitRatio(), NDCG()])
    assert isinstance(metrics, list), f'Expected "metrics" argument to be a list and found {type(metrics)}.'
    for m in metrics:
        assert isinstance(m, RankingMetricABC), f'Expected metric {m} to be an instance of type RankingMetricABC.'
    metric_sums = {(m.name, k_): [0, 0] for m in metrics for k_ in k}

    users = ds_test.unique('user').values_list('user', to_list=True)
    n_test_users = len(users) if n_test_users is None else min(n_test_users, len(users))
    tasks = []
    for i, user in enumerate(users[:n_test_users]):
        user = user.item() if hasattr(user, 'item') else user     # numeric raw user ids, as in the reference's protocol
        try:
            t = _user_lists(model, user, ds_test, thr, n_pos_interactions, n_neg_interactions, train_evaluation,
                            generate_negative_pairs, random.Random(seed + i))
        except Exception as e:                        # the reference logs and skips (ranking_evaluation.py:152-156)
            logging.error(e)
            t = None
        if t is not None:
            tasks.append(t)

    use_batched = hasattr(model, '_engine') and hasattr(model, '_all_user_items') and kwds.get('batched', True)
    if use_batched and tasks:
        ranked = _batched_rank(model, tasks, novelty)
    else:
        ranked = [[item for _, item in model.rank(t['user'], t['items'], novelty=novelty, skip_invalid_items=True)]
                  for t in tasks]

    for t, recommendations in zip(tasks, ranked):
        for m in metrics:
            names = m.__call__.__code__.co_varnames
            for k_ in k:
                params = {}
                for pn in names:
                    if pn == 'recommendations': params[pn] = recommendations
                    elif pn == 'relevant_recommendations': params[pn] = t['relevant']
                    elif pn == 'relevant_recommendation': params[pn] = t['best']
                    elif pn == 'relevancies': params[pn] = t['relevancies']
                    elif pn == 'k': params[pn] = k_
                try:
                    metric_sums[(m.name, k_)][0] += m(**params)
                    metric_sums[(m.name, k_)][1] += 1
                except Exception:                     # e.g. empty recommendation list (ranking_evaluation.py:243-246)
                    pass
    return {m + f'@{k_}': round(metric_sums[(m, k_)][0] / metric_sums[(m, k_)][1], 4) if metric_sums[(m, k_)][1] > 0 else 0
            for m, k_ in metric_sums}
