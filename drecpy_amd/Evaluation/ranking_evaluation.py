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

from ._protocol import MetricTable, as_k_list, resolve_metrics, sample_positives


def _user_lists(model, user, ds_test, thr, n_pos, n_neg, train_evaluation, generate_negative_pairs, rng):
    """Candidate construction for one user (ranking_evaluation.py:163-219); returns None when the user is skipped.
    RNG call order: sample positives, sample test negatives, generate extra negatives, shuffle."""
    user_ds = ds_test.select(f'user == {user}')
    drawn = sample_positives(user_ds, thr, n_pos, rng)
    if drawn is None:
        return None
    relevant, best_item, pos_ds = drawn

    neg_items = user_ds.select(f'interaction < {thr}').values_list(['item'], to_list=True)
    if n_neg is not None:
        if isinstance(n_neg, float):
            n_neg = int(n_neg * len(relevant))
        n_from_test = min(n_neg, len(neg_items))
        neg_items = rng.sample(neg_items, n_from_test)
        if generate_negative_pairs and len(neg_items) < n_neg:
            train_pos = pos_ds if train_evaluation else \
                model.interaction_dataset.select(f'user == {user}, interaction >= {thr}')
            blacklist = set(train_pos.unique('item').values_list('item', to_list=True))
            if not train_evaluation:
                blacklist |= set(pos_ds.unique('item').values_list('item', to_list=True))
            if model.n_items - len(blacklist) < n_neg:
                logging.warning(f'Skipping user {user}: only {model.n_items - len(blacklist)} eligible negative items for '
                                f'{n_neg} requested (decrease n_neg_interactions).')
                return None
            while len(neg_items) < n_neg:
                candidate = rng.randint(0, model.n_items - 1)     # internal-range integer used as a RAW id (quirk kept)
                if candidate not in blacklist and candidate not in neg_items:
                    neg_items.append(candidate)
    candidates = relevant + neg_items
    if not candidates:
        return None
    rng.shuffle(candidates)
    relevancies = {item: (user_ds.select_one(f'item == {item}', ['interaction'], to_list=True) or 0) for item in candidates}
    return {'user': user, 'items': candidates, 'relevant': relevant, 'best': best_item, 'relevancies': relevancies}


_RANK_CHUNK_BYTES = 1 << 30        # device bytes of one chunk's [R, N] prediction matrix


def _rank_chunk(model, tasks, novelty):
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
    with model._device_lock:
        _, pred = eng.forward(uids)
        mask = torch.as_tensor(pack_mask_bits(cand).view(np.int32)).to(eng.device)
        idx, _ = eng.topk(pred, kmax, mask)
        idx = idx.cpu().numpy()
    return [[ds.iid_to_item(int(i)) for i in idx[r] if i >= 0] for r in range(R)]


def _batched_rank(model, tasks, novelty):
    """Users in chunks on the device: forward [R,N], candidate masks, top-k with heapq.nlargest order.  A chunk holds as many
    users as keep its prediction matrix under _RANK_CHUNK_BYTES (and R*N under 2^31, the index range of drx_topk's long-row
    path); a chunk that fails falls back to one model.rank() per user, and a user that still fails is logged and skipped like
    the reference does (ranking_evaluation.py:152-156) — its entry is None."""
    N = max(int(model.n_items), 1)
    per = max(1, min(_RANK_CHUNK_BYTES // (4 * N), ((1 << 31) - 1) // N))
    out = []
    for lo in range(0, len(tasks), per):
        chunk = tasks[lo:lo + per]
        try:
            out.extend(_rank_chunk(model, chunk, novelty))
        except Exception as err:
            logging.error(f'batched ranking of users {lo}..{lo + len(chunk) - 1} failed ({err}); ranking them one by one')
            for t in chunk:
                try:
                    out.append([item for _, item in model.rank(t['user'], t['items'], novelty=novelty, skip_invalid_items=True)])
                except Exception as e:
                    logging.error(e)
                    out.append(None)
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
    ks = as_k_list(k)
    train_evaluation = ds_test is None or ds_test is model.interaction_dataset
    if train_evaluation:
        ds_test = model.interaction_dataset
    table = MetricTable(resolve_metrics(kwds), ks)

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
        if recommendations is None:                   # a user whose ranking failed: logged and skipped
            continue
        table.add(recommendations, t['relevant'], t['best'], t['relevancies'])
    return table.result()
