"""Top-N recommendation evaluation with the protocol of DRecPy/Evaluation/Processes/recommendation_evaluation.py:18-192
(used by examples/caser.py:17-18): per test user, `model.recommend(user, n=max(k), novelty)` against the user's sampled
positives; same per-user `random.Random(seed + i)` streams, metric plumbing and rounding."""
import logging
import random

from .metrics import HitRatio, NDCG, Precision, RankingMetricABC, Recall


def recommendation_evaluation(model, ds_test=None, n_test_users=None, k=10, n_pos_interactions=None, novelty=False,
                              ignore_low_predictions_threshold=None, seed=0, max_concurrent_threads=4, **kwds):
    assert n_test_users is None or n_test_users > 0, f'The number of test users ({n_test_users}) should be > 0.'
    assert n_pos_interactions is None or n_pos_interactions > 0, \
        f'The number of positive interactions ({n_pos_interactions}) should be None or an integer > 0.'
    thr = kwds.get('interaction_threshold', model.interaction_threshold)
    if type(k) is not list:
        k = [k]
    for k_ in k:
        assert k_ > 0, f'k ({k_}) should be > 0.'
    if ds_test is None:
        ds_test = model.interaction_dataset
    metrics = kwds.get('metrics', [Precision(), Recall(), HitRatio(), NDCG()])
    assert isinstance(metrics, list), f'Expected "metrics" argument to be a list and found {type(metrics)}.'
    for m in metrics:
        assert isinstance(m, RankingMetricABC), f'Expected metric {m} to be an instance of type RankingMetricABC.'
    metric_sums = {(m.name, k_): [0, 0] for m in metrics for k_ in k}
    users = ds_test.unique('user').values_list('user', to_list=True)
    n_test_users = len(users) if n_test_users is None else min(n_test_users, len(users))
    for i, user in enumerate(users[:n_test_users]):
        user = user.item() if hasattr(user, 'item') else user
        rng = random.Random(seed + i)
        try:
            user_ds = ds_test.select(f'user == {user}')
            pos_ds = user_ds.select(f'interaction >= {thr}')
            if n_pos_interactions is None:
                interacted = pos_ds.values_list(['item', 'interaction'])
            else:
                if len(pos_ds) < n_pos_interactions:
                    continue
                interacted = rng.sample(pos_ds.values_list(['item', 'interaction']), n_pos_interactions)
            best_item = None if len(interacted) == 0 else max(interacted, key=lambda p: -p['interaction'])['item']
            interacted = [p['item'] for p in interacted]
            if len(interacted) == 0:
                continue
            recommendations = [item for _, item in model.recommend(user, n=max(k), novelty=novelty, skip_invalid_items=True,
                                                                   interaction_threshold=ignore_low_predictions_threshold)]
            relevancies = {item: (user_ds.select_one(f'item == {item}', ['interaction'], to_list=True) or 0)
                           for item in set(interacted).union(set(recommendations))}
        except Exception as e:
            logging.error(e)
            continue
        for m in metrics:
            names = m.__call__.__code__.co_varnames
            for k_ in k:
                params = {}
                for pn in names:
                    if pn == 'recommendations': params[pn] = recommendations
                    elif pn == 'relevant_recommendations': params[pn] = interacted
                    elif pn == 'relevant_recommendation': params[pn] = best_item
                    elif pn == 'relevancies': params[pn] = relevancies
                    elif pn == 'k': params[pn] = k_
                try:
                    metric_sums[(m.name, k_)][0] += m(**params)
                    metric_sums[(m.name, k_)][1] += 1
                except Exception:
                    pass
    return {m + f'@{k_}': round(metric_sums[(m, k_)][0] / metric_sums[(m, k_)][1], 4) if metric_sums[(m, k_)][1] > 0 else 0
            for m, k_ in metric_sums}
