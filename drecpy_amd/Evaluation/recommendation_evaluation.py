"""Top-N recommendation evaluation with the protocol of DRecPy/Evaluation/Processes/recommendation_evaluation.py (used by
examples/caser.py:17-18): per test user `model.recommend(user, n=max(k), novelty)` against the user's (sampled) test
positives; user i draws from `random.Random(seed + i)`; result keys `metric@k`, rounded to 4 decimals."""
import logging
import random

from ._protocol import MetricTable, as_k_list, resolve_metrics, sample_positives


def recommendation_evaluation(model, ds_test=None, n_test_users=None, k=10, n_pos_interactions=None, novelty=False,
                              ignore_low_predictions_threshold=None, seed=0, max_concurrent_threads=4, **kwds):
    assert n_test_users is None or n_test_users > 0, f'The number of test users ({n_test_users}) should be > 0.'
    assert n_pos_interactions is None or n_pos_interactions > 0, \
        f'The number of positive interactions ({n_pos_interactions}) should be None or an integer > 0.'
    threshold = kwds.get('interaction_threshold', model.interaction_threshold)
    ks = as_k_list(k)
    table = MetricTable(resolve_metrics(kwds), ks)
    ds_test = model.interaction_dataset if ds_test is None else ds_test
    users = ds_test.unique('user').values_list('user', to_list=True)
    if n_test_users is not None:
        users = users[:n_test_users]
    for offset, user in enumerate(users):
        user = user.item() if hasattr(user, 'item') else user
        try:
            user_ds = ds_test.select(f'user == {user}')
            drawn = sample_positives(user_ds, threshold, n_pos_interactions, random.Random(seed + offset))
            if drawn is None or not drawn[0]:
                continue
            relevant, best, _ = drawn
            ranked = model.recommend(user, n=max(ks), novelty=novelty, skip_invalid_items=True,
                                     interaction_threshold=ignore_low_predictions_threshold)
            recommendations = [item for _, item in ranked]
            relevancies = {item: (user_ds.select_one(f'item == {item}', ['interaction'], to_list=True) or 0)
                           for item in set(relevant) | set(recommendations)}
        except Exception as err:          # the reference logs and skips the user
            logging.error(err)
            continue
        table.add(recommendations, relevant, best, relevancies)
    return table.result()
