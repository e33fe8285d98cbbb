"""Pieces shared by the two evaluation processes: per-user positive sampling and the metric accumulation that mirrors how
the reference evaluators call metric objects (by inspecting `__call__`'s parameter names)."""
from .metrics import HitRatio, NDCG, Precision, RankingMetricABC, Recall

_SOURCES = ('recommendations', 'relevant_recommendations', 'relevant_recommendation', 'relevancies')


def resolve_metrics(kwds):
    metrics = kwds.get('metrics', [Precision(), Recall(), HitRatio(), NDCG()])
    assert isinstance(metrics, list), \
        f'Expected "metrics" argument to be a list and found {type(metrics)}. Should contain instances of RankingMetricABC.'
    for m in metrics:
        assert isinstance(m, RankingMetricABC), f'Expected metric {m} to be an instance of type RankingMetricABC.'
    return metrics


def as_k_list(k):
    ks = k if type(k) is list else [k]
    for k_ in ks:
        assert k_ > 0, f'k ({k_}) should be > 0.'
    return ks


def sample_positives(user_ds, threshold, n_pos, rng):
    """(relevant item ids, 'best' item, positives dataset) or None when the user has too few positives.
    The 'best' item is the one with the SMALLEST interaction (max over -interaction) — reference behaviour."""
    pos_ds = user_ds.select(f'interaction >= {threshold}')
    pairs = pos_ds.values_list(['item', 'interaction'])
    if n_pos is not None:
        if len(pos_ds) < n_pos:
            return None
        pairs = rng.sample(pairs, n_pos)
    best = max(pairs, key=lambda p: -p['interaction'])['item'] if pairs else None
    return [p['item'] for p in pairs], best, pos_ds


class MetricTable:
    """Running sums of metric@k values; a metric call that raises (e.g. an empty ranking) is skipped."""

    def __init__(self, metrics, ks):
        self.metrics, self.ks = metrics, ks
        self.sums = {(m.name, k): [0, 0] for m in metrics for k in ks}

    def add(self, recommendations, relevant, best, relevancies):
        values = dict(zip(_SOURCES, (recommendations, relevant, best, relevancies)))
        for m in self.metrics:
            declared = getattr(m, 'inputs', None)
            names = declared if declared else m.__call__.__code__.co_varnames      # reference-style explicit signature
            wanted = [n for n in names if n in values and n != 'recommendations']
            for k in self.ks:
                try:
                    score = m(recommendations, k=k, **{n: values[n] for n in wanted})
                except Exception:
                    continue
                cell = self.sums[(m.name, k)]
                cell[0] += score
                cell[1] += 1

    def result(self):
        return {f'{name}@{k}': (round(tot / cnt, 4) if cnt > 0 else 0) for (name, k), (tot, cnt) in self.sums.items()}
