"""Ranking metrics (HR, NDCG, precision, recall, ...) usable exactly like the reference's
(DRecPy/Evaluation/Metrics/ranking.py): `Metric()(recommendations, k=..., relevant_recommendations=... | relevancies=...
| relevant_recommendation=...)`.  Host arithmetic on short lists.

Each metric declares the extra keyword inputs it needs in `inputs`; the shared `__call__` returns 0 when one of them is
missing, truncates the ranking to its first k entries and hands over to `_score`.  User-defined metrics may instead
override `__call__` with explicit parameter names like the reference's classes do — the evaluators support both.
"""
import math
from abc import ABC, abstractmethod


class RankingMetricABC(ABC):
    inputs = ()

    def __init__(self):
        self.name = type(self).__name__          # labels the `name@k` keys of evaluation results

    def __call__(self, recommendations, k=None, **given):
        extra = [given.get(name) for name in self.inputs]
        if any(value is None for value in extra):
            return 0
        top = recommendations if k is None else recommendations[:k]
        return self._score(top, k, *extra)

    @abstractmethod
    def _score(self, top, k, *extra):
        raise NotImplementedError


def _dcg(ranking, relevancies, strong):
    total = 0.0
    for position, item in enumerate(ranking):
        rel = float(relevancies[item])
        total += ((2.0 ** rel - 1.0) if strong else rel) / math.log2(position + 2)
    return total


class DCG(RankingMetricABC):
    """Discounted cumulative gain; `strong_relevancy` uses 2^rel - 1 as the gain."""
    inputs = ('relevancies',)

    def __init__(self, strong_relevancy=True):
        super().__init__()
        self.strong_relevancy = strong_relevancy

    def _score(self, top, k, relevancies):
        return _dcg(top, relevancies, self.strong_relevancy)


class NDCG(DCG):
    """DCG divided by the DCG of the ideal ordering of every id present in `relevancies`."""

    def _score(self, top, k, relevancies):
        ideal = sorted(relevancies, key=lambda item: -relevancies[item])
        ideal = ideal if k is None else ideal[:k]
        return _dcg(top, relevancies, self.strong_relevancy) / _dcg(ideal, relevancies, self.strong_relevancy)


class _SetMetric(RankingMetricABC):
    inputs = ('relevant_recommendations',)


class HitRatio(_SetMetric):
    """Share of the relevant items found in the top k; ids are compared as strings."""

    def _score(self, top, k, relevant):
        wanted = {str(item) for item in relevant}
        return len(wanted.intersection(str(item) for item in top)) / len(wanted)


class Recall(_SetMetric):
    def _score(self, top, k, relevant):
        return len(set(top).intersection(relevant)) / len(relevant)


class Precision(_SetMetric):
    def _score(self, top, k, relevant):
        return len(set(top).intersection(relevant)) / len(top)


class FScore(_SetMetric):
    """Weighted harmonic mean of precision and recall (beta > 1 favours recall)."""

    def __init__(self, beta=1):
        super().__init__()
        self.beta = beta

    def _score(self, top, k, relevant):
        hits = len(set(top).intersection(relevant))
        p, r, b2 = hits / len(top), hits / len(relevant), self.beta ** 2
        return (1 + b2) * p * r / (b2 * p + r)


class AveragePrecision(_SetMetric):
    """Sum of precision@i over the positions i where a relevant item appears for the first time, over min(|relevant|, k)."""

    def _score(self, top, k, relevant):
        seen, hits_so_far, total = set(), set(), 0.0
        for position, item in enumerate(top, start=1):
            if item in relevant:
                hits_so_far.add(item)
                if item not in seen:
                    total += len(hits_so_far) / position
            seen.add(item)
        return total / (len(relevant) if k is None else min(len(relevant), k))


class ReciprocalRank(RankingMetricABC):
    """1 / (1-based position of the single most relevant item), 0 when it is not in the top k."""
    inputs = ('relevant_recommendation',)

    def _score(self, top, k, best):
        return 1 / (top.index(best) + 1) if best in top else 0
