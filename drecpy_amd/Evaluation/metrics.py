"""Ranking metrics with the call signatures of DRecPy/Evaluation/Metrics/ranking.py (host arithmetic on small lists)."""
import math
from abc import ABC, abstractmethod


class RankingMetricABC(ABC):
    def __init__(self):
        self.name = self.__class__.__name__

    @abstractmethod
    def __call__(self, recommendations, k=None):
        pass


class DCG(RankingMetricABC):
    def __init__(self, strong_relevancy=True):
        super().__init__()
        self.strong_relevancy = strong_relevancy

    def __call__(self, recommendations, k=None, relevancies=None):
        if relevancies is None:
            return 0
        if k is not None:
            recommendations = recommendations[:k]
        total = 0
        for i, r in enumerate(recommendations):
            rel = float(relevancies[r])
            total += ((2 ** rel - 1) if self.strong_relevancy else rel) / math.log2(2 + i)
        return total


class NDCG(RankingMetricABC):
    def __init__(self, strong_relevancy=True):
        super().__init__()
        self.strong_relevancy = strong_relevancy
        self.dcg = DCG(strong_relevancy=strong_relevancy)

    def __call__(self, recommendations, k=None, relevancies=None):
        if relevancies is None:
            return 0
        cur = self.dcg(recommendations, relevancies=relevancies, k=k)
        ideal = sorted(relevancies.keys(), key=lambda x: -relevancies[x])
        return cur / self.dcg(ideal, relevancies=relevancies, k=k)


class HitRatio(RankingMetricABC):
    def __call__(self, recommendations, k=None, relevant_recommendations=None):
        if relevant_recommendations is None:
            return 0
        if k is not None:
            recommendations = recommendations[:k]
        rec = set(str(i) for i in recommendations)
        rel = set(str(i) for i in relevant_recommendations)
        return len(rec & rel) / len(rel)


class ReciprocalRank(RankingMetricABC):
    def __call__(self, recommendations, k=None, relevant_recommendation=None):
        if relevant_recommendation is None:
            return 0
        if k is not None:
            recommendations = recommendations[:k]
        if relevant_recommendation in recommendations:
            return 1 / (recommendations.index(relevant_recommendation) + 1)
        return 0


class Recall(RankingMetricABC):
    def __call__(self, recommendations, k=None, relevant_recommendations=None):
        if relevant_recommendations is None:
            return 0
        if k is not None:
            recommendations = recommendations[:k]
        return len(set(recommendations) & set(relevant_recommendations)) / len(relevant_recommendations)


class Precision(RankingMetricABC):
    def __call__(self, recommendations, k=None, relevant_recommendations=None):
        if relevant_recommendations is None:
            return 0
        if k is not None:
            recommendations = recommendations[:k]
        return len(set(recommendations) & set(relevant_recommendations)) / len(recommendations)


class FScore(RankingMetricABC):
    def __init__(self, beta=1):
        super().__init__()
        self.beta = beta
        self.precision = Precision()
        self.recall = Recall()

    def __call__(self, recommendations, k=None, relevant_recommendations=None):
        if relevant_recommendations is None:
            return 0
        p = self.precision(recommendations, relevant_recommendations=relevant_recommendations, k=k)
        r = self.recall(recommendations, relevant_recommendations=relevant_recommendations, k=k)
        return (1 + self.beta ** 2) * p * r / ((self.beta ** 2 * p) + r)


class AveragePrecision(RankingMetricABC):
    def __init__(self):
        super().__init__()
        self.precision = Precision()

    def __call__(self, recommendations, k=None, relevant_recommendations=None):
        if relevant_recommendations is None:
            return 0
        if k is not None:
            recommendations = recommendations[:k]
        total = 0
        for i, r in enumerate(recommendations, start=1):
            if r in relevant_recommendations and r not in recommendations[:i - 1]:
                total += self.precision(recommendations, relevant_recommendations=relevant_recommendations, k=i)
        if k is None:
            return total / len(relevant_recommendations)
        return total / min(len(relevant_recommendations), k)
