from .metrics import (DCG, NDCG, AveragePrecision, FScore, HitRatio, Precision, RankingMetricABC, Recall,
                      ReciprocalRank)
from .ranking_evaluation import ranking_evaluation

__all__ = ['ranking_evaluation', 'RankingMetricABC', 'DCG', 'NDCG', 'HitRatio', 'ReciprocalRank', 'Recall', 'Precision',
           'FScore', 'AveragePrecision']
