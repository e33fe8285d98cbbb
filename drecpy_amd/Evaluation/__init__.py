from .metrics import (DCG, NDCG, AveragePrecision, FScore, HitRatio, Precision, RankingMetricABC, Recall,
                      ReciprocalRank)
from .ranking_evaluation import ranking_evaluation
from .recommendation_evaluation import recommendation_evaluation
from .splits import leave_k_out

__all__ = ['ranking_evaluation', 'recommendation_evaluation', 'leave_k_out', 'RankingMetricABC', 'DCG', 'NDCG', 'HitRatio', 'ReciprocalRank', 'Recall', 'Precision',
           'FScore', 'AveragePrecision']
