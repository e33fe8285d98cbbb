from .recommender_abc import RecommenderABC
from .cdae import CDAE
from .early_stopping import EarlyStoppingRuleABC, MaxValidationValueRule

__all__ = ['RecommenderABC', 'CDAE', 'EarlyStoppingRuleABC', 'MaxValidationValueRule']
