from .recommender_abc import RecommenderABC
from .cdae import CDAE
from .caser import Caser
from .early_stopping import EarlyStoppingRuleABC, MaxValidationValueRule

__all__ = ['RecommenderABC', 'CDAE', 'Caser', 'EarlyStoppingRuleABC', 'MaxValidationValueRule']
