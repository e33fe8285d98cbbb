from .recommender_abc import RecommenderABC
from .cdae import CDAE
from .caser import Caser
from .dmf import DMF
from .early_stopping import EarlyStoppingRuleABC, MaxValidationValueRule
from .trainables import TrainableLayer, TrainableModel, Variable

__all__ = ['RecommenderABC', 'CDAE', 'Caser', 'DMF', 'EarlyStoppingRuleABC', 'MaxValidationValueRule', 'Variable',
           'TrainableLayer', 'TrainableModel']
