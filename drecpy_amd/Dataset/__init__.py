from .interaction_dataset import InteractionDataset, MemoryInteractionDataset, InteractionDatasetABC
from .movielens import load_movielens, read_ratings

__all__ = ['InteractionDataset', 'MemoryInteractionDataset', 'InteractionDatasetABC', 'load_movielens', 'read_ratings']
