from .interaction_dataset import InteractionDataset, MemoryInteractionDataset, InteractionDatasetABC

__all__ = ['InteractionDataset', 'MemoryInteractionDataset', 'InteractionDatasetABC']
