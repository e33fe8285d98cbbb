from .point_sampler import PointSampler
from .list_sampler import ListSampler

__all__ = ['PointSampler', 'ListSampler']
