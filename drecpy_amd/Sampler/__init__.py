from .point_sampler import PointSampler

__all__ = ['PointSampler']
