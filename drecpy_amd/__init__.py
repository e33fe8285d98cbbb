"""drecpy_amd — MI355X-native engine for DRecPy's deep-recommender hot path (RecommenderABC.fit -> CDAE).

Mirrors the reference package layout for the path it accelerates: `drecpy_amd.Recommender` (RecommenderABC, CDAE),
`drecpy_amd.Sampler` (PointSampler), `drecpy_amd.Dataset` (InteractionDataset).  All device arithmetic runs in
libdrx.so (hand-written HIP for gfx950) through the C ABI of include/drx.h; there is no CPU fallback.
"""
__version__ = '0.1.0'
