"""PointSampler with the reference's API and streams (DRecPy/Sampler/point_sampler.py:5-96).

The draws come from the C++ host sampler in libdrx.so (CPython-exact MT19937, three identically seeded streams,
CSR / sorted-pair membership instead of the O(nnz) pandas scans of mem_dataset.py:121,161); the streams are
bit-identical to the reference's (tests/test_sampler.py checks them against vectors recorded from the reference).
"""
import random

import numpy as np

from .. import _lib
from ..Dataset import InteractionDatasetABC


class _HostSampler:
    """Thin owner of a DrxSampler handle."""

    def __init__(self, ds, neg_ratio, interaction_threshold, seed):
        L = _lib.lib()
        self._L = L
        self._uid = np.ascontiguousarray(ds._cols['uid'], dtype=np.int32)
        self._iid = np.ascontiguousarray(ds._cols['iid'], dtype=np.int32)
        self._val = np.ascontiguousarray(ds._cols['interaction'], dtype=np.float64)
        self._raw_val = ds._cols['interaction']
        if seed is None:
            seed = random.getrandbits(62)
        has_thr = interaction_threshold is not None
        self._h = L.drx_sampler_create(self._uid.ctypes.data, self._iid.ctypes.data, self._val.ctypes.data, len(self._uid),
                                       int(neg_ratio), int(has_thr), float(interaction_threshold if has_thr else 0.0),
                                       int(seed))
        if not self._h:
            raise _lib.DrxError('drx_sampler_create failed')

    def draw(self, kind, n):
        u = np.empty(n, dtype=np.int32)
        i = np.empty(n, dtype=np.int32)
        v = np.empty(n, dtype=np.float64)
        neg = np.empty(n, dtype=np.uint8)
        _lib.check(self._L.drx_sampler_draw(self._h, kind, n, u.ctypes.data, i.ctypes.data, v.ctypes.data,
                                            neg.ctypes.data), 'drx_sampler_draw')
        return u, i, v, neg

    def __del__(self):
        if getattr(self, '_h', None):
            self._L.drx_sampler_destroy(self._h)
            self._h = None


class PointSampler:
    """Samples positive and negative (uid, iid, interaction) triples.

    Args: as the reference (point_sampler.py:8-18): interaction_dataset, neg_ratio, interaction_threshold=None, seed=None.
    """

    def __init__(self, interaction_dataset, neg_ratio, interaction_threshold=None, seed=None):
        assert interaction_dataset is not None, 'An interaction dataset instance is required.'
        assert neg_ratio is not None, 'A neg_ratio value is required.'
        assert isinstance(interaction_dataset, InteractionDatasetABC), \
            f'Provided interaction_dataset argument is not subclass of InteractionDataset (found type {type(interaction_dataset)}).'
        assert interaction_dataset.has_internal_ids, \
            'The provided interaction dataset instance does not have internal ids assigned.'
        self.interaction_dataset = interaction_dataset
        self.neg_ratio = neg_ratio
        self.interaction_threshold = interaction_threshold
        # the native sampler (copies of the id columns, the pair-membership structure, the positives' CSR: 0.6 s at 20 M rows) is built
        # on first use: a fit() that draws its triples on the device (CDAE mode='sampled', device_sampler=True) never needs it
        if seed is None:
            seed = random.getrandbits(62)
        self._host_args = (interaction_dataset, neg_ratio, interaction_threshold, seed)
        self._host_obj = None
        kind = interaction_dataset._cols['interaction'].dtype
        self._val_type = kind.type

    @property
    def _host(self):
        if self._host_obj is None:
            self._host_obj = _HostSampler(*self._host_args)
        return self._host_obj

    def sample_arrays(self, n=16):
        """(uid int32[n], iid int32[n], value float64[n], is_negative uint8[n]) without building Python tuples."""
        return self._host.draw(0, n)

    def sample(self, n=16):
        u, i, v, neg = self._host.draw(0, n)
        # negatives carry the int 0, positives the raw interaction value (point_sampler.py:83,96)
        vt = self._val_type
        return [(a, b, 0) if ng else (a, b, vt(c)) for a, b, c, ng in zip(u.tolist(), i.tolist(), v.tolist(), neg.tolist())]

    def sample_one(self):
        return self.sample(n=1)[0]

    def sample_negative(self):
        u, i, _, _ = self._host.draw(1, 1)
        return int(u[0]), int(i[0]), 0

    def sample_positive(self):
        u, i, v, _ = self._host.draw(2, 1)
        return int(u[0]), int(i[0]), self._val_type(v[0])
