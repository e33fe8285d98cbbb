"""ListSampler with the reference's API and stream (DRecPy/Sampler/list_sampler.py:5-151): grouped, sorted, windowed
record sequences with targets and sampled negative ids, as Caser consumes them (caser.py:72-84).

Host logic on stdlib `random.Random` (one MT19937 stream, list_sampler.py:69) over the column store: per-group row
lists are built once (no `select()` scan per draw).  `rng.sample(set, k)` (list_sampler.py:147) depends on CPython's
iteration order of a set of numpy integer scalars; the same objects are inserted in the same order here, so the stream
is identical (checked against vectors recorded from the reference, tests/test_sampler.py).  Python >= 3.11 removed
sampling from a set: the set is then materialised with `tuple()` exactly as 3.10's `random.sample` did.
"""
import random

import numpy as np

from ..Dataset import InteractionDatasetABC


class ListSampler:
    max_consecutive_tries = 20

    def __init__(self, interaction_dataset, group_columns, neg_ratio=3, n_targets=5, negative_ids_col='iid',
                 interaction_threshold=None, sort_column=None, min_positive_records=8, max_positive_records=None,
                 seed=None):
        assert interaction_dataset is not None, 'An interaction dataset instance is required.'
        assert isinstance(interaction_dataset, InteractionDatasetABC), \
            f'Provided interaction_dataset argument is not subclass of InteractionDataset (found type {type(interaction_dataset)}).'
        assert interaction_dataset.has_internal_ids, \
            'The provided interaction dataset instance does not have internal ids assigned.'
        assert neg_ratio is not None, 'A neg_ratio value is required.'
        assert isinstance(group_columns, list) and len(group_columns) > 0, 'group_columns must be a non-empty list.'
        ds = interaction_dataset
        self.interaction_dataset = ds
        self.group_columns = group_columns
        self.neg_ratio = neg_ratio
        self.n_targets = n_targets
        self.interaction_threshold = interaction_threshold
        self.sort_column = sort_column
        self.min_positive_records = min_positive_records
        self.max_positive_records = max_positive_records
        self.negative_ids_col = negative_ids_col
        self.rng = random.Random(seed)
        self._cols = [c for c in ds.columns if c != 'rid']
        # first-appearance-ordered unique groups / negative ids (dataset.unique keeps the first occurrence)
        ug = ds.unique(group_columns)
        if len(group_columns) == 1:
            self.unique_groups = list(ug._cols[group_columns[0]])
        else:
            self.unique_groups = [[ug._cols[c][r] for c in group_columns] for r in range(len(ug))]
        self.unique_negative_ids = set(ds.unique(negative_ids_col)._cols[negative_ids_col])
        self._group_rows = {}
        keys = zip(*[ds._cols[c].tolist() for c in group_columns])
        for r, k in enumerate(keys):
            self._group_rows.setdefault(k, []).append(r)

    def _record(self, r):
        ds = self.interaction_dataset
        rec = {c: ds._cols[c][r] for c in self._cols}
        rec['rid'] = ds._rid[r]
        return rec

    def sample_group_records(self, n=16):
        ds = self.interaction_dataset
        inter = ds._cols['interaction']
        out = []
        for _ in range(n):
            tries = 0
            while True:
                tries += 1
                grp = self.rng.choice(self.unique_groups)
                key = tuple(x.item() if hasattr(x, 'item') else x for x in (grp if isinstance(grp, list) else [grp]))
                rows = self._group_rows[key]
                pos = rows if self.interaction_threshold is None else \
                    [r for r in rows if inter[r] >= self.interaction_threshold]
                if len(pos) < self.min_positive_records or \
                        (self.n_targets is not None and len(pos) < self.min_positive_records + self.n_targets):
                    if tries > self.max_consecutive_tries:
                        raise Exception(f'Failed to sample group records, max consecutive tries reached '
                                        f'({self.max_consecutive_tries}): consider reducing the min_group_records '
                                        f'({self.min_positive_records}).')
                    continue
                pos = list(pos)
                if self.sort_column is not None:
                    sc = ds._cols[self.sort_column]
                    pos.sort(key=lambda r: sc[r])
                all_pos = pos
                padding = None
                if self.max_positive_records is not None and len(pos) > self.max_positive_records:
                    if self.n_targets is None:
                        padding = self.rng.randint(0, len(pos) - self.max_positive_records)
                    else:
                        padding = self.rng.randint(0, len(pos) - self.max_positive_records - self.n_targets)
                    pos = pos[padding:padding + self.max_positive_records]
                if self.n_targets is None:
                    out.append([self._record(r) for r in pos])
                    break
                nid = ds._cols[self.negative_ids_col]
                eligible = self.unique_negative_ids.difference(set([nid[r] for r in all_pos]))
                if padding is None:
                    targets = pos[self.n_targets:]
                    pos = pos[:self.n_targets]
                else:
                    targets = all_pos[padding + self.max_positive_records:
                                      padding + self.max_positive_records + self.n_targets]
                k = self.neg_ratio * len(targets)
                if len(eligible) < k:
                    if tries > self.max_consecutive_tries:
                        raise Exception(f'Failed to sample group records, max consecutive tries reached '
                                        f'({self.max_consecutive_tries}): consider reducing the neg_ratio '
                                        f'({self.neg_ratio}) or the n_targets ({self.n_targets}).')
                    continue
                negatives = self.rng.sample(tuple(eligible), k)      # == random.sample(set, k) of CPython <= 3.10
                out.append(([self._record(r) for r in pos], [self._record(r) for r in targets], negatives))
                break
        return out
