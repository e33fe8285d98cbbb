"""ListSampler with the reference's API and stream (DRecPy/Sampler/list_sampler.py): grouped, sorted, windowed record
sequences with targets and sampled negative ids, as Caser consumes them (caser.py:72-84).

Host logic on stdlib `random.Random` (one MT19937 stream) over the column store.  The rows of every group are indexed
once, so a draw costs O(group size) instead of a dataset scan.  `rng.sample(set, k)` in the reference depends on CPython's
iteration order of a set of numpy integer scalars; `_eligible_negatives` builds the same set objects in the same order, and
`tuple(set)` is what Python <= 3.10 did internally (3.11 removed sampling from sets).  The stream is checked against
vectors recorded from the reference (tests/test_sampler.py::test_list_sampler_streams_match_reference).

RNG call order per attempt: choice(group) -> [randint(window start)] -> [sample(negatives)]; an attempt that fails a
size check is retried up to 20 times after the first (then raises).
"""
import random

import numpy as np

from ..Dataset import InteractionDatasetABC


class _Retry(Exception):
    def __init__(self, hint):
        super().__init__(hint)
        self.hint = hint


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
        assert isinstance(group_columns, list) and group_columns, 'group_columns must be a non-empty list.'
        self.interaction_dataset = ds = interaction_dataset
        self.group_columns, self.negative_ids_col = group_columns, negative_ids_col
        self.neg_ratio, self.n_targets = neg_ratio, n_targets
        self.interaction_threshold, self.sort_column = interaction_threshold, sort_column
        self.min_positive_records, self.max_positive_records = min_positive_records, max_positive_records
        self.rng = random.Random(seed)
        self._record_cols = [c for c in ds.columns if c != 'rid']

        # groups and negative ids in first-appearance order (Dataset.unique keeps first occurrences)
        firsts = ds.unique(group_columns)
        cols = [firsts._cols[c] for c in group_columns]
        self.unique_groups = list(cols[0]) if len(cols) == 1 else [list(vals) for vals in zip(*cols)]
        self.unique_negative_ids = set(ds.unique(negative_ids_col)._cols[negative_ids_col])
        # rows of every group, already in their final order: record order, or (stable) sort_column order — one vectorised
        # stable sort over the whole dataset instead of a Python sort per group on its first draw
        self._rows_of_group = {}
        self._presorted = False
        if len(group_columns) == 1 and len(ds):
            gcol = ds._cols[group_columns[0]]
            if sort_column is not None:
                order = np.lexsort((ds._cols[sort_column], gcol))        # by group, then sort key; ties keep record order
            else:
                order = np.argsort(gcol, kind='stable')
            gs = gcol[order]
            cuts = np.flatnonzero(gs[1:] != gs[:-1]) + 1
            for key, rows in zip(gs[np.concatenate([[0], cuts])].tolist(), np.split(order, cuts)):
                self._rows_of_group[(key,)] = rows.tolist()
            self._presorted = True
        else:
            for row, key in enumerate(zip(*[ds._cols[c].tolist() for c in group_columns])):
                self._rows_of_group.setdefault(key, []).append(row)
        # per-group results that do not depend on the RNG (sorted positive rows, eligible-negative tuple) are memoised:
        # a group is drawn many times over a fit (4096 draws per step over a few thousand users in examples/caser.py)
        self._memo_rows, self._memo_negs = {}, {}
        self._memo_budget = 60_000_000          # cached tuple slots (about 0.5 GB of references) before caching stops

    # ---- pieces of one draw -----------------------------------------------------------------------------------
    def _as_record(self, row):
        ds = self.interaction_dataset
        record = {c: ds._cols[c][row] for c in self._record_cols}
        record['rid'] = ds._rid[row]
        return record

    def _group_key(self, group):
        parts = group if isinstance(group, list) else [group]
        return tuple(p.item() if hasattr(p, 'item') else p for p in parts)

    def _positive_rows(self, group):
        key = self._group_key(group)
        hit = self._memo_rows.get(key)
        if hit is None:
            try:
                hit = self._positive_rows_uncached(key)
            except _Retry as retry:
                hit = retry
            self._memo_rows[key] = hit
        if isinstance(hit, _Retry):
            raise hit
        return hit

    def _positive_rows_uncached(self, key):
        rows = self._rows_of_group[key]
        if self.interaction_threshold is not None:
            values = self.interaction_dataset._cols['interaction']
            rows = [r for r in rows if values[r] >= self.interaction_threshold]
        needed = self.min_positive_records + (self.n_targets or 0)
        if len(rows) < self.min_positive_records or len(rows) < needed:
            raise _Retry(f'consider reducing the min_group_records ({self.min_positive_records}).')
        if self.sort_column is not None and not self._presorted:
            keys = self.interaction_dataset._cols[self.sort_column]
            rows = sorted(rows, key=lambda r: keys[r])           # stable, like list.sort in the reference
        return list(rows)

    def _window_start(self, n_rows):
        """Random start of the window of max_positive_records inputs (None when the group is not longer than that)."""
        limit = self.max_positive_records
        if limit is None or n_rows <= limit:
            return None
        return self.rng.randint(0, n_rows - limit - (self.n_targets or 0))

    def _eligible_negatives(self, group, all_rows):
        """tuple(set) of the candidate negative ids of a group, in CPython's set iteration order."""
        key = self._group_key(group)
        hit = self._memo_negs.get(key)
        if hit is None:
            ids = self.interaction_dataset._cols[self.negative_ids_col]
            hit = tuple(self.unique_negative_ids.difference(ids[np.asarray(all_rows, dtype=np.int64)].tolist()))
            if self._memo_budget >= len(hit):
                self._memo_budget -= len(hit)
                self._memo_negs[key] = hit
        return hit

    def _draw_once(self):
        group = self.rng.choice(self.unique_groups)
        rows = self._positive_rows(group)
        start = self._window_start(len(rows))
        inputs = rows if start is None else rows[start:start + self.max_positive_records]
        if self.n_targets is None:
            return [self._as_record(r) for r in inputs]
        if start is None:        # reference quirk: without a window the split is inputs = first T, targets = the rest
            inputs, targets = rows[:self.n_targets], rows[self.n_targets:]
        else:
            stop = start + self.max_positive_records
            targets = rows[stop:stop + self.n_targets]
        eligible = self._eligible_negatives(group, rows)
        n_negatives = self.neg_ratio * len(targets)
        if len(eligible) < n_negatives:
            raise _Retry(f'consider reducing the neg_ratio ({self.neg_ratio}) or the n_targets ({self.n_targets}).')
        negatives = self.rng.sample(eligible, n_negatives)
        return [self._as_record(r) for r in inputs], [self._as_record(r) for r in targets], negatives

    def sample_group_records(self, n=16):
        """n draws: lists of input records (n_targets None) or (input records, target records, negative ids) triples."""
        out = []
        while len(out) < n:
            failures = 0
            while True:
                try:
                    out.append(self._draw_once())
                    break
                except _Retry as retry:
                    failures += 1
                    if failures > self.max_consecutive_tries:
                        raise Exception('Failed to sample group records, max consecutive tries reached '
                                        f'({self.max_consecutive_tries}): {retry.hint}')
        return out
