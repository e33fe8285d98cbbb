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
            bounds = np.concatenate([[0], cuts, [len(order)]])
            # (numpy slices of the one sorted order; a group's rows become a Python list only if the Python draw loop asks for them)
            self._sorted_rows = (order, gs[bounds[:-1]], bounds)
            for k, key in enumerate(gs[bounds[:-1]].tolist()):
                self._rows_of_group[(key,)] = order[bounds[k]:bounds[k + 1]]
            self._presorted = True
        else:
            for row, key in enumerate(zip(*[ds._cols[c].tolist() for c in group_columns])):
                self._rows_of_group.setdefault(key, []).append(row)
        # per-group results that do not depend on the RNG (sorted positive rows, eligible-negative tuple) are memoised:
        # a group is drawn many times over a fit (4096 draws per step over a few thousand users in examples/caser.py)
        self._memo_rows, self._memo_negs = {}, {}
        self._memo_budget = 60_000_000          # cached tuple slots (about 0.5 GB of references) before caching stops
        self._native = None
        self._build_native(seed)

    # ---- native draw loop (libdrx.so, csrc/drx_host.cpp: drx_list_sampler_*) ---------------------------------------------
    def _build_native(self, seed):
        """The whole draw loop in C++ with the same MT19937 stream and CPython's set iteration order, when the configuration
        allows it: one group column, dense internal ids as negatives, an int seed.  Otherwise the Python loop below runs."""
        ds = self.interaction_dataset
        if not (self._presorted and self.negative_ids_col in ('iid', 'uid') and isinstance(seed, (int, np.integer))
                and not isinstance(seed, bool) and abs(int(seed)) < 2 ** 63 and len(ds)):
            return
        ids = np.asarray(ds._cols[self.negative_ids_col])
        n_ids = len(self.unique_negative_ids)
        if ids.min() < 0 or int(ids.max()) + 1 != n_ids or n_ids >= 2 ** 31 or len(ds) >= 2 ** 62:
            return
        keep = None
        if self.interaction_threshold is not None:
            keep = np.asarray(ds._cols['interaction']) >= self.interaction_threshold
        # every group's rows, groups in the order rng.choice indexes (first appearance), without a Python loop over the groups
        order, sorted_keys, bounds = self._sorted_rows
        pos = np.searchsorted(sorted_keys, np.asarray(self.unique_groups))
        starts, glen = bounds[pos], bounds[pos + 1] - bounds[pos]
        ends = np.cumsum(glen)
        flat = order[np.repeat(starts - (ends - glen), glen) + np.arange(int(ends[-1]) if len(ends) else 0)].astype(np.int64)
        gid = np.repeat(np.arange(len(glen)), glen)
        if keep is not None:
            sel = keep[flat]
            flat, gid = flat[sel], gid[sel]
        lens = np.bincount(gid, minlength=len(glen))
        from .. import _lib
        L = _lib.lib()
        self._n_rows_flat = np.ascontiguousarray(flat)
        indptr = np.zeros(len(lens) + 1, dtype=np.int64)
        indptr[1:] = np.cumsum(lens)
        row_ids = np.ascontiguousarray(ids[self._n_rows_flat], dtype=np.int32)
        handle = L.drx_list_sampler_create(indptr.ctypes.data, self._n_rows_flat.ctypes.data, row_ids.ctypes.data, len(lens), n_ids,
                                           int(self.neg_ratio), -1 if self.n_targets is None else int(self.n_targets),
                                           int(self.min_positive_records),
                                           -1 if self.max_positive_records is None else int(self.max_positive_records), int(seed))
        if not handle:
            return
        self._native = (L, handle)
        self._max_group = int(lens.max()) if len(lens) else 0
        self._group_values = np.asarray(self.unique_groups)
        self._twin_arrays = (indptr, row_ids, n_ids)          # what device_twin() uploads

    def __del__(self):
        native = getattr(self, '_native', None)
        if native is not None:
            native[0].drx_list_sampler_destroy(native[1])
            self._native = None

    # ---- device twin (throughput mode: libdrx.so drx_list_sample_device, a counter-based generator instead of the MT stream) --------
    def twin_host_arrays(self):
        """The arrays drx_list_sample_device reads (include/drx.h DrxListGroups), as numpy: indptr, seq_ids, held_indptr, held,
        group_value, eligible, n_ids.  Needs a window configuration (n_targets and max_positive_records == min_positive_records)."""
        assert self._native is not None and self.n_targets is not None and self.max_positive_records == self.min_positive_records, \
            'the device list sampler draws fixed windows of min_positive_records inputs + n_targets targets over one group column'
        indptr, row_ids, n_ids = self._twin_arrays
        lens = np.diff(indptr)
        gid = np.repeat(np.arange(len(lens), dtype=np.int64), lens)
        pair = np.unique(gid * np.int64(n_ids) + row_ids.astype(np.int64))            # (group, id) pairs, ascending: a group's distinct ids
        held = (pair % n_ids).astype(np.int32)
        held_indptr = np.zeros(len(lens) + 1, dtype=np.int64)
        held_indptr[1:] = np.cumsum(np.bincount(pair // n_ids, minlength=len(lens)))
        n_held = np.diff(held_indptr)
        L_, T = int(self.min_positive_records), int(self.n_targets)
        eligible = np.flatnonzero((lens >= L_ + T) & (n_ids - n_held >= T * int(self.neg_ratio))).astype(np.int32)
        if not len(eligible):
            raise Exception('Failed to sample group records: no group holds a window of '
                            f'{L_} + {T} records with {T * int(self.neg_ratio)} ids left to draw negatives from.')
        return {'indptr': indptr, 'seq_ids': row_ids, 'held_indptr': held_indptr, 'held': held,
                'group_value': np.ascontiguousarray(self._group_values, dtype=np.int32), 'eligible': eligible, 'n_ids': int(n_ids)}

    def device_twin(self, device):
        import torch
        from .. import _lib
        a = self.twin_host_arrays()
        t = {k: torch.as_tensor(v).to(device) for k, v in a.items() if k != 'n_ids'}
        G = _lib.ListGroups()
        for k in ('indptr', 'seq_ids', 'held_indptr', 'held', 'group_value', 'eligible'):
            setattr(G, k, t[k].data_ptr())
        G.n_groups, G.n_eligible, G.n_ids = len(a['indptr']) - 1, len(a['eligible']), a['n_ids']
        self._twin = (G, t, torch.device(device))
        return self

    def sample_device(self, n, seed, out=None):
        """n windows drawn on the device (see include/drx.h drx_list_sample_device): int32 tensors group value [n], input ids
        [n, min_positive_records], target ids followed by negative ids [n, n_targets * (1 + neg_ratio)].  out: the three tensors to fill."""
        import ctypes as C
        import torch
        from .. import _lib
        G, _, dev = self._twin
        L_, T, neg = int(self.min_positive_records), int(self.n_targets), int(self.neg_ratio)
        if out is not None:
            grp, before, after = out
            assert grp.numel() == n and before.shape == (n, L_) and after.shape == (n, T * (1 + neg)) and grp.dtype == torch.int32
        else:
            grp = torch.empty(n, dtype=torch.int32, device=dev)
            before = torch.empty(n, L_, dtype=torch.int32, device=dev)
            after = torch.empty(n, T * (1 + neg), dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().drx_list_sample_device(C.byref(G), n, L_, T, neg, int(seed) & (2 ** 64 - 1), grp.data_ptr(), before.data_ptr(),
                                                     after.data_ptr(), _lib.stream_ptr(dev)), 'drx_list_sample_device')
        return grp, before, after

    def sample_group_arrays(self, n=16):
        """n draws as arrays (native loop only): group values [n], offsets + dataset rows of the inputs and of the targets,
        offsets + ids of the negatives.  Same stream, same draws as sample_group_records."""
        assert self._native is not None, 'the native draw loop is not available for this configuration'
        from .. import _lib
        L, handle = self._native
        T = self.n_targets
        # widest a draw can be: a window of max_positive_records inputs + T targets, or — the reference quirk for groups not
        # longer than the window — the first T rows as inputs and ALL the others as targets
        widest = self._max_group if self.max_positive_records is None else min(self._max_group, self.max_positive_records)
        in_cap = n * max(widest, T or 0, 1)
        tg_cap = 0 if T is None else n * max(widest, T, 1)
        ng_cap = tg_cap * self.neg_ratio
        grp = np.zeros(n, np.int32)
        in_off, tg_off, ng_off = (np.zeros(n + 1, np.int64) for _ in range(3))
        in_rows, tg_rows = np.zeros(max(in_cap, 1), np.int64), np.zeros(max(tg_cap, 1), np.int64)
        negs = np.zeros(max(ng_cap, 1), np.int32)
        rc = L.drx_list_sampler_sample(handle, n, grp.ctypes.data, in_off.ctypes.data, in_rows.ctypes.data, in_cap, tg_off.ctypes.data,
                                       tg_rows.ctypes.data, tg_cap, ng_off.ctypes.data, negs.ctypes.data, ng_cap)
        if rc == -4:                                       # DRX_ERETRY
            hint = (f'consider reducing the min_group_records ({self.min_positive_records}).' if L.drx_list_sampler_last_hint(handle) == 1
                    else f'consider reducing the neg_ratio ({self.neg_ratio}) or the n_targets ({self.n_targets}).')
            raise Exception('Failed to sample group records, max consecutive tries reached '
                            f'({self.max_consecutive_tries}): {hint}')
        _lib.check(rc, 'drx_list_sampler_sample')
        return (self._group_values[grp], in_off, in_rows[:in_off[-1]], tg_off, tg_rows[:tg_off[-1]], ng_off, negs[:ng_off[-1]])

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
        if isinstance(rows, np.ndarray):
            rows = rows.tolist()
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
            # a SET argument, as list_sampler.py:130 passes: CPython's set.difference picks copy-and-discard or rebuild-by-adding
            # from the argument's type and size, and the two can leave different table layouts (= iteration orders)
            hit = tuple(self.unique_negative_ids.difference(set(ids[np.asarray(all_rows, dtype=np.int64)].tolist())))
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
        if self._native is not None:
            _, in_off, in_rows, tg_off, tg_rows, ng_off, negs = self.sample_group_arrays(n)
            ids = self.interaction_dataset._cols[self.negative_ids_col]
            out = []
            for d in range(n):
                inputs = [self._as_record(r) for r in in_rows[in_off[d]:in_off[d + 1]]]
                if self.n_targets is None:
                    out.append(inputs)
                else:
                    # negatives as elements of the id column, like the members of unique_negative_ids
                    out.append((inputs, [self._as_record(r) for r in tg_rows[tg_off[d]:tg_off[d + 1]]],
                                [ids.dtype.type(x) for x in negs[ng_off[d]:ng_off[d + 1]]]))
            return out
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
