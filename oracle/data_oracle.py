"""CPU restatement of the reference's data path for the fit() loop (TEST INFRASTRUCTURE, see
oracle/__init__.py).  Pure Python + numpy; stdlib `random.Random` IS CPython's MT19937 and is not
part of the reference, so the streams below are bit-exact by construction.

Follows (all paths under /root/reference):
  * id map ................ DRecPy/Dataset/mem_dataset.py:309-330  (first-appearance codes)
  * interaction matrix .... DRecPy/Dataset/mem_dataset.py:165-176, 480-498 (CSR, duplicates summed)
  * positive generator .... DRecPy/Dataset/mem_dataset.py:111-129
  * null-pair generator ... DRecPy/Dataset/mem_dataset.py:131-163
  * PointSampler .......... DRecPy/Sampler/point_sampler.py:19-96
  * ListSampler ........... DRecPy/Sampler/list_sampler.py:33-151
  * corruption stream ..... DRecPy/Recommender/cdae.py:59-65 (`self._rng.uniform(0, 1)` per item)
"""
import random

import numpy as np


# ----------------------------------------------------------------------------------------------
# id map (mem_dataset.py:309-330): uid/iid = rank of first appearance in row order
# ----------------------------------------------------------------------------------------------
def first_appearance_codes(raw):
    """raw: sequence of hashables.  Returns (codes int64[n], categories list) exactly like
    `pd.Categorical(col, categories=col.unique()).codes` (mem_dataset.py:314-316)."""
    mapping = {}
    codes = np.empty(len(raw), dtype=np.int64)
    cats = []
    for r, x in enumerate(raw):
        c = mapping.get(x)
        if c is None:
            c = len(cats)
            mapping[x] = c
            cats.append(x)
        codes[r] = c
    return codes, cats


def interaction_csr(uid, iid, val, n_users=None, n_items=None):
    """CSR [U,N] of float64 interaction values, duplicates summed, columns sorted
    (scipy csr semantics used by mem_dataset.py:480-498 + `.toarray()` at cdae.py:61)."""
    uid = np.asarray(uid, dtype=np.int64)
    iid = np.asarray(iid, dtype=np.int64)
    val = np.asarray(val, dtype=np.float64)
    U = int(uid.max()) + 1 if n_users is None else n_users
    N = int(iid.max()) + 1 if n_items is None else n_items
    key = uid * N + iid
    order = np.argsort(key, kind='stable')
    ks = key[order]
    uniq, start = np.unique(ks, return_index=True)
    sums = np.add.reduceat(val[order], start) if len(ks) else np.zeros(0)
    rows = uniq // N
    cols = uniq % N
    indptr = np.zeros(U + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    indptr = np.cumsum(indptr)
    return indptr, cols.astype(np.int64), sums


def user_interaction_vec(indptr, cols, vals, uid, n_items):
    v = np.zeros(n_items, dtype=np.float64)
    s, e = indptr[uid], indptr[uid + 1]
    v[cols[s:e]] = vals[s:e]
    return v


# ----------------------------------------------------------------------------------------------
# PointSampler (point_sampler.py:44-61) over the in-memory backend
# ----------------------------------------------------------------------------------------------
class PointSamplerOracle:
    """Three identically seeded MT19937 streams (point_sampler.py:32, mem_dataset.py:115,136).

    negative: uniform (u,i) with NO dataframe row (u,i), whatever its value — the "existing null
      pair" branch is dead for the in-memory backend (mem_dataset.py:141-146: the generator is
      handed a DataFrame as `query`, the parse error is swallowed by the bare `except`).
    positive: uniform user, then uniform row (dataframe order) of that user among rows with
      interaction >= thr (all rows when thr is None, point_sampler.py:37-42); users without such
      rows are re-drawn (mem_dataset.py:121-122).
    """

    def __init__(self, uid, iid, val, neg_ratio, interaction_threshold=None, seed=None):
        self.uid = np.asarray(uid, dtype=np.int64)
        self.iid = np.asarray(iid, dtype=np.int64)
        self.val = np.asarray(val)
        self.neg_ratio = neg_ratio
        self.r_sel = random.Random(seed)
        self.r_neg = random.Random(seed)
        self.r_pos = random.Random(seed)
        self.max_uid = int(self.uid.max())
        self.max_iid = int(self.iid.max())
        self.pairs = set(zip(self.uid.tolist(), self.iid.tolist()))
        keep = np.ones(len(self.uid), bool) if interaction_threshold is None \
            else (self.val >= interaction_threshold)
        self.user_rows = [[] for _ in range(self.max_uid + 1)]
        for r in np.nonzero(keep)[0].tolist():
            self.user_rows[int(self.uid[r])].append(r)

    def sample_negative(self):
        while True:
            u = self.r_neg.randint(0, self.max_uid)
            i = self.r_neg.randint(0, self.max_iid)
            if (u, i) not in self.pairs:
                return u, i, 0

    def sample_positive(self):
        while True:
            u = self.r_pos.randint(0, self.max_uid)
            rows = self.user_rows[u]
            if not rows:
                continue
            r = rows[self.r_pos.randint(0, len(rows) - 1)]
            return int(self.uid[r]), int(self.iid[r]), self.val[r]

    def sample(self, n=16):
        out = []
        while len(out) != n:
            null_pair = self.r_sel.uniform(0, self.neg_ratio + 1) > 1
            out.append(self.sample_negative() if null_pair else self.sample_positive())
        return out


# ----------------------------------------------------------------------------------------------
# ListSampler as Caser configures it (list_sampler.py:74-151, caser.py:72-75)
# ----------------------------------------------------------------------------------------------
class ListSamplerOracle:
    """One MT stream (list_sampler.py:69).  Records are row indices into the frame.

    `rng.sample(set, k)` (list_sampler.py:147) depends on CPython-3.10's iteration order of a set of
    numpy int scalars built exactly like the reference builds it: `set(unique values in first-
    appearance order).difference(set(ids of the group's positives))`; this restatement builds the
    same Python objects in the same order, so the order is identical by construction.
    """
    max_consecutive_tries = 20

    def __init__(self, frame, group_column, neg_ratio=3, n_targets=5, negative_ids_col='iid',
                 interaction_threshold=None, sort_column=None, min_positive_records=8,
                 max_positive_records=None, seed=None):
        self.frame = frame                      # dict column -> numpy array (same dtypes as the df)
        self.group_column = group_column
        self.neg_ratio = neg_ratio
        self.n_targets = n_targets
        self.negative_ids_col = negative_ids_col
        self.thr = interaction_threshold
        self.sort_column = sort_column
        self.min_pos = min_positive_records
        self.max_pos = max_positive_records
        self.rng = random.Random(seed)
        g = frame[group_column]
        _, first = np.unique(g, return_index=True)
        self.unique_groups = [g[i] for i in sorted(first.tolist())]
        nid = frame[negative_ids_col]
        _, first = np.unique(nid, return_index=True)
        self.unique_negative_ids = set(nid[i] for i in sorted(first.tolist()))
        self.group_rows = {}
        for r, x in enumerate(g.tolist()):
            self.group_rows.setdefault(x, []).append(r)

    def sample_group_records(self, n=16):
        out = []
        f = self.frame
        for _ in range(n):
            tries = 0
            while True:
                tries += 1
                grp = self.rng.choice(self.unique_groups)
                rows = self.group_rows[grp.item() if hasattr(grp, 'item') else grp]
                pos = rows if self.thr is None else [r for r in rows if f['interaction'][r] >= self.thr]
                if len(pos) < self.min_pos or \
                        (self.n_targets is not None and len(pos) < self.min_pos + self.n_targets):
                    if tries > self.max_consecutive_tries:
                        raise Exception('Failed to sample group records, max consecutive tries reached')
                    continue
                pos = list(pos)
                if self.sort_column is not None:
                    pos.sort(key=lambda r: f[self.sort_column][r])
                all_pos = pos
                padding = None
                if self.max_pos is not None and len(pos) > self.max_pos:
                    if self.n_targets is None:
                        padding = self.rng.randint(0, len(pos) - self.max_pos)
                    else:
                        padding = self.rng.randint(0, len(pos) - self.max_pos - self.n_targets)
                    pos = pos[padding:padding + self.max_pos]
                if self.n_targets is None:
                    out.append(pos)
                    break
                eligible = self.unique_negative_ids.difference(
                    set([f[self.negative_ids_col][r] for r in all_pos]))
                if padding is None:
                    targets = pos[self.n_targets:]
                    pos = pos[:self.n_targets]
                else:
                    targets = all_pos[padding + self.max_pos:padding + self.max_pos + self.n_targets]
                k = self.neg_ratio * len(targets)
                if len(eligible) < k:
                    if tries > self.max_consecutive_tries:
                        raise Exception('Failed to sample group records, max consecutive tries reached')
                    continue
                negs = self.rng.sample(eligible, k)
                out.append((pos, targets, [int(x) for x in negs]))
                break
        return out


# ----------------------------------------------------------------------------------------------
# CDAE corruption stream (cdae.py:63): one `uniform(0,1)` per item n = 0..N-1, rows in batch order
# ----------------------------------------------------------------------------------------------
def corruption_keep_mask(rng, n_rows, n_items, corruption_level):
    """Returns bool[n_rows, n_items]: True where the input survives (uniform(0,1) >= q)."""
    keep = np.empty((n_rows, n_items), dtype=bool)
    for b in range(n_rows):
        for n in range(n_items):
            keep[b, n] = not (rng.uniform(0, 1) < corruption_level)
    return keep


def list_sample_counter(twin, n, n_inputs, n_targets, neg_ratio, seed):
    """CPU restatement of drx_list_sample_device (include/drx.h): n windows from the counter-based generator drx_hash_u32(seed, d, k).
    twin: the arrays of ListSampler.twin_host_arrays().  Returns (group values [n], before [n, n_inputs], after [n, T * (1 + neg)]).
    Distribution of the reference's ListSampler as Caser configures it (list_sampler.py:74-151, caser.py:72-75): uniform group among
    those that can yield a window, uniform window start, negatives uniform without replacement among the ids the group does not hold."""
    from oracle.cdae_oracle import drx_hash_u32

    def h(d, k):
        return int(drx_hash_u32(seed, np.uint32(d), np.uint32(k)))

    def complement_at(held, j):
        lo, hi = 0, len(held)
        while lo < hi:
            mid = (lo + hi) // 2
            if int(held[mid]) - mid <= j:
                lo = mid + 1
            else:
                hi = mid
        return j + lo
    L, T = n_inputs, n_targets
    grp = np.zeros(n, np.int32)
    before = np.zeros((n, L), np.int32)
    after = np.zeros((n, T * (1 + neg_ratio)), np.int32)
    for d in range(n):
        g = int(twin['eligible'][(h(d, 0) * len(twin['eligible'])) >> 32])
        r0, r1 = int(twin['indptr'][g]), int(twin['indptr'][g + 1])
        start = (h(d, 1) * (r1 - r0 - L - T + 1)) >> 32
        grp[d] = twin['group_value'][g]
        before[d] = twin['seq_ids'][r0 + start:r0 + start + L]
        after[d, :T] = twin['seq_ids'][r0 + start + L:r0 + start + L + T]
        held = twin['held'][int(twin['held_indptr'][g]):int(twin['held_indptr'][g + 1])]
        n_pop = twin['n_ids'] - len(held)
        picked = []
        for i in range(T * neg_ratio):
            dup, a, j, v = True, 0, 0, 0
            while a < 16 and dup:
                j = (h(d, 2 + 16 * i + a) * n_pop) >> 32
                v = complement_at(held, j)
                dup = v in picked
                a += 1
            while dup:
                j = 0 if j + 1 == n_pop else j + 1
                v = complement_at(held, j)
                dup = v in picked
            picked.append(v)
            after[d, T + i] = v
    return grp, before, after
