"""Seeded synthetic interaction sets shaped like the reference's benchmark data (S1 ml-100k-like,
S2 ml-1m-like, S3 10M users x 1M items x ~200M interactions).  No MovieLens files exist offline.

Integer-only and counter-based: user u's degree and items depend only on (seed, u), so any contiguous user shard of
the same set can be generated independently on any rank/device (CPU and GPU give identical bits).  Output is the
positives CSR the engine trains on (implicit value 1, duplicates merged, columns ascending).
"""
import math

import numpy as np
import torch

_M63 = (1 << 63) - 1


def _srl(x, k):
    """logical right shift on int64 tensors"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _i64(c):
    """python int -> wrapped int64 constant"""
    c &= (1 << 64) - 1
    return c - (1 << 64) if c >= (1 << 63) else c


def hash63(seed, a, b):
    """splitmix-style mix of (seed, a, b) -> int64 in [0, 2^63).  a, b: int64 tensors (broadcastable)."""
    x = a * _i64(0x9E3779B97F4A7C15) + b * _i64(0xD1B54A32D192ED03) + _i64(seed)
    x = (x ^ _srl(x, 30)) * _i64(0xBF58476D1CE4E5B9)
    x = (x ^ _srl(x, 27)) * _i64(0x94D049BB133111EB)
    x = x ^ _srl(x, 31)
    return x & _M63


def _norm_cdf(z):
    return 0.5 * (1.0 + math.erf(z / math.sqrt(2.0)))


def degree_thresholds(mean_deg, min_deg, sigma=1.0, cap=4000):
    """int64 thresholds T[x] = floor(P(X <= x) * 2^63) for X = floor(LogNormal(mu, sigma)), E[X] ~ mean_deg - min_deg."""
    ex = max(mean_deg - min_deg, 0.5)
    mu = math.log(ex) - 0.5 * sigma * sigma
    t = [min(int(_norm_cdf((math.log(x + 1.0) - mu) / sigma) * float(1 << 63)), _M63) for x in range(cap)]
    return torch.tensor(t, dtype=torch.int64)


def zipf_thresholds(n_items, alpha):
    w = 1.0 / np.arange(1, n_items + 1, dtype=np.float64) ** alpha
    c = np.cumsum(w)
    c /= c[-1]
    t = np.minimum((c * float(1 << 63)), float(_M63 - 1024)).astype(np.int64)
    t[-1] = _M63
    return torch.from_numpy(t)


def _coprime_mult(n):
    a = 2654435761 % n
    if a < 2:
        a = 2 if n > 2 else 1
    while math.gcd(a, n) != 1:
        a += 1
    return a


def synth_history(n_users_total, n_items, mean_deg=20, min_deg=5, zipf_alpha=1.05, seed=0, device='cpu',
                  user_lo=0, user_hi=None, chunk_pairs=1 << 25):
    """Positives CSR of users [user_lo, user_hi) of the synthetic set.  Returns (indptr int64[n+1], indices int32[nnz])."""
    user_hi = n_users_total if user_hi is None else user_hi
    dev = torch.device(device)
    dth = degree_thresholds(mean_deg, min_deg).to(dev)
    zth = zipf_thresholds(n_items, zipf_alpha).to(dev)
    mult = _coprime_mult(n_items)
    n_local = user_hi - user_lo
    users_per_chunk = max(1, int(chunk_pairs // max(mean_deg, 1)))
    counts = []
    cols = []
    for lo in range(user_lo, user_hi, users_per_chunk):
        hi = min(user_hi, lo + users_per_chunk)
        u = torch.arange(lo, hi, dtype=torch.int64, device=dev)
        deg = min_deg + torch.searchsorted(dth, hash63(seed, u, torch.zeros_like(u)), right=True)
        deg = torch.clamp(deg, max=n_items)
        off = torch.cumsum(deg, 0)
        total = int(off[-1].item())
        row = torch.repeat_interleave(torch.arange(hi - lo, dtype=torch.int64, device=dev), deg)
        j = torch.arange(total, dtype=torch.int64, device=dev) - (off - deg)[row]
        rank = torch.searchsorted(zth, hash63(seed, u[row], j + 1), right=True).clamp_(max=n_items - 1)
        item = (rank * mult + 12345) % n_items
        key = torch.unique(row * n_items + item, sorted=True)          # merges duplicate (u, i), sorts rows/cols
        r = key // n_items
        counts.append(torch.bincount(r, minlength=hi - lo))
        cols.append((key - r * n_items).to(torch.int32))
        del row, j, rank, item, key, r
    cnt = torch.cat(counts)
    indptr = torch.zeros(n_local + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(cnt, 0)
    return indptr, torch.cat(cols)


SHAPES = {
    # name: (n_users, n_items, mean_deg, min_deg, zipf_alpha)
    # mean_deg is the pre-merge draw count, calibrated so that the merged sets have the reference's densities
    'ml-100k': (943, 1682, 176, 20, 1.0),       # ~90.5k positives (ml-100k train: 90 570)
    'ml-1m': (6040, 3706, 300, 20, 1.0),        # ~940k positives (ml-1m train: 939 809)
    'synth-10m': (10_000_000, 1_000_000, 23.4, 5, 1.05),   # ~200M positives
}


def dataset(shape='ml-100k', seed=0, extra_per_user=0, with_timestamp=True):
    """InteractionDataset shaped like a MovieLens set (users, items, ratings 1..5, shuffled rows, timestamps): the
    offline stand-in for DRecPy's get_train_dataset('ml-100k') etc. (integrated_datasets.py:111), which download."""
    from .Dataset import InteractionDataset
    U, N, md, mn, a = SHAPES[shape]
    ip, idx = synth_history(U, N, md + extra_per_user, mn, a, seed=seed)
    ip, idx = ip.numpy(), idx.numpy()
    rng = np.random.RandomState(seed)
    user = np.repeat(np.arange(U, dtype=np.int64), np.diff(ip)) + 1
    perm = rng.permutation(len(user))
    cols = {'user': user[perm], 'item': (idx.astype(np.int64) + 1)[perm], 'interaction': rng.randint(1, 6, size=len(user))[perm]}
    if with_timestamp:
        cols['timestamp'] = rng.randint(0, 10 ** 9, size=len(user))[perm]
    return InteractionDataset.read_df(cols, verbose=False)
