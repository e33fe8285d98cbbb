"""How much of a rank's row request at step s+1 was updated by step s (any rank)?  The rows that were NOT could be gathered and sent
while step s still computes; the ones that were must wait for step s's owner apply.  NumPy model of the
synth-10m batch statistics: per rank B triples of distinct users, a log-normal history of Zipf(1.05) items per user, corruption 0.2,
one output item per triple (1 positive : 5 uniform negatives).  python scripts/late_fraction.py [world]"""
import sys

import numpy as np

N, B, ALPHA = 1_000_000, 65536, 1.05
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(0)
cdf = np.cumsum(1.0 / np.arange(1, N + 1) ** ALPHA)
cdf /= cdf[-1]


def zipf(n):
    return np.minimum(np.searchsorted(cdf, rng.random(n)), N - 1)


def rank_request():
    """distinct W rows (kept history items) and W2T rows (output items) of one rank's batch: keys [0, N) and [N, 2N)"""
    mu = np.log(18.4) - 0.5
    deg = 5 + np.floor(rng.lognormal(mu, 1.0, B)).astype(np.int64)
    items = zipf(int(deg.sum()))
    kept = items[rng.random(items.size) >= 0.2]
    pos = rng.random(B) < 1.0 / 6.0
    out = np.where(pos, zipf(B), rng.integers(0, N, B))
    return np.unique(np.concatenate([kept, N + out]))


for w in sorted({1, 2, 4, world}):
    prev = np.unique(np.concatenate([rank_request() for _ in range(w)]))        # rows updated by step s: the union over ranks
    mine = rank_request()                                                       # rank 0's request at step s + 1
    late = np.isin(mine, prev, assume_unique=True).mean()
    print(f'world {w}: rank request {mine.size} rows, step-s update set {prev.size} rows, late fraction {late:.3f}')
