"""leave_k_out split with the semantics of DRecPy/Evaluation/Splits/leave_k_out.py:14-135 (seeded, per-user
`random.Random(seed + 1 + i)`, same heap walk for `last_timestamps`), on the column store (no thread pool needed)."""
import random
from heapq import heappush, heapreplace



def leave_k_out(interaction_dataset, k=1, min_user_interactions=0, last_timestamps=False, timestamp_label='timestamp',
                seed=0, max_concurrent_threads=4, **kwds):
    assert k > 0, f'The value of k ({k}) must be > 0.'
    ratio_variant = isinstance(k, float)
    if ratio_variant and k >= 1:
        raise Exception('The k parameter should be in the (0, 1) range when it\'s used as the percentage of '
                        'interactions to sample to the test set, per user. Current value: ' + str(k))
    ds = interaction_dataset
    users = ds.unique('user').values_list('user', to_list=True)
    ucol = ds._cols['user']
    rows_of = {}
    for r, u in enumerate(ucol.tolist()):
        rows_of.setdefault(u, []).append(r)
    rid = ds._rid
    train_rem, test = [], []
    for user in users:
        seed += 1
        rng = random.Random(seed)
        user = user.item() if hasattr(user, 'item') else user
        rows = rows_of[user]
        kk = int(len(rows) * k) if ratio_variant else k
        if len(rows) < min_user_interactions:
            train_rem.extend(int(rid[r]) for r in rows)
        elif len(rows) > kk > 0:
            if last_timestamps:
                ts = ds._cols[timestamp_label]
                heap = []
                for r in rows:
                    if len(heap) < kk:
                        heappush(heap, (ts[r], int(rid[r])))
                    else:
                        heapreplace(heap, (ts[r], int(rid[r])))         # quirk kept: replaces unconditionally
                test.extend(x for _, x in heap)
            else:
                test.extend(rng.sample([int(rid[r]) for r in rows], kk))
    return ds.drop(train_rem + test), ds.drop(test, keep=True)
