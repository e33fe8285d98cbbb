"""Shared helpers for tests: golden-frame loading and oracle-side dataset preparation."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_frames():
    z = np.load(os.path.join(GOLDEN, 'frames.npz'), allow_pickle=False)
    frames = {}
    for key in z.files:
        k, c = key.split('__')
        frames.setdefault(k, {})[c] = z[key]
    return frames


# ---- synthetic CDAE problems shared by the GPU parity tests -------------------------------------
def hash_u32(seed, a, b):
    """numpy restatement of drx_hash_u32 (include/drx.h) for uint32 arrays a, b."""
    M = (1 << 64) - 1
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    with np.errstate(over='ignore'):
        x = (np.uint64(seed & M) + a * np.uint64(0x9E3779B97F4A7C15) + b * np.uint64(0xD1B54A32D192ED03))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(32)).astype(np.uint32)


def q_threshold(q):
    t = float(np.float32(q)) * 4294967296.0
    return 0 if t <= 0 else min(int(t), 0xFFFFFFFF)


def synth_history(rng, n_users, n_items, mean_deg, zipf=1.0, min_deg=0):
    """Random positives CSR (sorted, unique columns per row)."""
    pop = 1.0 / np.arange(1, n_items + 1) ** zipf
    pop /= pop.sum()
    indptr = [0]
    idx = []
    for u in range(n_users):
        d = min(n_items, max(min_deg, int(rng.poisson(mean_deg))))
        cols = np.sort(rng.choice(n_items, size=d, replace=False, p=pop)) if d else np.zeros(0, np.int64)
        idx.append(cols)
        indptr.append(indptr[-1] + d)
    return np.asarray(indptr, np.int64), (np.concatenate(idx) if idx else np.zeros(0)).astype(np.int32)


def batch_rows(indptr, indices, uids, n_items, keep_flat=None):
    """Dense targets t [B,N], keep offsets and (optionally) the kept-item lists per row."""
    B = len(uids)
    t = np.zeros((B, n_items), dtype=bool)
    keep_off = np.zeros(B + 1, dtype=np.int32)
    kept = []
    for b, u in enumerate(uids):
        s, e = indptr[u], indptr[u + 1]
        t[b, indices[s:e]] = True
        keep_off[b + 1] = keep_off[b] + (e - s)
        if keep_flat is not None:
            kf = keep_flat[keep_off[b]:keep_off[b + 1]].astype(bool)
            kept.append(indices[s:e][kf].tolist())
    return t, keep_off, kept


def x_tilde(t, kept, q, dtype):
    x = np.zeros(t.shape, dtype=dtype)
    for b, k in enumerate(kept):
        x[b, k] = 1.0 / (1.0 - q)
    return x


def new_rendezvous(tmp_path):
    """A fresh `init_method` for torch.distributed: a file store under the test's own tmp_path.  No TCP port is chosen at all, so
    nothing can collide with the ephemeral range (32768-60999) gloo's own pair connections draw from — GPUTEST_r03 went red on a
    port COMPUTED into that range.  Every call names a new file: a repeated test never meets a stale store."""
    import uuid
    return f'file://{tmp_path}/rdzv_{uuid.uuid4().hex}'


_INFRA = ('EADDRINUSE', 'Address already in use', 'Connection reset', 'Connection refused', 'connectFullMesh', 'Socket Timeout',
          'timed out', 'Timed out', 'store', 'rendezvous')


def retry_infra(fn):
    """For the tests that run N ranks as N processes on one box: ONE repetition, and only when the failure is the process group's
    plumbing (rendezvous, sockets, timeouts) — never for an AssertionError or any mismatch, which fail at once (those tests exist to
    catch ordering races).  The first failure's full traceback is printed before the repeat, and the repeat builds a new rendezvous
    (the test body calls new_rendezvous again)."""
    import functools
    import traceback

    @functools.wraps(fn)
    def wrapper(*a, **k):
        try:
            return fn(*a, **k)
        except Exception as e:                      # noqa: BLE001
            text = ''.join(traceback.format_exception(type(e), e, e.__traceback__))
            if 'AssertionError' in text or 'Mismatch' in text or not any(p in text for p in _INFRA):
                raise
            print(f'[retry_infra] {fn.__name__}: first attempt failed in the process-group plumbing; full traceback:\n{text}',
                  flush=True)
            return fn(*a, **k)
    return wrapper
