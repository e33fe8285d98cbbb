"""DMF all-pairs scoring + per-user top-k: the two launches of today (k_score_pairs_bf16 writes [n_u, n_i] fp32 scores, drx_topk reads them back)
timed apart, 2048 users x 3706 items (the ml-1m shape), k = 10."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drecpy_amd import _lib                                      # noqa: E402

L, dev = _lib.lib(), torch.device('cuda')
n_u, n_i, k = 2048, 3706, 10
ru = torch.nn.functional.normalize(torch.randn(n_u, 64, device=dev), dim=1)
ri = torch.nn.functional.normalize(torch.randn(n_i, 64, device=dev), dim=1)
pitch = (n_i + 31) // 32 * 32
sc = torch.empty(n_u, pitch, device=dev)
oi = torch.empty(n_u, k, dtype=torch.int32, device=dev)
ov = torch.empty(n_u, k, device=dev)


def ev(fn, n=200):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


score = lambda: _lib.check(L.drx_score_pairs_bf16(_lib.ptr(ru), n_u, _lib.ptr(ri), n_i, 64, 32, None, _lib.ptr(sc), pitch, _lib.stream_ptr(dev)), 'score')
flat = sc[:, :n_i].contiguous()
sb = L.drx_topk_scratch_bytes_k(n_u, n_i, k)
scr = torch.empty(max(sb, 16), dtype=torch.uint8, device=dev)
topk = lambda: _lib.check(L.drx_topk(_lib.ptr(flat), None, n_u, n_i, k, _lib.ptr(oi), _lib.ptr(ov), _lib.ptr(scr) if sb else None, sb, _lib.stream_ptr(dev)), 'topk')
print('k_score_pairs_bf16', round(ev(score), 1), 'us;  drx_topk (k = 10)', round(ev(topk), 1), 'us;  both', round(ev(lambda: (score(), topk())), 1), 'us')
for kk in (1, 4, 10, 32, 64, 100):
    oi2 = torch.empty(n_u, kk, dtype=torch.int32, device=dev)
    ov2 = torch.empty(n_u, kk, device=dev)
    t = ev(lambda: _lib.check(L.drx_topk(_lib.ptr(flat), None, n_u, n_i, kk, _lib.ptr(oi2), _lib.ptr(ov2), None, 0, _lib.stream_ptr(dev)), 'topk'), 50)
    print('  drx_topk k =', kk, round(t, 1), 'us')
for rows in (256, 1024):
    t = ev(lambda: _lib.check(L.drx_topk(_lib.ptr(flat), None, rows, n_i, 10, _lib.ptr(oi), _lib.ptr(ov), None, 0, _lib.stream_ptr(dev)), 'topk'), 50)
    print('  drx_topk k = 10, rows =', rows, round(t, 1), 'us')
