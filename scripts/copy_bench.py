"""The library's streaming copy in its five forms (csrc/drx_generic.hip k_copy_f4) beside torch's copy_, 2 GiB, read + write counted."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drecpy_amd import _lib                                      # noqa: E402

L, dev = _lib.lib(), torch.device('cuda')
n = 2 * (1 << 30) // 4
src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)


def rate(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


print('torch copy_', round(rate(lambda: dst.copy_(src))), 'GB/s')
for v in range(5):
    dst.zero_()
    r = rate(lambda: _lib.check(L.drx_copy_f4_variant(_lib.ptr(dst), _lib.ptr(src), n * 4, v, _lib.stream_ptr(dev)), 'copy'))
    print('variant', v, round(r), 'GB/s', 'ok' if torch.equal(dst, src) else 'WRONG')
