"""k_score_pairs_bf16 at config 3's size (2048 users x 3706 items, 32 factors) for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402
from drecpy_amd import _lib                    # noqa: E402

L = _lib.lib()
n_u, n_i = 2048, 3706
ru = torch.nn.functional.normalize(torch.randn(n_u, 64, device='cuda'), dim=1)
ri = torch.nn.functional.normalize(torch.randn(n_i, 64, device='cuda'), dim=1)
out = torch.empty(n_u, (n_i + 31) // 32 * 32, device='cuda')
for _ in range(200):
    _lib.check(L.drx_score_pairs_bf16(_lib.ptr(ru), n_u, _lib.ptr(ri), n_i, 64, 32, None, _lib.ptr(out), out.shape[1], _lib.stream_ptr()), 'score')
torch.cuda.synchronize()
print('ok', float(out[0, 0]))
