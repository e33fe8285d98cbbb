"""The touch-list sort alone: 1.18 M (key, sample) pairs with the headline's key distribution (24-bit keys: 1.05 M Zipf item rows, 65 536
output rows, 65 536 user rows), timed with HIP events, checked against torch's stable sort.  A library variant is picked with
DRX_HOST_SANITIZER_LIB=drecpy_amd/csrc/build/libdrx_<name>.so (scripts/build_variant.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drecpy_amd import _lib                                      # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
g = torch.Generator(device=dev); g.manual_seed(0)
N, U, B, T = 1_000_000, 10_000_000, 65536, 1_180_000
w = 1.0 / torch.arange(1, N + 1, device=dev, dtype=torch.float64) ** 1.05
items = torch.multinomial((w / w.sum()).float(), T - 2 * B, replacement=True, generator=g)
keys = torch.cat([items, N + torch.randint(0, N, (B,), device=dev, generator=g), 2 * N + torch.randint(0, U, (B,), device=dev, generator=g)])
keys = keys[torch.randperm(T, device=dev, generator=g)].to(torch.int32).contiguous()
vals = torch.arange(T, device=dev, dtype=torch.int32)
ko, vo = torch.empty_like(keys), torch.empty_like(vals)
bits = 24
need = L.drx_sort_pairs_temp_bytes(T, bits)
tmp = torch.empty(need, dtype=torch.uint8, device=dev)
run = lambda: _lib.check(L.drx_sort_pairs(_lib.ptr(keys), _lib.ptr(ko), _lib.ptr(vals), _lib.ptr(vo), T, bits, _lib.ptr(tmp), need,
                                          _lib.stream_ptr(dev)), 'drx_sort_pairs')
for _ in range(5):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(100):
    run()
e1.record()
torch.cuda.synchronize()
sk, order = torch.sort(keys.long(), stable=True)
ok = bool(torch.equal(ko.long(), sk) and torch.equal(vo.long(), order))
print(os.environ.get('DRX_HOST_SANITIZER_LIB', 'default'), 'sort of', T, 'pairs,', bits, 'bits:', round(e0.elapsed_time(e1) * 10, 2), 'us', 'correct' if ok else 'WRONG')
