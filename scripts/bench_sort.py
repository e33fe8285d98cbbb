"""Times drx_sort_pairs alone at the sizes of the sparse step's touch list (python scripts/bench_sort.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                              # noqa: E402
from drecpy_amd import _lib               # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
for n, bits in ((1_440_000, 24), (1_440_000, 20), (1_180_000, 20), (131_072, 24), (10_400_000, 24), (50_000, 22)):
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    hi = (1 << bits) - 1
    keys = ((torch.rand(n, generator=g, device='cuda', dtype=torch.float64) ** 3) * hi).to(torch.int64).to(torch.int32)
    vals = torch.arange(n, device='cuda', dtype=torch.int32)
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    need = L.drx_sort_pairs_temp_bytes(n, bits)
    tmp = torch.empty(need, dtype=torch.uint8, device='cuda')
    st = _lib.stream_ptr(dev)

    def run():
        _lib.check(L.drx_sort_pairs(_lib.ptr(keys), _lib.ptr(ko), _lib.ptr(vals), _lib.ptr(vo), n, bits, _lib.ptr(tmp), need, st), 'sort')
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    want = torch.sort(keys.long() & hi, stable=True)
    ok = bool(torch.equal(vo.long(), want.indices))
    print(f'n={n} bits={bits}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per sort, correct={ok}', flush=True)
