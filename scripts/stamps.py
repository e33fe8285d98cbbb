"""Phase stamps of the sparse step's two big kernels (diagnostic build: bash scripts/build_variant.sh stamps "-DDRX_STAMPS").
    DRX_HOST_SANITIZER_LIB=$PWD/drecpy_amd/csrc/build/libdrx_stamps.so python scripts/stamps.py
Runs a few steps of the bench workload, then one stamped step; prints mean / median / p90 of every phase in microseconds (the device's
100 MHz constant clock: 10 ns resolution) and each kernel's own span (first start to last end)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drecpy_amd import _lib, synth                     # noqa: E402
from drecpy_amd.engine import CdaeEngine, SampledPipeline      # noqa: E402


def main():
    dev = torch.device('cuda:0')
    U, N, md, mn, a = synth.SHAPES[os.environ.get('SHAPE', 'synth-10m')]          # SHAPE=ml-1m: the shared-form step (only the reduction's stamps apply)
    U = int(os.environ.get('USERS', U))
    B = 65536
    indptr, indices = synth.synth_history(U, N, md, mn, a, seed=0, device=dev)
    eng = CdaeEngine(U, N, 128, device=dev)
    eng.init_glorot_device(10); eng.set_history(indptr, indices); eng.init_optimizer('adagrad', 0.05, 1e-3)
    pipe = SampledPipeline(eng, B, 5, 0.2, lambda s: 5000 + 7919 * s, lambda s: 5000 + 7919 * s, n_items=N)
    for i in range(30):
        pipe.run_step()
        if os.environ.get('STAMPS_DEBUG'):
            torch.cuda.synchronize()
            print('step', i, 'ok', file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    units = 65536 + 45000
    buf = torch.zeros(units * 16, dtype=torch.int64, device=dev)
    L = C.CDLL(os.environ['DRX_HOST_SANITIZER_LIB'])
    L.drx_debug_set_stamps.argtypes = [C.c_void_p, C.c_uint]
    assert L.drx_debug_set_stamps(buf.data_ptr(), units) == 0
    torch.cuda.synchronize()
    pipe.run_step()
    torch.cuda.synchronize()
    assert L.drx_debug_set_stamps(None, 0) == 0
    st = buf.cpu().numpy().reshape(units, 16)
    out = {}

    def report(name, rows, cols, labels):
        t = rows[:, cols].astype(np.float64)
        ok = (t > 0).all(axis=1)
        t = t[ok] * 0.01                                   # 100 MHz ticks -> us
        d = np.diff(t, axis=1)
        res = {'units': int(ok.sum()), 'kernel_span_us': float(t.max() - t.min()), 'unit_life_us': {'mean': float((t[:, -1] - t[:, 0]).mean()), 'p50': float(np.median(t[:, -1] - t[:, 0])), 'p90': float(np.percentile(t[:, -1] - t[:, 0], 90))}}
        for i, lab in enumerate(labels):
            res[lab] = {'mean': round(float(d[:, i].mean()), 2), 'p50': round(float(np.median(d[:, i])), 2), 'p90': round(float(np.percentile(d[:, i], 90)), 2)}
        # how many units are alive at once (chip-wide), sampled at 200 points of the span
        grid = np.linspace(t.min(), t.max(), 200)
        alive = [(int(((t[:, 0] <= x) & (t[:, -1] > x)).sum())) for x in grid]
        res['alive_units_mean'] = float(np.mean(alive)); res['alive_units_max'] = int(np.max(alive))
        out[name] = res
    if os.environ.get('SHAPE', 'synth-10m') == 'synth-10m':
      report('k_sampled_fwd_bwd_pf (per triple)', st[:B], [0, 1, 2, 3, 4, 5, 6, 7, 8],
             ['uid/iid -> LDS-DMA issued', 'indptr', 'indices + mask', 'first rows', 'rest of the gather', 'b, b2, y + DMA landed', 'loss, dz1 stored', 'sole-toucher updates'])
    # (the reduction's kernels carry no stamps since r05: the planned kernel's were taken out of the hot loop, the streamed one never had
    # any — its phases were sized with library variants side by side, scripts/ab_headline.sh)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
