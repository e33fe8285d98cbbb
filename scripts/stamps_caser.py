"""Phase stamps of k_caser (diagnostic build: bash scripts/build_variant.sh stamps "-DDRX_STAMPS").
    DRX_HOST_SANITIZER_LIB=$PWD/drecpy_amd/csrc/build/libdrx_stamps.so python scripts/stamps_caser.py
One Caser step at the ml-1m shape, B = 4096 (BASELINE configuration 5); prints each phase's mean / p50 / p90 per workgroup (its first tile of 16 samples) in
microseconds of the device's 100 MHz clock."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_configs import frame_of                     # noqa: E402


def main():
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Recommender import Caser
    B = int(os.environ.get('B', 4096))
    ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
    m = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, seed=10, verbose=False, device='cuda:0')
    m.fit(ds, epochs=2, batch_size=B, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3)
    batch = m._sample_batch(B)
    for s in range(2, 8):
        m._do_batch(batch, step=s)
    torch.cuda.synchronize()
    buf = torch.zeros(B * 16, dtype=torch.int64, device='cuda:0')
    L = C.CDLL(os.environ['DRX_HOST_SANITIZER_LIB'])
    L.drx_debug_set_caser_stamps.argtypes = [C.c_void_p]
    assert L.drx_debug_set_caser_stamps(buf.data_ptr()) == 0
    m._do_batch(batch, step=8)
    torch.cuda.synchronize()
    assert L.drx_debug_set_caser_stamps(None) == 0
    G = (B + 15) // 16                                  # one row of stamps per tile
    grid = int(os.environ.get('GRID', 256))
    raw = buf.cpu().numpy().reshape(B, 16)[:G].astype(np.float64) * 0.01
    st = raw[:, :11]
    cols = [0, 1, 2, 3, 5, 4, 6, 7, 8, 9, 10]      # (stamp 4 is taken inside step 4, after stamp 5)
    st = st[:, cols]
    ok = (st > 0).all(axis=1)
    st = st[ok]
    labels = ['0a rows + target rows issued, weights -> LDS (first tile)', '0b rows into LDS', '1 convs forward + max / act / dropout', '3 dense_0',
              '4a targets, first sample of wave 0', '4b targets, second sample + barrier', '5 dense_0 backward + act_h backward, scatter',
              '7 dE rows (wave 0)', '8 small-weight gradients (wave 0)', '8b biases + barrier']
    later = np.arange(G)[ok] >= grid                  # tiles a workgroup takes after its first one
    if later.any():
        dl = np.diff(st[later], axis=1)
        print('later tiles:', {lab: round(float(dl[:, i].mean()), 2) for i, lab in enumerate(labels)}, 'life', round(float((st[later][:, -1] - st[later][:, 0]).mean()), 2))
        st = st[~later]
    d = np.diff(st, axis=1)
    out = {'workgroups': int(ok.sum()), 'kernel_span_us': float(st.max() - st.min()),
           'first_tile_life_us': {'mean': float((st[:, -1] - st[:, 0]).mean()), 'p90': float(np.percentile(st[:, -1] - st[:, 0], 90))},
           'first_start_spread_us': float(st[:, 0].max() - st[:, 0].min())}
    for i, lab in enumerate(labels):
        out[lab] = {'mean': round(float(d[:, i].mean()), 2), 'p50': round(float(np.median(d[:, i])), 2), 'p90': round(float(np.percentile(d[:, i], 90)), 2)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
