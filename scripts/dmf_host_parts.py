"""Host cost of one DMF batch at the ml-1m shape: the sampler draw, the batch preparation, the step's issue."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from measure_models import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import DMF
ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
for B in (256, 4096):
    m = DMF(seed=10, verbose=False)
    m.fit(ds, epochs=2, batch_size=B, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
    def t(fn, n=200):
        fn(); t0 = time.perf_counter()
        for _ in range(n): fn()
        return (time.perf_counter() - t0) / n * 1e3
    draw = t(lambda: m._sampler.sample_arrays(B))
    u, i, v, _ = m._sampler.sample_arrays(B)
    std = t(lambda: m._standardize_value(v) if m.use_nce else v)
    y = m._standardize_value(v) if m.use_nce else v
    prep = t(lambda: m._engine.prepare_batch(u, i, y))
    whole = t(lambda: m._sample_batch(B))
    batch = m._sample_batch(B)
    torch.cuda.synchronize()
    issue = t(lambda: m._do_batch(batch, step=3))
    torch.cuda.synchronize()
    print(f'B={B}: draw {draw:.3f} ms, standardise {std:.3f}, prepare_batch {prep:.3f}, _sample_batch {whole:.3f}, _do_batch (issue, device-bound if larger than the step) {issue:.3f}')
