"""Is a slow fit(5000) slow on the host or on the device?  Alternates fits with a tight device-bound loop of the same step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from measure_models import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import CDAE
ds = InteractionDataset.read_df(frame_of('ml-100k'), verbose=False)
m = CDAE(hidden_factors=50, corruption_level=0.2, seed=10, verbose=False)
m.fit(ds, epochs=10, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
print('affinity size', len(os.sched_getaffinity(0)), 'threads', torch.get_num_threads())
for rep in range(6):
    t0 = time.perf_counter()
    m.fit(ds, epochs=5000, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
    torch.cuda.synchronize()
    f = time.perf_counter() - t0
    batch = m._sample_batch(64)
    bt = m._engine.batch_in_slot(batch.slot, 64, int(batch.keep_off[-1]), 0.2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(3000):
        m._engine.step_dense(s, bt)
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    d = time.perf_counter() - t0
    t0 = time.perf_counter()
    x = 0
    for i in range(200000):
        x += i
    py = time.perf_counter() - t0
    print(f'fit(5000) {f:.3f} s | 3000 steps back to back: issue {issue / 3000 * 1e6:.1f} us/step, done {d / 3000 * 1e6:.1f} us/step | python loop {py * 1e3:.1f} ms')
