"""Where a Caser.fit(device_sampler=True) step goes: the draw, the step on a device batch (wall and device time), the host's share."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_configs import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import Caser
B = 4096
ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
m = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, seed=10, verbose=False)
m.fit(ds, epochs=3, batch_size=B, learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3, device_sampler=True)
torch.cuda.synchronize()
def timed(fn, n=100):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    issue = (time.perf_counter() - t0) / n
    e1.record(); torch.cuda.synchronize()
    return {'host_issue_ms': issue * 1e3, 'device_ms': e0.elapsed_time(e1) / n, 'wall_ms': (time.perf_counter() - t0) / n * 1e3}
print('draw', timed(lambda: m._sample_batch(B)))
batch = m._sample_batch(B)
st = {'s': 5}
def step():
    m._do_batch(batch, step=st['s']); st['s'] += 1
print('step on a device batch', timed(step))
def both():
    b = m._sample_batch(B); m._do_batch(b, step=st['s']); st['s'] += 1
print('draw + step', timed(both))
import cProfile, pstats
cProfile.run('for _ in range(50): both()', '/tmp/cf.prof')
pstats.Stats('/tmp/cf.prof').sort_stats('tottime').print_stats(14)
