"""Host time to ISSUE a step of the sampled pipeline against the device time of the step (is the host far enough ahead?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drecpy_amd import synth
from drecpy_amd.engine import CdaeEngine, SampledPipeline
dev = torch.device('cuda:0')
U, N, md, mn, a = synth.SHAPES['synth-10m']
indptr, indices = synth.synth_history(U, N, md, mn, a, seed=0, device=dev)
eng = CdaeEngine(U, N, 128, device=dev)
eng.init_glorot_device(10); eng.set_history(indptr, indices); eng.init_optimizer('adagrad', 0.05, 1e-3)
pipe = SampledPipeline(eng, 65536, 5, 0.2, lambda s: 5000 + 7919 * s, lambda s: 5000 + 7919 * s, n_items=N)
for _ in range(30): pipe.run_step()
torch.cuda.synchronize()
for n in (20, 200):
    t0 = time.perf_counter()
    for _ in range(n): pipe.run_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{n} steps: host issue {(t1 - t0) / n * 1e3:.4f} ms/step, until done {(t2 - t0) / n * 1e3:.4f} ms/step')
import cProfile, pstats
cProfile.run('for _ in range(200): pipe.run_step()', '/tmp/hi.prof')
torch.cuda.synchronize()
pstats.Stats('/tmp/hi.prof').sort_stats('tottime').print_stats(12)
