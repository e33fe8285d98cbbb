"""cProfile of DMF.fit() / Caser.fit() at the ml-1m shape: where the host time of a step goes.  python scripts/fit_profile_models.py dmf 4096"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch                                                     # noqa: E402
from measure_models import frame_of                              # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import DMF, Caser                    # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'dmf'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
if which == 'dmf':
    m = DMF(seed=10, verbose=False)
    kw = dict(learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
else:
    m = Caser(seed=10, verbose=False, dropout_rate=0.5)
    kw = dict(learning_rate=5e-3, reg_rate=1e-6, neg_ratio=3)
m.fit(ds, epochs=3, batch_size=B, **kw)
for rep in range(2):
    t0 = time.perf_counter()
    m.fit(ds, epochs=n, batch_size=B, **kw)
    torch.cuda.synchronize()
    print(f'{which} B={B}: fit({n}) {time.perf_counter() - t0:.3f} s = {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step')
pr = cProfile.Profile()
pr.enable()
m.fit(ds, epochs=n, batch_size=B, **kw)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(32)
