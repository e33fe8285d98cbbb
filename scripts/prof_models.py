"""Runs 100 DMF or Caser steps at ml-1m-shaped data (for `rocprofv3 --kernel-trace --stats -- python3 scripts/prof_models.py dmf|caser`)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch                                                     # noqa: E402
from measure_models import frame_of                              # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import DMF, Caser                    # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'dmf'
B_arg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
if which == 'dmf':
    m, B = DMF(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=False), (B_arg or 256)
    m.fit(ds, epochs=1, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5)
else:
    m, B = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, sort_column='timestamp', seed=10, verbose=False), (B_arg or 512)
    m.fit(ds, epochs=1, batch_size=B, learning_rate=1e-3, reg_rate=1e-6, neg_ratio=3)
batch = m._sample_batch(B)
for s in range(1, 4):
    m._do_batch(batch, step=s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(4, 104):
    m._do_batch(batch, step=s)
torch.cuda.synchronize()
print(which, 'B', B, 'ms/step (one batch reused, incl. host packing)', (time.perf_counter() - t0) * 10)
