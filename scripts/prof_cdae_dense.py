"""300 reference-mode CDAE steps (dense Keras Adam) at ml-1m-shaped data, K = 128, B = 64, one batch reused — for
`rocprofv3 --kernel-trace --stats --output-format csv -- python3 scripts/prof_cdae_dense.py`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch                                                     # noqa: E402
from measure_models import frame_of                              # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import CDAE                          # noqa: E402

shape, K, B = (sys.argv[1] if len(sys.argv) > 1 else 'ml-1m'), int(sys.argv[2]) if len(sys.argv) > 2 else 128, 64
ds = InteractionDataset.read_df(frame_of(shape), verbose=False)
m = CDAE(hidden_factors=K, corruption_level=0.2, seed=10, verbose=False)
m.fit(ds, epochs=1, batch_size=B, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
batch = m._sample_batch(B)
uid, _, _ = m._batch_arrays(batch)
keep_off, keep = m._corruption_keep(uid)
bt, alive = m._engine.make_batch(uid, keep_off=keep_off, keep=keep, q=0.2, n_touch_slots=int(keep_off[-1]))
for s in range(5):
    m._engine.step_dense(s, bt)
torch.cuda.synchronize()
n = int(os.environ.get('DRX_PROF_STEPS', 300))
for rep in range(int(os.environ.get('DRX_PROF_REPS', 1))):
    t0 = time.perf_counter()
    for s in range(5, 5 + n):
        m._engine.step_dense(s, bt)
    torch.cuda.synchronize()
    print(shape, K, 'device ms/step', (time.perf_counter() - t0) / n * 1e3)
