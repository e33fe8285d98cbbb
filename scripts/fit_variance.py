"""Repeated timings of reference-mode fit() (ml-100k shape) to see run-to-run variance of the host pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from measure_models import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import CDAE
shape, K = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ('ml-100k', 50)
if len(sys.argv) > 3:
    sys.setswitchinterval(float(sys.argv[3]))
ds = InteractionDataset.read_df(frame_of(shape), verbose=False)
m = CDAE(hidden_factors=K, corruption_level=0.2, seed=10, verbose=False)
def fit_s(n):
    t0 = time.perf_counter()
    m.fit(ds, epochs=n, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
fit_s(10)
print(shape, 'switchinterval', sys.getswitchinterval(), 'fit(500):', ' '.join(f'{fit_s(500):.3f}' for _ in range(3)), '| fit(5000):', ' '.join(f'{fit_s(5000):.3f}' for _ in range(6)))
