"""Where the wall time of reference-mode fit(5000) goes, run by run: waiting for the drawn batch vs issuing the step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from measure_models import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import CDAE
ds = InteractionDataset.read_df(frame_of('ml-100k'), verbose=False)
m = CDAE(hidden_factors=50, corruption_level=0.2, seed=10, verbose=False)
acc = {'sample': 0.0, 'do': 0.0}
os_, od_ = m._sample_batch, m._do_batch
def sb(*a, **k):
    t = time.perf_counter(); r = os_(*a, **k); acc['sample'] += time.perf_counter() - t; return r
def db(*a, **k):
    t = time.perf_counter(); r = od_(*a, **k); acc['do'] += time.perf_counter() - t; return r
m._sample_batch, m._do_batch = sb, db
m.fit(ds, epochs=10, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
for rep in range(8):
    acc['sample'] = acc['do'] = 0.0
    t0 = time.perf_counter()
    m.fit(ds, epochs=5000, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'fit {t1 - t0:.3f} s (+{t2 - t1:.3f} sync) | in _sample_batch {acc["sample"]:.3f} | in _do_batch {acc["do"]:.3f}')
