import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from bench_configs import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import DMF
ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
for B in (256, 4096):
    m = DMF(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=False)
    m.fit(ds, epochs=3, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5)
    res = []
    for n in (200, 600):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.fit(ds, epochs=n, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5)
        torch.cuda.synchronize(); res.append(time.perf_counter() - t0)
    print('B', B, 'steady ms/step', round((res[1] - res[0]) / 400 * 1e3, 4), 'incl setup (200)', round(res[0] / 200 * 1e3, 4))
