"""cProfile of the set-up part of DMF / Caser fit() (second call, epochs=1) at ml-1m shape."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from measure_models import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import DMF, Caser
which = sys.argv[1] if len(sys.argv) > 1 else 'dmf'
ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
if which == 'dmf':
    m, kw = DMF(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=False), dict(batch_size=256, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5)
else:
    m, kw = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, sort_column='timestamp', seed=10, verbose=False), dict(batch_size=512, learning_rate=1e-3, reg_rate=1e-6, neg_ratio=3)
m.fit(ds, epochs=1, **kw)
pr = cProfile.Profile()
pr.enable()
m.fit(ds, epochs=1, **kw)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
