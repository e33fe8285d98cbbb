"""Host-side cost of a Caser step (one batch reused): cProfile of Caser._do_batch at the ml-1m shape, B = 4096."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch                                                     # noqa: E402
from measure_models import frame_of                              # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import Caser                         # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ds = InteractionDataset.read_df(frame_of('ml-1m'), verbose=False)
m = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, sort_column='timestamp', seed=10, verbose=False)
m.fit(ds, epochs=1, batch_size=B, learning_rate=1e-3, reg_rate=1e-6, neg_ratio=3)
batch = m._sample_batch(B)
for s in range(1, 20):
    m._do_batch(batch, step=s)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for s in range(20, 320):
    m._do_batch(batch, step=s)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
