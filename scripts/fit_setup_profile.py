"""cProfile of the set-up part of a reference-mode CDAE.fit() (second call, epochs=1) at ml-100k shape."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from measure_models import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import CDAE
shape, K = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ('ml-100k', 50)
ds = InteractionDataset.read_df(frame_of(shape), verbose=False)
m = CDAE(hidden_factors=K, corruption_level=0.2, seed=10, verbose=False)
m.fit(ds, epochs=1, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
pr = cProfile.Profile()
pr.enable()
m.fit(ds, epochs=1, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
