"""DMF.fit() on the reference-exact host sampler at B = 256 (ml-1m shape): steady ms per step by sys.setswitchinterval (the GIL hand-over
between the sampler's worker thread and the issuing thread)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                     # noqa: E402
import bench_configs as bc                                       # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import DMF                           # noqa: E402

ds = InteractionDataset.read_df(bc.frame_of('ml-1m'), verbose=False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DMF._prefetch_mode = os.environ.get('DRX_HOST_PREFETCH', 'spin')     # (this script's own switch: the package reads no environment variable)
from drecpy_amd.engine_dmf import DmfEngine                      # noqa: E402
DmfEngine.host_step_cache = os.environ.get('DRX_HOST_CACHE', '1') == '1'      # (A/B: the cached argument structs of host-batch steps)
if os.environ.get('DRX_FORCE_WORKER') == '1':
    DMF._prefetch_from = 0                                       # (A/B: the worker thread at every batch size)
for si in (5e-3, 5e-3, 5e-3):
    sys.setswitchinterval(si)
    md = DMF(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=False, device='cuda:0')
    md.fit(ds, epochs=3, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5)
    _, steady, spread = bc._fit_steady(md, lambda n: md.fit(ds, epochs=n, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5), 400)
    print('switchinterval', si, 'B', B, 'steady ms/step', round(steady * 1e3, 4), {k: round(v, 4) if isinstance(v, float) else v for k, v in spread.items()}, flush=True)
