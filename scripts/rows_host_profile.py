"""Host-side cost of one row-sharded step (world 1 through a 1-rank RCCL communicator): cProfile of ShardedPipeline.run_step.
    DRX_BENCH_RCCL1=1 python scripts/rows_host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import tempfile

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drecpy_amd import synth                                   # noqa: E402
from drecpy_amd.dist import ShardedCdae, ShardedPipeline       # noqa: E402
from drecpy_amd.engine import DeviceBatchSource                # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
chunks = int(os.environ.get('CHUNKS', 4))
bypass = os.environ.get('BYPASS', '0') == '1'
transport = os.environ.get('TRANSPORT', 'rccl')
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method=f"file://{tempfile.mkdtemp()}/r", rank=0, world_size=1, device_id=dev)
U, N, K, B = 2_000_000, 1_000_000, 128, 65536
indptr, indices = synth.synth_history(U, N, 23.4, 5, 1.05, seed=0, device=dev)
m = ShardedCdae(U, N, K, 0, 1, dev, indptr, indices, force_collectives=True, chunks=chunks, self_bypass=bypass,
                transport='rccl' if transport == 'rccl' else None)
src = DeviceBatchSource(m.engine, B, 5, 0.2, lambda s: 5000 + s, lambda s: 5000 + s, n_items=N)
pipe = ShardedPipeline(m, src, steps + 20)
for _ in range(20):
    pipe.run_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    pipe.run_step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(60)
st.sort_stats('tottime').print_stats(25)
