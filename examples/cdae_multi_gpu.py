"""Row-sharded CDAE across the GPUs of a node: users (V rows, histories, their triples) stay on their GPU, item rows are
sharded by item range and travel by all-to-all over xGMI (drecpy_amd/dist.py).  One process per GPU:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/cdae_multi_gpu.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from drecpy_amd import synth
from drecpy_amd.dist import ShardedCdae, ShardedPipeline

rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
dist.init_process_group('nccl', device_id=dev)

U, N, mean_deg, min_deg, alpha = synth.SHAPES['synth-10m']
K, B, STEPS, Q = 128, 65536, 200, 0.2
lo, hi = U * rank // world, U * (rank + 1) // world
indptr, indices = synth.synth_history(U, N, mean_deg, min_deg, alpha, seed=0, device=dev, user_lo=lo, user_hi=hi)   # this rank's users
model = ShardedCdae(U, N, K, rank, world, dev, indptr, indices, seed=10, lr=0.05, reg=1e-3, q=Q)
eng = model.engine

from drecpy_amd.engine import DeviceBatchSource     # this rank's B triples of step s, drawn on the device two requests ahead
batch_of = DeviceBatchSource(eng, B, 5, Q, lambda s: 1000 + 7919 * s + rank, lambda s: 31 * s + rank, n_items=N)
pipe = ShardedPipeline(model, batch_of, STEPS)
for s in range(10):
    pipe.run_step()
torch.cuda.synchronize(); dist.barrier(); t0 = time.time()
for s in range(10, STEPS):
    loss = pipe.run_step(want_loss=(s == STEPS - 1))
torch.cuda.synchronize(); dist.barrier()
if rank == 0:
    dt = time.time() - t0
    print(f'{world} GPUs: {world * B * (STEPS - 10) / dt / 1e6:.1f} M triples/s, final loss {loss:.4f}')
dist.destroy_process_group()
