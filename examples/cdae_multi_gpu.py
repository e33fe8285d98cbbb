"""Multi-GPU CDAE (sampled sparse-Adagrad mode), one process per GPU, column-sharded layout (drecpy_amd/dist.py,
ColumnShardedCdae): every rank holds all rows but K/N columns of every table and trains on the same global batch; the only
exchange of a step is the all-reduce of the per-triple partial dot products.  (`ShardedCdae` + `ShardedPipeline` is the
row-sharded alternative: users and item rows sharded by range, rows travel by all-to-all.)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/cdae_multi_gpu.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from drecpy_amd import synth
from drecpy_amd.dist import ColumnShardedCdae

rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
dist.init_process_group('nccl', device_id=dev)

U, N, mean_deg, min_deg, alpha = synth.SHAPES['synth-10m']
K, B, STEPS, Q = 128, 65536, 200, 0.2
indptr, indices = synth.synth_history(U, N, mean_deg, min_deg, alpha, seed=0, device=dev)       # every rank: the histories of ALL users
model = ColumnShardedCdae(U, N, K, rank, world, dev, indptr, indices, seed=10, lr=0.05, reg=1e-3, q=Q,
                          force_collectives=True)      # (world 1 too: exercises the RCCL all-reduce)
pipe = model.pipeline(B * world, 5, lambda s: 1000 + 7919 * s, lambda s: 31 * s)                 # the same seeds on every rank
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
