"""Can two ranks of the library's own RCCL communicator share ONE GPU?  (RCCL normally refuses duplicate devices.)  If they can, the
world-2 row-sharded step runs through real send / recv pairs on a one-GPU box."""
import ctypes as C
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                     # noqa: E402
import torch.distributed as dist                                 # noqa: E402
import torch.multiprocessing as mp                               # noqa: E402


def worker(rank, world, rdzv):
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', init_method=rdzv, rank=rank, world_size=world)
    from drecpy_amd import _lib
    L = _lib.lib()
    ident = C.create_string_buffer(128)
    if rank == 0:
        print('unique id rc', L.drx_comm_unique_id(ident), flush=True)
    box = [ident.raw if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ident = C.create_string_buffer(box[0], 128)
    h = C.c_void_p()
    rc = L.drx_comm_create(ident, world, rank, 1, C.byref(h))
    print('rank', rank, 'drx_comm_create rc', rc, L.drx_comm_last_error().decode() if rc else '', flush=True)
    if rc == 0:
        n = 1 << 20
        send = torch.full((world * n,), float(rank + 1), device='cuda:0')
        recv = torch.zeros(world * n, device='cuda:0')
        so = (C.c_int64 * world)(*[p * n * 4 for p in range(world)])
        sb = (C.c_int64 * world)(*[n * 4] * world)
        t = L.drx_comm_alltoallv(h, send.data_ptr(), so, sb, recv.data_ptr(), so, sb, _lib.stream_ptr(torch.device('cuda:0')))
        print('rank', rank, 'ticket', t, flush=True)
        if t >= 0:
            L.drx_comm_wait(h, t, _lib.stream_ptr(torch.device('cuda:0')))
            torch.cuda.synchronize()
            print('rank', rank, 'received', [float(recv[p * n]) for p in range(world)], flush=True)
        L.drx_comm_destroy(h)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    d = tempfile.mkdtemp()
    mp.spawn(worker, args=(2, f'file://{d}/rdzv'), nprocs=2, join=True)
