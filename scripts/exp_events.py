"""What do stream events cost the sparse step?  All batches and their touch lists are prepared up front (nothing runs beside the
training stream); variants add event records / a side stream doing trivial work."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                        # noqa: E402
from drecpy_amd import _lib, synth                  # noqa: E402
from drecpy_amd.engine import CdaeEngine            # noqa: E402

dev = torch.device('cuda:0')
U, N, md, mn, a = synth.SHAPES['synth-10m']
U = int(os.environ.get('USERS', U))
ip, idx = synth.synth_history(U, N, md, mn, a, seed=0, device=dev)
eng = CdaeEngine(U, N, 128, device=dev)
eng.init_glorot_device(10)
eng.set_history(ip, idx)
eng.init_optimizer('adagrad', 0.05, 1e-3)
B, NB = 65536, 8
bts, preps = [], []
for i in range(NB):
    uid, iid, y, ko = eng.sample_device(B, 5, 1000 + i, n_items=N)
    torch.cuda.synchronize()
    bt, alive = eng.make_batch(uid.clone(), iid.clone(), y.clone(), keep_off=ko.clone(), q=0.2, mask_seed=1000 + i, n_touch_slots=int(ko[-1].item()))
    bts.append((bt, alive))
    preps.append(eng.prepare_sparse(bt).clone())
torch.cuda.synchronize()
L = _lib.lib()
main = torch.cuda.current_stream()
side = torch.cuda.Stream(priority=-1)
tev = [torch.cuda.Event() for _ in range(4)]
rev = [L.drx_event_create() for _ in range(4)]
small = torch.zeros(1024, device=dev)


def loop(variant, steps=200):
    for s in range(steps):
        eng.step_sparse(s, bts[s % NB][0], 'bce', prepared=preps[s % NB])
        if variant == 'torch_events':
            tev[0].record(main); tev[1].record(main)
        elif variant == 'light_events':
            L.drx_event_record(rev[0], C.c_void_p(main.cuda_stream)); L.drx_event_record(rev[1], C.c_void_p(main.cuda_stream))
        elif variant == 'side_trivial_torch':
            tev[0].record(main)
            side.wait_event(tev[0])
            with torch.cuda.stream(side):
                small.add_(1.0)
                tev[1].record(side)
            main.wait_event(tev[1])
        elif variant == 'side_trivial_light':
            L.drx_event_record(rev[0], C.c_void_p(main.cuda_stream))
            L.drx_stream_wait_event(C.c_void_p(side.cuda_stream), rev[0])
            with torch.cuda.stream(side):
                small.add_(1.0)
            L.drx_event_record(rev[1], C.c_void_p(side.cuda_stream))
            L.drx_stream_wait_event(C.c_void_p(main.cuda_stream), rev[1])
        elif variant == 'side_trivial_nosync':
            with torch.cuda.stream(side):
                small.add_(1.0)
        elif variant == 'side_prepare_nosync':          # the whole touch-list preparation beside the step, result unused
            with torch.cuda.stream(side):
                side_bufs[s % 2] = eng.prepare_sparse(bts[(s + 1) % NB][0], side_bufs[s % 2])
        elif variant in ('side_sort_nosync', 'lowprio_sort_nosync'):
            with torch.cuda.stream(side if variant == 'side_sort_nosync' else side0):
                _lib.check(L.drx_sort_pairs(_lib.ptr(skeys), _lib.ptr(sko), _lib.ptr(svals), _lib.ptr(svo), sn, 24, _lib.ptr(stmp), stmp.numel(),
                                            _lib.stream_ptr(dev)), 'sort')
        elif variant == 'lowprio_prepare_nosync':
            with torch.cuda.stream(side0):
                side_bufs[s % 2] = eng.prepare_sparse(bts[(s + 1) % NB][0], side_bufs[s % 2])
        elif variant == 'side_sample_nosync':
            with torch.cuda.stream(side):
                eng.sample_device(B, 5, 77 + s, n_items=N, out=ring)
        elif variant == 'side_both_nosync':
            with torch.cuda.stream(side):
                eng.sample_device(B, 5, 77 + s, n_items=N, out=ring)
                side_bufs[s % 2] = eng.prepare_sparse(bts[(s + 1) % NB][0], side_bufs[s % 2])


side_bufs = [None, None]
side0 = torch.cuda.Stream(priority=0)
sn = 1_440_000
skeys = (torch.rand(sn, device=dev, dtype=torch.float64) ** 3 * ((1 << 24) - 1)).to(torch.int64).to(torch.int32)
svals = torch.arange(sn, device=dev, dtype=torch.int32)
sko, svo = torch.empty_like(skeys), torch.empty_like(svals)
stmp = torch.empty(L.drx_sort_pairs_temp_bytes(sn, 24), dtype=torch.uint8, device=dev)
ring = eng.sample_device(B, 5, 1, n_items=N)
variants = sys.argv[1:] or ['none', 'torch_events', 'light_events', 'side_trivial_torch', 'side_trivial_light', 'side_trivial_nosync', 'none']
for variant in variants:
    NS = int(os.environ.get('STEPS', 300))
    loop(variant, 30 if NS >= 300 else 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(variant, NS)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / NS
    print(f'{variant:22s} {dt * 1e3:.4f} ms/step  {B / dt / 1e6:.1f} M triples/s', flush=True)
