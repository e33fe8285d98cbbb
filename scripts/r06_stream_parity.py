"""Which run-ahead stream is a good one?  K dummy streams are created first (and kept alive), then DMF.fit(device_sampler=True) at
B = 256 creates its side stream: steady ms per step by K and by the dummies' priority."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                     # noqa: E402
import bench_configs as bc                                       # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import DMF                           # noqa: E402

K, prio = int(sys.argv[1]), int(sys.argv[2])
if os.environ.get('DRX_STREAM_PROBE') == '0':                      # (this script's own switch: the package reads no environment variable)
    from drecpy_amd import engine
    engine.STREAM_PROBE = False
keep = [torch.cuda.Stream('cuda:0', priority=prio) for _ in range(K)]
if len(sys.argv) > 3 and sys.argv[3] == 'use':                  # the dummies do some work: a stream gets its hardware queue when first used
    for st in keep:
        with torch.cuda.stream(st):
            torch.zeros(1024, device='cuda:0').add_(1)
    torch.cuda.synchronize()
ds = InteractionDataset.read_df(bc.frame_of('ml-1m'), verbose=False)
B = 256
md = DMF(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=False, device='cuda:0')
md.fit(ds, epochs=3, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5, device_sampler=True)
_, steady, spread = bc._fit_steady(md, lambda n: md.fit(ds, epochs=n, batch_size=B, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5, device_sampler=True), 400)


def runs_beside(side):
    """does a tiny launch on `side` finish while the current stream is busy with a long one?"""
    main = torch.cuda.current_stream()
    x = torch.zeros(1024, device='cuda:0')
    torch.cuda.synchronize()
    done_main, done_side = torch.cuda.Event(), torch.cuda.Event()
    torch.cuda._sleep(4_000_000)                                  # ~2 ms of spinning on the current stream
    done_main.record(main)
    with torch.cuda.stream(side):
        x.add_(1)
        done_side.record(side)
    import time
    t0 = time.perf_counter()
    while not done_side.query() and not done_main.query() and time.perf_counter() - t0 < 1.0:
        pass
    ok = done_side.query() and not done_main.query()
    torch.cuda.synchronize()
    return ok




def ping_pong_us(side, n=200):
    """microseconds per round of: side waits for main's last event, launches, records; main waits for that, launches, records"""
    import time
    main = torch.cuda.current_stream()
    x, y = torch.zeros(1024, device='cuda:0'), torch.zeros(1024, device='cuda:0')
    ev_main = torch.cuda.Event()
    ev_main.record(main)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        side.wait_event(ev_main)
        with torch.cuda.stream(side):
            x.add_(1)
        ev_side = torch.cuda.Event()
        ev_side.record(side)
        main.wait_event(ev_side)
        y.add_(1)
        ev_main = torch.cuda.Event()
        ev_main.record(main)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e6, 1)


print('ping-pong us per round: side stream', [ping_pong_us(md._dev_side) for _ in range(2)], 'dummies', [ping_pong_us(st) for st in keep])
print('runs beside the training stream:', [runs_beside(md._dev_side) for _ in range(3)], 'dummies:', [runs_beside(st) for st in keep])
print('dummies', K, 'priority', prio, 'steady ms/step', round(steady * 1e3, 4), 'side stream', hex(md._dev_side.cuda_stream), flush=True)
