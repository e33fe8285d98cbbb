"""bench.py — training samples/s of the CDAE hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload synth-10m|ml-1m|ml-100k] [--batch B]

A "step" is one pass of the hot path over one batch of B (u, i, y) triples per GPU: gather + hidden layer + sampled output unit + BCE +
backward + sparse-Adagrad update (drx_cdae_step_sparse_prepared).  Every step trains on a FRESH batch drawn by the device PointSampler
(drx_point_sample) on a side stream, and the batch's sorted touch list (drx_cdae_sparse_prepare) is built ahead on the same side stream —
both depend only on the data, never on the parameters — so the timed region is the whole training loop including sampling.  All inputs
live in HBM; nothing crosses PCIe in the timed region except the 8-byte touch count the sampler posts to a pinned mailbox per step.

Timing: W warm-up steps, then `--windows` (default 5) back-to-back windows of EXACTLY K steps, each bracketed by barrier +
synchronize on both sides and reduced by MAX over ranks; `value` and `ms_per_step` come from the MEDIAN window, `window_ms_min/max`
give the spread (a 20-step window is 8 ms: one shot cannot tell 5 % from noise).

N GPUs (one process per GPU, RCCL over xGMI).  `python bench.py --gpus N` starts the N ranks itself when WORLD_SIZE is not set (no
GPU call is made by the starting process); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` every worker
is its own rank's coordinator.  Either way each sharding LAYOUT runs in a fresh child process per rank (a fault in one layout cannot
take the other's number with it): `rows` — BASELINE.json's partitioning: users (V rows, histories, samples) and item rows sharded by
range, rows and gradient rows travel by all-to-all(v), bias by all-reduce — and `columns` — every rank all rows x K/N columns of every
table on the same global batch, one all-reduce of B floats per step.  The line's headline is the faster layout; `layouts` carries
both, `config.sharding` says which is which, `rccl_ranks` is the sum of an RCCL all-reduce of ones.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the definitions of roofline / cpu_baseline / hr_at_10 / configs).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
L2_GATHER_GBS = 17000.0      # indexed rows shared by every workgroup, served by the XCDs' L2 (guide: 16.8-18.8 TB/s chip-wide)
MALL_GATHER_GBS = 8600.0     # uniformly random rows of a 38 MB table, served by the Infinity Cache (guide: 8.6 TB/s)
# What THIS chip delivered for the step's two access patterns free of any segment logic (scripts/mb/mb_rows.hip, mb_tlb.hip; logs:
# profiles/r05_mb_rows_*.log, r05_mb_footprint.log): 512-byte rows gathered from a 33.5 MB buffer 7.3 - 7.9 TB/s; read-modify-write of
# random 512-byte rows of tables far larger than the Infinity Cache, cold, 4.6 - 4.9 TB/s (whatever the table's size, 0.5 - 5 GB);
# both in ONE launch took the SUM of their times.  `measured_reference` prices a kernel's bytes at these rates: context, not a bound.
MB_GATHER_GBS = 7600.0
MB_RMW_COLD_GBS = 4800.0
K = 128
Q = 0.2
NEG_RATIO = 5
LR, REG = 0.05, 1e-3
LAYOUTS = ('rows', 'columns')


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--windows', type=int, default=5, help='back-to-back timed windows of --steps steps; value = median window')
    ap.add_argument('--workload', default='synth-10m', choices=['synth-10m', 'ml-1m', 'ml-100k'])
    ap.add_argument('--batch', type=int, default=65536, help='triples per GPU per step')
    ap.add_argument('--n-batches', type=int, default=8, help='batches sampled at setup (R estimate, CPU baseline; cycled with --presampled)')
    ap.add_argument('--presampled', action='store_true', help='cycle through the setup batches instead of sampling a fresh batch every step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-all-cores', action='store_true', help='skip the all-cores CPU baseline (one worker process per host core)')
    ap.add_argument('--cpu-budget-s', type=float, default=12.0)
    ap.add_argument('--cpu-triples', type=int, default=1024)
    ap.add_argument('--no-hr', action='store_true', help='skip the HR@10 sanity run (ml-100k-shaped set, reference mode)')
    ap.add_argument('--no-configs', action='store_true', help='skip the `configs` block (BASELINE configurations 2, 3, 5)')
    ap.add_argument('--force-configs', action='store_true', help='emit the `configs` block whatever the main workload is (tests)')
    ap.add_argument('--no-overlap', action='store_true', help='build the touch list inline instead of ahead on a side stream')
    ap.add_argument('--prep-cus', type=int, default=-1, help='CUs of every XCD the run-ahead streams (sampler, list preparation) are confined to '
                    '(drx_stream_create_cu_slice); 0: the whole chip, high priority; default: engine.PREP_CUS_PER_XCD')
    ap.add_argument('--side-streams', type=int, default=2, help='run-ahead streams (the work of step s on stream s %% n)')
    ap.add_argument('--prep-ahead', type=int, default=3, help='steps of lead of the list preparation')
    ap.add_argument('--no-share-users', action='store_true', help='A/B: plain lists where DRX_BATCH_SHARE_USERS would apply')
    ap.add_argument('--no-sample-by-user', action='store_true', help='A/B: device-sampled batches in draw order')
    ap.add_argument('--users', type=int, default=0, help='override the number of users (debug)')
    ap.add_argument('--force-sharded', action='store_true', help='run the row-sharded step even at 1 GPU (measures its overhead)')
    ap.add_argument('--optimizer', default='adagrad', choices=['adagrad', 'adam', 'rowwise_adagrad'], help='sparse optimizer of the sampled mode (S = 1 / 2 slots per parameter; single-GPU path)')
    ap.add_argument('--k', type=int, default=0, help='hidden factors (default 128); with --force-columns --k 128/N --batch 65536*N one GPU runs the '
                                                      'shape of ONE rank of an N-GPU column-sharded job')
    ap.add_argument('--layout', default='both', choices=['both', 'columns', 'rows'],
                    help='multi-GPU layout(s) to measure: rows = users and item rows sharded by range, rows and gradient rows travel by all-to-all '
                         '(BASELINE.json); columns = every rank all rows x K/N columns, same global batch, one all-reduce of B scalars per step')
    ap.add_argument('--force-columns', action='store_true', help='run the column-sharded code path even at 1 GPU')
    ap.add_argument('--prepare', default='auto', choices=['auto', 'local', 'turns', 'parts'],
                    help='column layout, who sorts the touch list of a step: every rank all of it (local), rank s %% N for all (turns), every rank '
                         '1/N of it (parts); auto = turns from 4 GPUs on (at 2 it saves nothing), else local')
    ap.add_argument('--no-self-bypass', action='store_true', help='row layout: send the rank\'s OWN rows through the collectives too (at world 1: the '
                    'whole exchange goes through the communicator — every row "remote", the link replaced by a device copy)')
    ap.add_argument('--chunks', type=int, default=2, help='row layout: exchange chunks per owner (a power of two; every exchange is that many all-to-alls, '
                    'pipelined with the owner apply and the next step\'s gather: drecpy_amd/dist.py); 1 = one all-to-all per exchange (r05)')
    ap.add_argument('--transport', default='rccl', choices=['rccl', 'torch'], help='row layout: who issues the all-to-all(v) exchanges — the library\'s own '
                    'RCCL communicator (csrc/drx_comm.hip: one ncclGroup per exchange, enqueued from C) or torch.distributed')
    ap.add_argument('--no-phases', action='store_true', help='row layout, --transport rccl: issue the exchanges of a step call by call from Python '
                    '(drecpy_amd/dist.py) instead of through the library\'s four phase calls (drx_shard_phase_*: A/B of the host cost)')
    ap.add_argument('--no-comm-thread', action='store_true', help='row layout at world 1 (DRX_BENCH_RCCL1): the library\'s communicator without its issuing thread')
    ap.add_argument('--micro', type=int, default=1, help='micro-batches per sharded step (exchanges of one overlap the compute of the other); default 1')
    ap.add_argument('--launch-dry-run', action='store_true', help='print the per-rank child command lines of an N-GPU run and exit (no GPU call)')
    ap.add_argument('--launch-selftest', action='store_true', help='children only rendezvous over gloo and all-reduce on the CPU (tests the launcher)')
    ap.add_argument('--layout-timeout-s', type=float, default=900.0, help='a layout whose children run longer is killed and reported as an error')
    ap.add_argument('--child-layout', default=None, choices=['columns', 'rows', 'selftest'], help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------------------
# N-GPU launch: no GPU call anywhere in this section
# ------------------------------------------------------------------------------------------------------------------------------
def _child_argv(argv, layout, override=()):
    """The command line of one rank's child for `layout`: this script with the same measurement flags (`override`: flag, value, …
    that replace the caller's)."""
    drop_with_value = {'--layout', '--child-layout', '--layout-timeout-s'} | set(override[0::2])
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in drop_with_value:
            skip = True
            continue
        if any(a.startswith(d + '=') for d in drop_with_value) or a in ('--launch-dry-run',):
            continue
        out.append(a)
    return [sys.executable, os.path.join(ROOT, 'bench.py')] + out + list(override) + ['--child-layout', layout]


def _agree(rdzv_dir, name, ranks, world, ok, wait_s=120.0):
    """Did attempt `name` succeed on EVERY rank?  Under torch.distributed.run each rank has its own coordinator: all of them must take
    the same decision (retry / go on), or the next layout's children would wait for ranks that never come.  One small file per rank in
    the rendezvous directory (one node: a shared file system); a rank that never reports counts as failed."""
    for r in ranks:
        with open(os.path.join(rdzv_dir, f'{name}.status.{r}.tmp'), 'w') as f:
            f.write('ok' if ok else 'bad')
        os.replace(os.path.join(rdzv_dir, f'{name}.status.{r}.tmp'), os.path.join(rdzv_dir, f'{name}.status.{r}'))
    t0, seen = time.time(), {}
    while len(seen) < world and time.time() - t0 < wait_s:
        for r in range(world):
            if r not in seen:
                try:
                    with open(os.path.join(rdzv_dir, f'{name}.status.{r}')) as f:
                        seen[r] = f.read().strip()
                except OSError:
                    pass
        if len(seen) < world:
            time.sleep(0.05)
    return len(seen) == world and all(v == 'ok' for v in seen.values())


def _child_env(rank, local_rank, world, rdzv):
    env = dict(os.environ)
    for k in list(env):
        if k.startswith('TORCHELASTIC_') or k in ('GROUP_RANK', 'ROLE_RANK', 'ROLE_NAME', 'ROLE_WORLD_SIZE', 'GROUP_WORLD_SIZE', 'MASTER_ADDR',
                                                  'MASTER_PORT'):
            env.pop(k)            # the children rendezvous among themselves through a FILE store (DRX_RDZV), not through an agent or a port
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(local_rank), 'WORLD_SIZE': str(world), 'LOCAL_WORLD_SIZE': str(world), 'DRX_RDZV': rdzv})
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: the only kind the host driver supports
    return env


def _supervise(procs, limit_s):
    """Waits for ALL children at once (waiting in rank order left rank 0 inside RCCL for minutes after another rank had
    died).  The first child that exits non-zero — or the time limit — ends the layout: the survivors (children this process started,
    by PID) are killed at once.  Returns ({rank: exit code | 'timeout' | 'killed'}, rank 0's stdout, the rank that failed first)."""
    import threading
    out0 = {}

    def drain(pr):                       # rank 0's pipe must be read while it runs, or a long line blocks it
        out0['b'] = pr.stdout.read()
    readers = []
    for r, pr in procs:
        if pr.stdout is not None:
            t = threading.Thread(target=drain, args=(pr,), daemon=True)
            t.start()
            readers.append(t)
    t0, rcs, first_bad = time.time(), {}, None
    live = dict(procs)
    while live:
        for r, pr in list(live.items()):
            rc = pr.poll()
            if rc is not None:
                rcs[r] = rc
                live.pop(r)
                if rc != 0 and first_bad is None:
                    first_bad = r
        timed_out = time.time() - t0 > limit_s
        if live and (first_bad is not None or timed_out):
            for r, pr in live.items():
                pr.kill()
                pr.wait()
                rcs[r] = 'timeout' if (timed_out and first_bad is None) else 'killed'
            live = {}
        if live:
            time.sleep(0.05)
    for t in readers:
        t.join(timeout=10.0)
    return rcs, out0.get('b') or b'', first_bad


def coordinate(args, argv, ranks, world, rdzv_dir, local_of=None):
    """Runs every layout as a set of fresh child processes — one per rank in `ranks` (all of them when this process started the job,
    just its own under torch.distributed.run) — and, where rank 0 is among them, returns the merged line.  A layout whose children
    fail or exceed the time limit is reported inside `layouts` and cannot be the headline.  The children of a layout meet through
    `file://<rdzv_dir>/<layout>`: no TCP port is chosen anywhere (a computed port inside the ephemeral range failed GPUTEST_r03)."""
    layouts = ['selftest'] if args.launch_selftest else (list(LAYOUTS) if args.layout == 'both' else [args.layout])
    # (name, layout, flags that replace the caller's, only if that attempt failed).  The row layout's second chance: the same step with
    # its exchanges issued through torch.distributed — the library's own communicator has only ever run at world 1 (one-GPU boxes)
    attempts = []
    for lay in layouts:
        attempts.append((lay, lay, (), None))
        if lay == 'rows' and args.transport == 'rccl':
            attempts.append(('rows_torch_transport', 'rows', ('--transport', 'torch'), 'rows'))
    results, errors, failed, done_ok = {}, {}, set(), set()
    local_of = local_of or {}
    for name, lay, override, only_after in attempts:
        if only_after is not None and only_after not in failed and not args.launch_dry_run:
            continue
        rdzv = f'file://{rdzv_dir}/{name}'
        cmd = _child_argv(argv, lay, override)
        if args.launch_dry_run:
            for r in ranks:
                print(f'[{name}{" (only if " + only_after + " fails)" if only_after else ""}] RANK={r} LOCAL_RANK={r} WORLD_SIZE={world} '
                      f'DRX_RDZV={rdzv} ' + ' '.join(cmd), flush=True)
            continue
        t0 = time.time()
        procs = [(r, subprocess.Popen(cmd, env=_child_env(r, local_of.get(r, r), world, rdzv),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None)) for r in ranks]
        rcs, out0, first_bad = _supervise(procs, args.layout_timeout_s)
        bad = {r: c for r, c in rcs.items() if c != 0}
        line = None
        for ln in out0.decode(errors='replace').splitlines():
            if ln.startswith('{'):
                line = ln
        if bad:
            errors[name] = {'error': f'child exit codes {bad}', 'first_failed_rank': first_bad, 'seconds': round(time.time() - t0, 1)}
        elif 0 in ranks and line is None:
            errors[name] = {'error': 'rank 0 printed no JSON line', 'seconds': round(time.time() - t0, 1)}
        if not _agree(rdzv_dir, name, ranks, world, name not in errors):
            failed.add(name)
            errors.setdefault(name, {'error': 'failed on another rank', 'seconds': round(time.time() - t0, 1)})
        else:
            done_ok.add(lay)
            if line is not None:
                results[lay] = json.loads(line)
                if name != lay:
                    results[lay]['attempt'] = name
    if args.launch_dry_run:
        return None, 0
    rc = 0 if (results or (0 not in ranks and not errors)) else 1
    if 0 not in ranks:
        return None, (0 if done_ok else 1)
    if not results:
        return {'metric': 'training samples/sec (user-item pairs)', 'value': None, 'n_gpus': world, 'layouts': errors, 'error': 'every layout failed'}, rc
    # north_star fixes the ROW-wise shard: it is the N > 1 headline whenever it ran (r06; through r05 the faster layout was); the column
    # layout — every rank walks the whole global batch — stays in `layouts` as an extra, and is the headline only if the row layout failed
    best = 'rows' if 'rows' in results else max(results, key=lambda l_: results[l_].get('value') or 0.0)
    out = dict(results[best])
    out['headline_layout'] = best
    out['headline_rule'] = 'rows (north_star) when it ran; otherwise the layout that did'
    brief = {}
    for lay, r in results.items():
        brief[lay] = {k_: r.get(k_) for k_ in ('value', 'ms_per_step', 'window_ms_min', 'window_ms_max', 'phases_ms', 'rccl_ranks', 'host_issue_ms_per_step', 'attempt')}
        brief[lay]['sharding'] = (r.get('config') or {}).get('sharding')
        brief[lay]['roofline'] = {k_: (r.get('roofline') or {}).get(k_) for k_ in ('kernel', 'frac', 'achieved', 'whole_step_frac', 'model_frac')}
    brief.update(errors)
    out['layouts'] = brief
    if 'config' in out and not args.launch_selftest:
        out['config'] = dict(out['config'])
        out['config']['sharding'] = {
            'headline': best,
            'rows': 'north_star / BASELINE.json configuration 4: V rows, histories and samples sharded by user range, W / W2T / b2 rows by item '
                    'range; all-to-all(v) of requested rows and of merged gradient rows, all-reduce of the hidden-bias gradient (dist.ShardedCdae)',
            'columns': 'every rank holds all rows x K/N columns of every table and trains on the same global batch; one all-reduce of B floats '
                       'per step (dist.ColumnShardedCdae)',
            'this_line': out['config'].get('sharding')}
    return out, rc


def _shared_rendezvous_dir(rank):
    """Under torch.distributed.run the ranks' coordinators are separate processes: rank 0 makes a fresh directory and publishes its
    name through the AGENT's store (MASTER_ADDR:MASTER_PORT, which torch.distributed.run already hosts — this process only connects as
    a client, it opens no port of its own); the others read it."""
    import tempfile
    from torch.distributed import TCPStore
    store = TCPStore(os.environ.get('MASTER_ADDR', '127.0.0.1'), int(os.environ['MASTER_PORT']), is_master=False)
    key = 'drx_bench/rendezvous_dir/' + os.environ.get('TORCHELASTIC_RUN_ID', 'none') + '/' + os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')
    if rank == 0:
        d = tempfile.mkdtemp(prefix='drx_rdzv_')
        store.set(key, d)
        return d
    return store.get(key).decode()


def launch_or_coordinate(args, argv):
    """--gpus N > 1 without --child-layout: start (or, under torch.distributed.run, be) the per-rank coordinators."""
    import tempfile
    world = args.gpus
    if args.launch_dry_run:
        ranks, local_of, rdzv_dir = list(range(world)), None, os.path.join(tempfile.gettempdir(), 'drx_rdzv_<fresh>')
    elif 'WORLD_SIZE' in os.environ:
        assert int(os.environ['WORLD_SIZE']) == world, f"--gpus {world} but WORLD_SIZE={os.environ['WORLD_SIZE']}"
        ranks = [int(os.environ.get('RANK', 0))]
        local_of = {ranks[0]: int(os.environ.get('LOCAL_RANK', ranks[0]))}
        rdzv_dir = _shared_rendezvous_dir(ranks[0])
    else:
        ranks, local_of = list(range(world)), None
        rdzv_dir = tempfile.mkdtemp(prefix='drx_rdzv_')
    out, rc = coordinate(args, argv, ranks, world, rdzv_dir, local_of)
    if out is not None:
        print(json.dumps(out), flush=True)
    return rc


def _init_group(dist, backend, rank, world, **kw):
    """The ranks of a measurement meet through the file store named by DRX_RDZV (set by the coordinator); a lone process
    (1-rank RCCL debugging run) makes its own.  Never a TCP port."""
    import tempfile
    rdzv = os.environ.get('DRX_RDZV') or f"file://{tempfile.mkdtemp(prefix='drx_rdzv_')}/solo"
    dist.init_process_group(backend, init_method=rdzv, rank=rank, world_size=world, **kw)


def selftest_child():
    """Children of --launch-selftest: a gloo rendezvous and one all-reduce on the CPU; rank 0 prints a line of the bench's shape."""
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    _init_group(dist, 'gloo', rank, world)
    t = torch.ones(1)
    dist.all_reduce(t)
    t0 = time.perf_counter()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({'metric': 'launcher selftest', 'selftest': True, 'value': float(world), 'unit': 'ranks', 'n_gpus': world,
                          'rccl_ranks': int(t.item()), 'backend': 'gloo', 'ms_per_step': float(dt.item()) * 1e3,
                          'config': {'sharding': 'none (launcher selftest)'}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------------------
# measurement helpers
# ------------------------------------------------------------------------------------------------------------------------------
def hbm_copy_gbs(dev, gib=2, reps=8):
    """Achievable HBM rate on this box: the library's own float4 copy kernel (drx_copy_f4) over `gib` GiB, read + write counted, GB/s
    (the guide measured 6.29 TB/s for such a kernel).  r06: through r05 this timed torch's `copy_` (4.96 TB/s)."""
    from drecpy_amd._lib import check, lib, ptr, stream_ptr
    n = gib * (1 << 30) // 4
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    run = lambda: check(lib().drx_copy_f4(ptr(dst), ptr(src), n * 4, stream_ptr(dev)), 'drx_copy_f4')
    run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    assert torch.equal(dst[:4096], src[:4096]) and torch.equal(dst[-4096:], src[-4096:])
    return 2.0 * n * 4 * reps / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9


def hash_u32_torch(seed, a, b):
    """drx_hash_u32 restated with wrapping int64 torch ops (to count surviving inputs of a batch exactly)."""
    def c(v):
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    def srl(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)
    x = a * c(0x9E3779B97F4A7C15) + b * c(0xD1B54A32D192ED03) + c(seed)
    x = (x ^ srl(x, 30)) * c(0xBF58476D1CE4E5B9)
    x = (x ^ srl(x, 27)) * c(0x94D049BB133111EB)
    x = x ^ srl(x, 31)
    return srl(x, 32)


def q_threshold(q):
    t = float(np.float32(q)) * 4294967296.0
    return 0 if t <= 0 else min(int(t), 0xFFFFFFFF)


def kept_count(keep_off, seed, q):
    B = keep_off.numel() - 1
    deg = (keep_off[1:] - keep_off[:-1]).long()
    row = torch.repeat_interleave(torch.arange(B, device=keep_off.device), deg)
    j = torch.arange(int(keep_off[-1].item()), device=keep_off.device) - keep_off[:-1].long()[row]
    return int((hash_u32_torch(seed, row, j) >= q_threshold(q)).sum().item())


SHARE_TRIPLES = 16          # csrc/drx_prep.hpp kShareTriples: triples of one work item of DRX_BATCH_SHARE_USERS


def batch_row_stats(indptr, indices, uid, iid, keep_off, seed, q, share=False):
    """Exact row statistics of ONE batch, counted on the device: per key class (W = input rows of the kept history items,
    V = user rows, O = W2T output rows) the number of touches (occurrences) and of DISTINCT rows, and the rows a single triple
    touches (the forward kernel updates those in place).  These are what the dedup-aware byte model of the roofline needs."""
    B = uid.numel()
    dev = uid.device
    deg = (keep_off[1:] - keep_off[:-1]).long()
    row = torch.repeat_interleave(torch.arange(B, device=dev), deg)
    j = torch.arange(int(keep_off[-1].item()), device=dev) - keep_off[:-1].long()[row]
    keep = hash_u32_torch(seed, row, j) >= q_threshold(q)
    pos = (indptr[uid.long()][row] + j)[keep]
    items = indices[pos]
    _, cw = torch.unique(items, return_counts=True)
    st = {'B': B, 'history_items': int(deg.sum().item()), 'occ_W': int(keep.sum().item()), 'dist_W': int(cw.numel()),
          'solo_W': int((cw == 1).sum().item())}
    for name, col in (('V', uid), ('O', iid)):
        _, cnt = torch.unique(col.long(), return_counts=True)
        st['dist_' + name], st['solo_' + name] = int(cnt.numel()), int((cnt == 1).sum().item())
    if share:
        # DRX_BATCH_SHARE_USERS (csrc/drx_prep.hpp k_tp_item_* / k_tp_expand<SHARE>): the samples of a user, ascending, in work items of
        # 16; per (work item, history position): nq samples, k of them keep the position -> one touch of the item's summed row and one
        # per dropper where that is the shorter form (1 + nq - k < k), else one per keeper.  The forward kernel loads a history row once
        # per work item.
        order = torch.argsort(uid.long(), stable=True)
        su = uid.long()[order]
        head = torch.ones(B, dtype=torch.bool, device=dev)
        head[1:] = su[1:] != su[:-1]
        run_start = torch.cummax(torch.where(head, torch.arange(B, device=dev), torch.zeros(B, dtype=torch.long, device=dev)), 0).values
        item_head = (torch.arange(B, device=dev) - run_start) % SHARE_TRIPLES == 0
        item_sorted = torch.cumsum(item_head.long(), 0) - 1
        witem = torch.empty(B, dtype=torch.long, device=dev)
        witem[order] = item_sorted
        n_items = int(item_sorted[-1].item()) + 1
        max_deg = int(deg.max().item()) + 1
        grp = witem[row] * max_deg + j
        ug, inv = torch.unique(grp, return_inverse=True)
        nq = torch.zeros(ug.numel(), dtype=torch.long, device=dev).scatter_add_(0, inv, torch.ones_like(inv))
        kk = torch.zeros(ug.numel(), dtype=torch.long, device=dev).scatter_add_(0, inv, keep.long())
        shared = (nq > 1) & (1 + nq - kk < kk)
        st['share_touches'] = int(torch.where(shared, 1 + nq - kk, kk).sum().item())
        st['share_item_rows'] = int(ug.numel())
        st['share_items'] = n_items
    return st


def byte_model(st, k, S, fused_solo, fused_solo_w=False, n_users=None, n_items=None):
    """Dedup-aware algorithmic HBM bytes of one sparse step, rows of 4k bytes, S optimizer slots per parameter:
      forward kernel : gathers one row per OCCURRENCE (kept W rows + V row + W2T row) from a table larger than the 256 MiB Infinity
                       Cache, one per DISTINCT row from a table that fits it (MovieLens shapes: the whole model is cache-resident and
                       re-reads never reach HBM); writes dz1[b] for every triple and g2[b] for the triples whose W2T row is shared; the
                       V / W2T (/ W) rows only this triple touches are updated in place from registers (read S slot rows, write
                       parameter + S slot rows) when the touch list was prepared ahead (fused_solo, fused_solo_w);
                       + the CSR indices of the batch users (4 B per history item) and 40 B of ids / offsets per triple
      reduction      : one read-modify-write of parameter + S slot rows per DISTINCT remaining row: (2 + 2S) rows, + 8 B per touch
                       (sorted key, sample).  The gradient rows it sums (dz1 / g2, one read per occurrence) were written by the forward
                       kernel just before — 2B rows = 67 MB at B = 65 536 — and are ASSUMED served by L2 / the 256 MiB Infinity Cache:
                       they are reported as cache_bytes, not as HBM bytes.
    Returns bytes per launch of each kernel."""
    row = 4.0 * k
    B = st['B']
    cache = 256.0 * (1 << 20)
    big_items = n_items is None or n_items * row > cache
    big_users = n_users is None or n_users * row > cache
    solo_V, solo_O = (st['solo_V'], st['solo_O']) if fused_solo else (0, 0)
    solo_W = st['solo_W'] if (fused_solo and fused_solo_w) else 0       # W rows with one touch, updated by the forward kernel too
    gather_rows = (st['occ_W'] if big_items else st['dist_W']) + (B if big_users else st['dist_V']) + (B if big_items else st['dist_O'])
    fwd = row * gather_rows + row * (B + (B - solo_O)) + row * (solo_V + solo_O + solo_W) * (1 + 2 * S) + 4.0 * st['history_items'] + 40.0 * B
    n_touch = st['occ_W'] - solo_W + (B - solo_V) + (B - solo_O)
    share = 'share_touches' in st
    if share:                        # the W part of the list in the shared form; one more gradient row written per work item
        n_touch = st['share_touches'] + (B - solo_V) + (B - solo_O)
        fwd += row * st['share_items']
    red = row * (st['dist_W'] - solo_W + st['dist_V'] - solo_V + st['dist_O'] - solo_O) * (2 + 2 * S) + 8.0 * n_touch
    # STRICTLY NECESSARY bytes: every gathered row counted once per DISTINCT row whatever the table's size — what
    # must cross the HBM interface even with a perfect cache.  The headline `frac` is priced on these: it can never exceed the
    # fraction the PMC counters support (r03: the per-occurrence model gave the forward kernel 0.71 against 0.52 of counted traffic).
    fwd_nec = fwd - row * (gather_rows - (st['dist_W'] + st['dist_V'] + st['dist_O']))
    # REQUESTED bytes, cache level: what the kernels ask of L2 / the Infinity Cache — one row per occurrence in the forward kernel,
    # one gradient row per touch in the reduction, on top of their HBM bytes (for cache-resident shapes this is the binding traffic)
    fwd_req = fwd_nec + row * (st['occ_W'] + 2 * B - (st['dist_W'] + st['dist_V'] + st['dist_O']))
    if share:                        # a history row once per work item, the user's V row once per work item, a W2T row per triple
        fwd_req = fwd_nec + row * (st['share_item_rows'] + st['share_items'] + B - (st['dist_W'] + st['dist_V'] + st['dist_O']))
    red_req = red + row * n_touch
    return {'k_sampled_fwd_bwd': fwd, 'k_seg_reduce': red, 'cache_bytes_k_seg_reduce': row * n_touch,
            'cache_bytes_k_sampled_fwd_bwd': row * (st['occ_W'] + 2 * B - gather_rows),
            'necessary_k_sampled_fwd_bwd': fwd_nec, 'necessary_k_seg_reduce': red,
            'requested_k_sampled_fwd_bwd': fwd_req, 'requested_k_seg_reduce': red_req,
            'cache_resident': not big_items and not big_users}


# the translation units and headers the sampled step's kernels (sampler, preparation, sort, forward / backward, reduction, spans) are
# built from: a PMC profile of those kernels is quoted for exactly this code.  (r06: through r05 the hash ran over every file of csrc/
# and include/drx.h, so an edit to the DMF or Caser kernels, or a new declaration, silenced the headline's traffic figures.)
SAMPLED_STEP_SOURCES = ('drx_cdae.hip', 'drx_sort.hip', 'drx_sampler.hip', 'drx_common.hpp', 'drx_rows.hpp', 'drx_segreduce.hpp',
                        'drx_scan.hpp', 'drx_prep.hpp', 'drx_segstream.hpp')


def kernel_source_hash():
    """sha256 over the sources the sampled step's kernels are built from: a PMC profile is only quoted for the code it was taken on."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'drecpy_amd', 'csrc')
    for name in sorted(SAMPLED_STEP_SOURCES):
        with open(os.path.join(src, name), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def hr_at_10(dev, with_cpu=True):
    """The HR@10 half of BASELINE.json's metric: CDAE in REFERENCE mode with the README configuration (K=50, q=0.2, BCE,
    100 one-batch epochs of 64, lr 1e-3, reg 1e-3, neg_ratio 5, seed 10) on the ml-100k-shaped synthetic set, leave-10-out,
    evaluated with the protocol of examples/cdae.py:15-17.  No MovieLens files exist offline: not comparable with
    README.md:140-141 (0.5536 on the real ml-100k); an untrained model scores ~10/101."""
    from drecpy_amd import synth
    from drecpy_amd.Dataset import InteractionDataset
    from drecpy_amd.Evaluation import leave_k_out, ranking_evaluation
    from drecpy_amd.Recommender import CDAE
    U, N, md, mn, a = synth.SHAPES['ml-100k']
    ip, idx = synth.synth_history(U, N, md + 12, mn, a, seed=0)
    ip, idx = ip.numpy(), idx.numpy()
    rng = np.random.RandomState(0)
    user = np.repeat(np.arange(U), np.diff(ip)) + 1
    perm = rng.permutation(len(user))
    ds = InteractionDataset.read_df({'user': user[perm], 'item': (idx.astype(np.int64) + 1)[perm],
                                     'interaction': rng.randint(1, 6, size=len(user))[perm]}, verbose=False)
    tr, te = leave_k_out(ds, k=10, min_user_interactions=10, seed=10, verbose=False)
    m = CDAE(hidden_factors=50, corruption_level=0.2, loss='bce', seed=10, verbose=False, device=str(dev))
    m.fit(tr, epochs=2, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)       # loads the dense-mode kernels once
    t0 = time.perf_counter()
    m.fit(tr, epochs=100, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)     # a fresh fit: new tables, new sampler
    torch.cuda.synchronize()
    fit_s = time.perf_counter() - t0
    res = ranking_evaluation(m, te, k=[1, 5, 10], novelty=True, n_test_users=100, n_pos_interactions=1, n_neg_interactions=100,
                             generate_negative_pairs=True, seed=10, verbose=False)
    # the same configuration over a fit long enough for set-up and the GPU's clock ramp not to dominate (the metric's "CDAE ml-100k"
    # training rate): a third fresh fit, 5000 one-batch epochs
    t0 = time.perf_counter()
    m2 = CDAE(hidden_factors=50, corruption_level=0.2, loss='bce', seed=10, verbose=False, device=str(dev))
    m2.fit(tr, epochs=5000, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
    torch.cuda.synchronize()
    long_s = time.perf_counter() - t0
    from bench_cpu import cpu_reference_fit
    cpu_ref = cpu_reference_fit(tr) if with_cpu else None
    return {'value': res['HitRatio@10'], 'ndcg_at_10': res['NDCG@10'], 'cpu_baseline_reference_mode': cpu_ref,
            'fit_seconds_100_steps_of_64': round(fit_s, 3),
            'fit_samples_per_s_incl_host': round(6400 / fit_s, 1),
            'fit_seconds_5000_steps_of_64': round(long_s, 3), 'fit_samples_per_s_5000_steps_incl_host_and_setup': round(320000 / long_s, 1),
            'setup': 'CDAE reference mode (dense Keras Adam), README.md:106-114 configuration, ml-100k-shaped synthetic '
                     f'({len(tr)} train / {len(te)} test rows), protocol examples/cdae.py:15-17'}



class Windows:
    """`windows` back-to-back timed windows of exactly `steps` steps, each bracketed by barrier + synchronize and reduced by MAX over
    ranks.  Phase events are recorded by the library on every EVERY-th step of every window."""

    def __init__(self, steps, windows, n_events, world, dist, dev):
        self.steps, self.windows, self.world, self.dist, self.dev = steps, max(1, windows), world, dist, dev
        # six event records per step cost ~4 % of the step: only every EVERY-th timed step carries them (r06: was 4 — each costs
        # ~28 us: the records between launches keep the next kernel from starting early)
        # short windows: two event-carrying steps per window, evenly (20-step windows with a cadence of 8 alternated between two and three of
        # them: the median window was always one of the slower kind); long windows: every 16th step
        auto = max(8, self.steps // 2) if self.steps <= 64 else 16
        self.every = max(1, int(os.environ.get('DRX_BENCH_EVENTS_EVERY', auto)))
        total = self.steps * self.windows
        self.evs = [[torch.cuda.Event(enable_timing=True) for _ in range(n_events)] if i % self.every == 0 else None for i in range(total)]
        for es in self.evs:
            for e in (es or []):
                e.record()
        self.times = []

    def _fence(self):
        torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        torch.cuda.synchronize()

    def run(self, step_fn):
        """step_fn(i, events_or_None, last): i = 0-based index over all timed steps."""
        total = self.steps * self.windows
        for w in range(self.windows):
            self._fence()
            t0 = time.perf_counter()
            for j in range(self.steps):
                i = w * self.steps + j
                step_fn(i, self.evs[i], i == total - 1)
            self._fence()
            dt = time.perf_counter() - t0
            if self.world > 1:
                t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                dt = float(t.item())
            self.times.append(dt)
        return self

    @property
    def median(self):
        return float(np.median(self.times))

    def fields(self, samples_per_step):
        ms = [t * 1e3 for t in self.times]
        return {'value': samples_per_step * self.steps / self.median, 'ms_per_step': self.median / self.steps * 1e3,
                'windows': self.windows, 'window_ms': [round(m, 4) for m in ms], 'window_ms_min': min(ms), 'window_ms_max': max(ms),
                'value_from': 'median window'}

    def phase_means(self, n_phases):
        rows = [[es[i].elapsed_time(es[i + 1]) for i in range(n_phases)] for es in self.evs if es is not None]
        return np.array(rows).mean(axis=0), len(rows)


def rccl_ranks(dist, dev, world, debug_gloo):
    """Sum over ranks of an all-reduce of ones: the number of ranks that really took part in a collective of this job's backend."""
    if world == 1 and not dist.is_initialized():
        return 1
    t = torch.ones(1, device='cpu' if debug_gloo else dev)
    dist.all_reduce(t)
    return int(t.item())


def run_columns(args, rank, world, dev, dist, debug_gloo, rccl1):
    """Column-sharded layout (dist.ColumnShardedCdae): the job's global batch (B per GPU x N GPUs) is drawn identically on every
    rank, each rank trains its K/N columns of every table on all of it; one all-reduce of B_global floats per step."""
    from drecpy_amd import synth
    from drecpy_amd.dist import ColumnShardedCdae
    U, N, md, mn, alpha = synth.SHAPES[args.workload]
    if args.users:
        U = args.users
    Bg = args.batch * world
    t_setup = time.time()
    indptr, indices = synth.synth_history(U, N, md, mn, alpha, seed=0, device=dev)          # every rank: all users
    nnz = int(indptr[-1].item())
    model = ColumnShardedCdae(U, N, K, rank, world, dev, indptr, indices, seed=10, lr=1e-3 if args.optimizer == 'adam' else LR, reg=REG,
                              optimizer=args.optimizer, q=Q, cpu_staging=debug_gloo, force_collectives=rccl1)
    eng = model.engine
    uid, iid, y, keep_off = eng.sample_device(Bg, NEG_RATIO, 1000, n_items=N)
    rows_per_sample = kept_count(keep_off, 1000, Q) / Bg + 2.0
    f_solo = 0.0
    for col in (uid.long(), iid.long()):
        _, inv, cnt = torch.unique(col, return_inverse=True, return_counts=True)
        f_solo += float((cnt[inv] == 1).float().mean().item())
    setup_s = time.time() - t_setup
    seed_of = lambda s: 5000 + 7919 * s                                                              # same seeds on every rank
    emu = int(os.environ.get('DRX_BENCH_EMULATE_RANKS', 0))
    if args.prepare == 'auto':
        args.prepare = 'turns' if max(world, emu) >= 4 else 'local'
    if emu > 1 and world == 1 and args.prepare != 'local':
        # debugging aid (one GPU standing in for one rank of `emu`): the batches cycle and what the other ranks would send comes
        # from a cache filled at the first encounter; a device copy stands in for the broadcast / all-gather
        seed_of = lambda s: 5000 + 7919 * (s % args.n_batches)
        cache, recv = {}, [None, None]

        def emulated_build(s, bt, out):
            c = s % args.n_batches
            if c not in cache:
                cache[c] = eng.prepare_sparse(bt).clone()
            return eng.prepare_sparse(bt, out) if s % emu == 0 else eng.prep_buffer(bt, out)

        def emulated_deliver(s, bt, out):
            if s % emu:
                n = eng.prep_result_bytes(bt)
                out[:n].copy_(cache[s % args.n_batches][:n])

        def emulated_parts(s, bt, out):
            c = s % args.n_batches
            if c not in cache:
                cache[c] = torch.cat([eng.prepare_part(bt, r, emu).clone() for r in range(emu)])
            part = eng.prepare_part(bt, 0, emu, slot=s % 2)
            n = cache[c].numel()
            if recv[s % 2] is None or recv[s % 2].numel() < n:
                recv[s % 2] = torch.empty(int(n * 1.05), dtype=torch.uint8, device=dev)
            got = recv[s % 2][:n]
            got.copy_(cache[c])
            got[:part.numel()].copy_(part)
            out, model._oflow[s % 2] = eng.prepare_assemble(bt, got, emu, out, model._oflow[s % 2])
            return out
        model.prepare_mode = args.prepare
        model.prepare, model.build_in_turns, model.deliver_in_turns = emulated_parts, emulated_build, emulated_deliver
    elif world > 1:
        model.prepare_mode = args.prepare
    pipe = model.pipeline(Bg, NEG_RATIO, seed_of, seed_of)
    for _ in range(args.warmup):
        pipe.run_step()
    win = Windows(args.steps, args.windows, 7, world, dist, dev)

    def step_fn(i, es, last):
        if es is not None:
            es[0].record()                        # [0,1): forward half + all-reduce of the partial dot products
            pipe.run_step(events=es[1:])
        else:
            pipe.run_step()
    win.run(step_fn)
    ph, n_timed = win.phase_means(6)
    names = ['k_kshard_fwd+allreduce(dot)', 'k_kshard_rest', 'touch_sort(overlapped on side stream)', 'k_seg_reduce_planned(+bias partials)',
             'k_span_planned(short | long spans | bias update)', '(unused)']
    kl = model.k_hi - model.k_lo
    S_opt = 2.0 if args.optimizer == 'adam' else 1.0
    alg_upd = Bg * 4.0 * kl * (rows_per_sample - f_solo) * (2.0 + 2.0 * S_opt)      # this rank's columns of the global batch
    alg_fwd = Bg * 4.0 * kl * rows_per_sample
    dom, dom_ms, dom_alg = ('k_seg_reduce_planned', ph[3], alg_upd) if ph[3] >= ph[0] else ('k_kshard_fwd+allreduce(dot)', ph[0], alg_fwd)
    step_alg = Bg * 4.0 * kl * rows_per_sample * (3.0 + 2.0 * S_opt)
    n_ranks = rccl_ranks(dist, dev, world, debug_gloo)
    if rank != 0:
        return None
    copy_gbs = hbm_copy_gbs(dev)
    step_s = win.median / args.steps
    out = {'metric': 'training samples/sec (user-item pairs)', 'unit': 'samples/s', 'n_gpus': world,
           'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'rccl_ranks': n_ranks,
           'config': {'workload': f'CDAE hidden_factors={K} sampled-output sparse-{"Adam (lazy)" if S_opt == 2.0 else "Adagrad"} on '
                                  f'{args.workload}-shaped synthetic ({U} users x {N} items, {nnz} positives), corruption {Q}, neg_ratio {NEG_RATIO}',
                      'batch_per_gpu': args.batch, 'global_batch': Bg, 'rows_per_sample': round(rows_per_sample, 3),
                      'sole_toucher_rows_per_sample': round(f_solo, 3),
                      'batches': 'fresh device-sampled global batch every step, drawn identically on every rank (sampler running ahead on a side stream)',
                      'sharding': f'columns: every rank holds all rows x {kl} of {K} columns and trains on the whole global batch; '
                                  f'one all-reduce of {Bg} floats per step; touch list '
                                  + {'local': 'sorted whole on every rank', 'turns': 'of step s sorted by rank s % N and broadcast on a side stream',
                                     'parts': 'sorted in parts (1/N per rank) + one all-gather on a side stream'}[model.prepare_mode]},
           'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': dom_alg / (dom_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': None, 'model_frac': dom_alg / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                        'algorithmic_bytes_per_launch': dom_alg, 'avg_launch_ms': float(dom_ms), 'timed_launches': n_timed,
                        'whole_step_achieved': step_alg / step_s / 1e9,
                        'model_whole_step_frac': step_alg / step_s / 1e9 / HBM_PEAK_GBS, 'hbm_copy_achievable': copy_gbs,
                        'note': 'per-rank figures: this rank\'s K/N columns of the global batch; bytes per SURVEY 8d charge a '
                                'read-modify-write per touched-row occurrence, the kernel merges occurrences first (DESIGN.md section 3), '
                                'so these are model_* numbers, not fractions'},
           'phases_ms': {n: float(v) for n, v in zip(names, ph)}, 'setup_s': round(setup_s, 1), 'cpu_baseline': None, 'hr_at_10': None}
    out.update(win.fields(Bg))
    return out


def _trace(msg):
    if os.environ.get('DRX_BENCH_TRACE'):
        print(f'[bench {time.time():.3f}] {msg}', file=sys.stderr, flush=True)


def run_direct(args, rank, world, dev, dist, debug_gloo=False, rccl1=False):
    """The single-GPU step (world 1) or the ROW-sharded layout (dist.ShardedCdae; world > 1 or --force-sharded).  Returns the line's
    dict on rank 0, None elsewhere."""
    from drecpy_amd import synth
    from drecpy_amd.engine import CdaeEngine
    U, N, md, mn, alpha = synth.SHAPES[args.workload]
    if args.users:
        U = args.users
    B = args.batch
    lo, hi = U * rank // world, U * (rank + 1) // world
    t_setup = time.time()
    indptr, indices = synth.synth_history(U, N, md, mn, alpha, seed=0, device=dev, user_lo=lo, user_hi=hi)
    nnz_local = int(indptr[-1].item())

    if world == 1 and not args.force_sharded:
        eng = CdaeEngine(hi - lo, N, K, device=dev)
        eng.share_users = not args.no_share_users
        eng.sample_by_user = not args.no_sample_by_user
        eng.init_glorot_device(10)
        eng.set_history(indptr, indices)
        eng.init_optimizer(args.optimizer, 1e-3 if args.optimizer == 'adam' else LR, REG)
        stepper = None
    else:
        from drecpy_amd.dist import RcclTransport, ShardedCdae
        transport = 'rccl' if (args.transport == 'rccl' and not debug_gloo and (world > 1 or rccl1)) else None
        if transport and args.no_comm_thread and world == 1:        # (A/B at world 1: the nccl* calls made by the posting thread itself)
            transport = RcclTransport(1, 0, dev, threaded=False)
        stepper = ShardedCdae(U, N, K, rank, world, dev, indptr, indices, seed=10, lr=LR, reg=REG, q=Q,
                              cpu_staging=debug_gloo, force_collectives=rccl1, self_bypass=not args.no_self_bypass, chunks=args.chunks,
                              transport=transport, phases=False if args.no_phases else None)
        eng = stepper.engine

    micro = max(1, args.micro)       # > 1: micro-batches whose exchanges overlap each other's compute (measured at world 1: the split costs more than it hides)
    # ---- pre-sampled batches, resident in HBM -------------------------------------------------------------
    batches, structs, kept_tot = [], [], 0
    for i in range(args.n_batches):
        seed = 1000 + 7919 * i + 104729 * rank
        uid, iid, y, keep_off = eng.sample_device(B, NEG_RATIO, seed, n_items=N)
        torch.cuda.synchronize()
        n_slots = int(keep_off[-1].item())
        kept_tot += kept_count(keep_off, seed, Q)
        batches.append((uid, iid, y, keep_off, seed))
        if stepper is not None and micro > 1:
            # micro-batches of a sharded step: the triples split by uid so that no user is in two of them
            parts = []
            for m in range(micro):
                ix = torch.nonzero(uid % micro == m).squeeze(1)
                um = uid[ix].contiguous()
                deg = (indptr[um.long() + 1] - indptr[um.long()]).to(torch.int32)
                ko = torch.zeros(um.numel() + 1, dtype=torch.int32, device=dev)
                ko[1:] = torch.cumsum(deg, 0)
                parts.append(eng.make_batch(um, iid[ix].contiguous(), y[ix].contiguous(), keep_off=ko, q=Q, mask_seed=seed + 1000003 * m,
                                            n_touch_slots=int(ko[-1].item())))
            structs.append(([p_[0] for p_ in parts], parts))
        else:
            bt, alive = eng.make_batch(uid, iid, y, keep_off=keep_off, q=Q, mask_seed=seed, n_touch_slots=n_slots)
            structs.append((bt, alive))
    rows_per_sample = kept_tot / (args.n_batches * B) + 2.0          # R: kept W rows + V row + W2T row
    setup_s = time.time() - t_setup
    _trace('setup done')

    # The touch list of a batch (sorted row keys) does not depend on the parameters: it is prepared ahead on a side stream
    overlap = not args.no_overlap
    main = torch.cuda.current_stream()
    from drecpy_amd.engine import run_ahead_stream
    side = run_ahead_stream(dev, 0) if overlap else None            # (the process-wide pool: probed for a hardware queue of its own)
    prep_bufs = [None, None]
    prep_done = [torch.cuda.Event(), torch.cuda.Event()]
    step_done = [torch.cuda.Event(), torch.cuda.Event()]

    fresh = overlap and stepper is None and not args.presampled
    # fresh mode: the device PointSampler draws a NEW batch for every step ahead on a side stream and the batch's touch list is sorted
    # ahead too — drecpy_amd.engine.SampledPipeline, the code path of CDAE.fit(mode='sampled', device_sampler=True).
    spipe = None
    if fresh:
        from drecpy_amd.engine import SampledPipeline
        spipe = SampledPipeline(eng, B, NEG_RATIO, Q, lambda s: 5000 + 7919 * s + 104729 * rank,
                                lambda s: 5000 + 7919 * s + 104729 * rank, n_items=N, prep_ahead=args.prep_ahead,
                                side_streams=args.side_streams, side_cus_per_xcd=None if args.prep_cus < 0 else args.prep_cus)

    def batch_of(s):
        return structs[s % len(structs)][0]

    prep_cache = {} if os.environ.get('DRX_BENCH_CACHE_PREP') == '1' else None     # diagnostic (with --presampled): every list built once

    def prepare(s):
        bt = batch_of(s)
        if prep_cache is not None:
            j = s % len(structs)
            if j not in prep_cache:
                with torch.cuda.stream(side):
                    prep_cache[j] = eng.prepare_sparse(bt, None)
            prep_bufs[s % 2] = prep_cache[j]
            prep_done[s % 2].record(side)
            return
        side.wait_event(step_done[s % 2])            # the buffer's previous user (step s-2) must be finished
        with torch.cuda.stream(side):
            prep_bufs[s % 2] = eng.prepare_sparse(bt, prep_bufs[s % 2])
            prep_done[s % 2].record(side)

    pipe, fresh_sharded = None, False
    total_steps = args.warmup + args.steps * max(1, args.windows)
    if stepper is not None and overlap:
        from drecpy_amd.dist import ShardedPipeline
        fresh_sharded = not args.presampled and micro == 1 and not debug_gloo
        if fresh_sharded:      # a new device-sampled batch of this rank's users every step, drawn ahead on its own stream
            from drecpy_amd.engine import DeviceBatchSource
            source = DeviceBatchSource(eng, B, NEG_RATIO, Q, lambda s: 5000 + 7919 * s + 104729 * rank,
                                       lambda s: 5000 + 7919 * s + 104729 * rank, n_items=N,
                                       stream_slot=0)         # (the pipeline's own run-ahead stream: a rank's streams share four hardware queues)
        else:
            source = lambda s: structs[s % len(structs)][0]
        pipe = ShardedPipeline(stepper, source, total_steps)

    def run_step(s, events=None, last=False):
        if pipe is not None:           # keys of batch s+1 and counts of batch s+2 travel ahead of step s (dist.ShardedPipeline)
            pipe.run_step(events=events)
            return
        if spipe is not None:
            spipe.run_step(events=events)
            return
        if not overlap:
            bt = batch_of(s)
            if stepper is not None:
                stepper.step(s, bt, events=events)
            else:
                eng.step_sparse(s, bt, 'bce', events=events)
            return
        if not last:
            prepare(s + 1)
        bt = batch_of(s)
        main.wait_event(prep_done[s % 2])
        eng.step_sparse(s, bt, 'bce', events=events, prepared=prep_bufs[s % 2])
        step_done[s % 2].record(main)

    for e in step_done:
        e.record(main)
    if overlap and pipe is None and spipe is None:
        prepare(0)
    _trace('pipeline created')
    for s in range(args.warmup):
        run_step(s)
    _trace('warm-up queued')
    win = Windows(args.steps, args.windows, 6, world, dist, dev)
    win.run(lambda i, es, last: run_step(args.warmup + i, events=es, last=last))
    _trace('windows done')
    n_ranks = rccl_ranks(dist, dev, world, debug_gloo) if dist is not None else 1
    ph, n_timed = win.phase_means(5)
    step_s = win.median / args.steps
    # SURVEY 8d's per-occurrence model (kept as model_*): forward reads 4K*R per sample; the update reads and writes parameter + S
    # optimizer slots per touched-row occurrence: 4K*R*(2+2S), S = 1 for Adagrad.
    f_solo = 0.0
    if stepper is None and overlap:
        uid0, iid0 = batches[0][0].long(), batches[0][1].long()
        for col in (uid0, iid0):
            _, inv, cnt = torch.unique(col, return_inverse=True, return_counts=True)
            f_solo += float((cnt[inv] == 1).float().mean().item())
    # optimizer slots per parameter (row-wise Adagrad keeps 1/K of a slot: one float per row)
    S_opt = {'adam': 2.0, 'adagrad': 1.0, 'rowwise_adagrad': 1.0 / K}[args.optimizer] if stepper is None else 1.0
    alg_fwd = B * 4.0 * K * (rows_per_sample + f_solo * (2.0 + 2.0 * S_opt))
    alg_upd = B * 4.0 * K * (rows_per_sample - f_solo) * (2.0 + 2.0 * S_opt)
    # the lists of this run are in the shared form (DRX_BATCH_SHARE_USERS; csrc/drx_prep.hpp share_users(): the history's transpose exists,
    # long segments, rows of 33 .. 255 floats, lists prepared ahead): the forward kernel is then k_items_fwd_bwd
    may_share = bool(stepper is None and overlap and getattr(eng, '_hist_t', None) is not None and eng.share_users and 32 < K < 256)
    shared_run = may_share and int(batches[0][3][-1].item()) + 2 * B > 8 * (2 * N + (hi - lo))
    FWD = 'k_items_fwd_bwd' if shared_run else 'k_sampled_fwd_bwd'
    # lists of SHORT segments over rows of exactly 64 / 128 / 256 floats, Adagrad: the streamed reduction (csrc/drx_segstream.hpp)
    long_run = int(batches[0][3][-1].item()) + 2 * B > 8 * (2 * N + (hi - lo))              # csrc/drx_prep.hpp long_segments()
    # (the row layout's local reduction is the same kernel under LocalPolicyT: csrc/drx_shard.hip launches the streamed form on the same
    # condition; the sharded stepper is always Adagrad here)
    RED = 'k_seg_reduce_stream' if ((stepper is not None or args.optimizer == 'adagrad') and K in (64, 128, 256) and not long_run) else 'k_seg_reduce_planned'
    if stepper is None:
        names = [FWD, 'touch_sort(overlapped on side stream)' if overlap else 'touch_sort', RED + '(+bias partials)',
                 'k_span_planned(short | long spans | bias update)', '(unused)']
        dom, dom_ms, dom_alg = (RED, ph[2], alg_upd) if ph[2] >= ph[0] else (FWD, ph[0], alg_fwd)
    else:
        names = (['wait for the rows (fetched behind the previous step\'s apply)', 'k_shard_fwd_bwd', RED + '<LocalPolicy>(+bias partials)',
                  'k_span_planned(+bias row)', 'grad_exchange+k_shard_apply(+bias update)+next step\'s row_gather+row_exchange, chunk by chunk'] if micro == 1 else
                 ['row_gather+first_row_exchange', 'fwd_bwd+local_reduce of all micro-batches', '-', '-', 'grad_exchange+k_shard_apply'])
        # forward reads one row per occurrence, the local reduce one gradient row per occurrence
        dom, dom_ms, dom_alg = 'k_shard_fwd_bwd+k_seg_reduce<LocalPolicy>', ph[1] + ph[2], 2.0 * B * 4.0 * K * rows_per_sample
    achieved = dom_alg / (dom_ms * 1e-3) / 1e9
    step_alg = B * 4.0 * K * rows_per_sample * (3.0 + 2.0 * S_opt)
    # ---- dedup-aware byte model: exact occurrence / distinct-row counts of batches of the TIMED region ----------------------------
    dedup = None
    n_timed_steps = args.steps * max(1, args.windows)
    if stepper is None or (world == 1 and micro == 1):
        picks = sorted(set(int(x) for x in np.linspace(0, n_timed_steps - 1, 6)))
        stats = []
        for s_ in picks:
            if fresh or fresh_sharded:                  # the pipeline drew step (warmup + s_) from these seeds: draw it again
                seed_ = 5000 + 7919 * (args.warmup + s_) + 104729 * rank
                u_, i_, _, ko_ = eng.sample_device(B, NEG_RATIO, seed_, n_items=N)
            else:
                u_, i_, _, ko_, seed_ = batches[(args.warmup + s_) % len(batches)]
            long_segments = int(ko_[-1].item()) + 2 * B > 8 * (2 * N + (hi - lo))             # csrc/drx_prep.hpp long_segments()
            stats.append(batch_row_stats(indptr, indices, u_, i_, ko_, seed_, Q, share=may_share and long_segments))
        _trace('row statistics done')
        mean_st = {k_: float(np.mean([st[k_] for st in stats])) for k_ in stats[0]}
        solo_w_on = False           # (the variant whose forward kernel also updated W rows with one touch was removed in r04: HISTORY.md)
        bm = byte_model(mean_st, K, S_opt, fused_solo=overlap, fused_solo_w=solo_w_on, n_users=hi - lo, n_items=N)
        dedup = {'batches_counted': len(stats), 'per_batch_mean': {k_: round(v, 1) for k_, v in mean_st.items()}, 'bytes_per_launch': bm,
                 'sole_toucher_w_rows_fused': solo_w_on}
    kernel_hash = kernel_source_hash()
    # HBM traffic from the PMC passes (profiles/pmc_traffic.json), only when that profile was taken on this very workload AND on
    # these very kernel sources (scripts/profile_round.sh stores their hash); rocprofv3 cannot run inside the bench itself.
    traffic_of, hit_of, traffic_note = {}, {}, 'no PMC profile for this workload'
    pmc_name = 'pmc_traffic.json' if args.workload == 'synth-10m' else f'pmc_traffic_{args.workload}.json'
    try:
        with open(os.path.join(ROOT, 'profiles', pmc_name)) as f:
            pmc = json.load(f)
        m = pmc['_meta']
        if m['workload'] == args.workload and m['batch_per_gpu'] == B and m['n_gpus'] == world and not args.users and K == 128 \
                and m.get('optimizer', 'adagrad') == args.optimizer:
            if m.get('kernel_source_hash') == kernel_hash:
                traffic_of = {k_.replace('drx::', ''): v['hbm_bytes_per_launch'] for k_, v in pmc['kernels'].items()}
                hit_of = {k_.replace('drx::', ''): v['l2_hit_rate'] for k_, v in pmc['kernels'].items() if v.get('l2_hit_rate') is not None}
                traffic_note = f"profiles/{pmc_name} ({m.get('round')}), same kernel sources ({kernel_hash})"
            else:
                traffic_note = f"profiles/{pmc_name} is STALE: taken on kernel sources {m.get('kernel_source_hash')}, this tree is {kernel_hash}"
    except (OSError, KeyError, ValueError):
        pass
    traffic = traffic_of.get(dom)
    if rank != 0:
        return None
    copy_gbs = hbm_copy_gbs(dev)
    out = {
        'metric': 'training samples/sec (user-item pairs)', 'unit': 'samples/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'rccl_ranks': n_ranks,
        'config': {'workload': f'CDAE hidden_factors={K} sampled-output sparse-{dict(adam="Adam (lazy)", adagrad="Adagrad", rowwise_adagrad="row-wise Adagrad")[args.optimizer if stepper is None else "adagrad"]} on {args.workload}-shaped synthetic '
                               f'({U} users x {N} items, {nnz_local * world if world > 1 else nnz_local} positives), '
                               f'corruption {Q}, neg_ratio {NEG_RATIO}',
                   'batch_per_gpu': B, 'global_batch': B * world, 'rows_per_sample': round(rows_per_sample, 3),
                   'sole_toucher_rows_per_sample': round(f_solo, 3),
                   'touch_list': ('keys exchanged one batch ahead, counts two (dist.ShardedPipeline)' if pipe is not None else 'prepared ahead on a side stream') if overlap else 'inline',
                   'batches': 'fresh device-sampled batch every step (sampler running ahead on the side stream)' if (fresh or (pipe is not None and fresh_sharded))
                   else f'{args.n_batches} pre-sampled batches cycled',
                   'micro_batches': (micro if stepper is not None else None),
                   'exchange_chunks': (stepper.chunks if stepper is not None else None),
                   'transport': (type(stepper.xfer).__name__ if stepper is not None else None),
                   'exchanges_issued_by': (('library (drx_shard_phase_*)' if (stepper.phases and max(1, args.micro) == 1) else 'dist.py, call by call')
                                           if stepper is not None else None),
                   'sharding': ('single GPU' if stepper is None else
                                'row-sharded code path at world 1, ' + ('every row sent through the communicator (--no-self-bypass: all rows "remote", the link '
                                                                        'replaced by a device copy)' if args.no_self_bypass else
                                                                        'own rows bypass the collectives (at world 1: all of them; the exchanges are empty)')) if world == 1
                   else f'rows: users (V rows, histories, samples) sharded x{world} by range, item rows sharded by range; all-to-all(v) of rows and gradient '
                        f'rows (one float buffer per direction; a rank\'s own rows bypass them), bias gradient in the sentinel rows of the gradient exchange'},
        'roofline': None,
        'phases_ms': {n: float(v) for n, v in zip(names, ph)},
        'setup_s': round(setup_s, 1),
        'host_issue_ms_per_step': ([round(1e3 * t / total_steps, 4) for t in pipe.host_s + [stepper.wait_s]] if pipe is not None else None),
    }
    out.update(win.fields(world * B))
    if stepper is None and len(ph) >= 4:
        # (i) end to end incl. the sampler = `value` (a fresh batch is drawn and prepared on the device for every step, beside
        # the training kernels); (ii) the training launches alone, HIP events around each of them on the timed steps that carry events
        kern_ms = float(ph[0]) + float(ph[2]) + float(ph[3])
        out['device_only'] = {'ms_per_step': kern_ms, 'value': world * B / (kern_ms * 1e-3),
                              'what': 'sum of the three training launches between their HIP events (forward, reduction, spans) on the timed '
                                      'steps: the step without the launch gaps; the draw and the preparation of later batches run beside them'}
    ms_of = {'k_sampled_fwd_bwd': float(ph[0]), 'k_seg_reduce': float(ph[2])} if stepper is None else {}
    if dedup is not None and stepper is not None:
        # row layout at world 1: strictly necessary HBM bytes of its three big kernels from the same exact row counts —
        #   forward   : every distinct W / W2T row once (from the tables with the self-bypass, from the received buffer otherwise), the V
        #               row of every triple, dz1 written, g2 written where the W2T row is shared, else the gradient row straight into the
        #               exchange buffer; V rows one triple touches updated in place (S slot rows in, parameter + S slot rows out)
        #   reduction : one gradient row WRITTEN per distinct W row and shared W2T row, 8 B per touch, shared V rows read-modify-written
        #   apply     : per distinct item row the gradient row in, parameter + S slot rows in and out
        st_, row_ = dedup['per_batch_mean'], 4.0 * K
        dist_item = st_['dist_W'] + st_['dist_O']
        nb = {'k_shard_fwd_bwd': row_ * (dist_item + st_['dist_V'] + B + (B - st_['solo_O']) + st_['solo_O'] + st_['solo_V'] * (1 + 2 * S_opt))
                                 + 4.0 * st_['history_items'] + 40.0 * B,
              RED: row_ * (st_['dist_W'] + (st_['dist_O'] - st_['solo_O']) + (st_['dist_V'] - st_['solo_V']) * (2 + 2 * S_opt))
                                      + 8.0 * (st_['occ_W'] + (B - st_['solo_O']) + (B - st_['solo_V'])),
              'k_shard_apply': row_ * dist_item * (3 + 2 * S_opt)}
        tm = {'k_shard_fwd_bwd': float(ph[1]), RED: float(ph[2]), 'k_shard_apply': float(ph[4])}
        per_kernel = {k_: {'bytes_per_launch': nb[k_], 'avg_launch_ms': tm[k_], 'achieved': nb[k_] / (tm[k_] * 1e-3) / 1e9,
                           'frac': nb[k_] / (tm[k_] * 1e-3) / 1e9 / HBM_PEAK_GBS} for k_ in nb}
        domk = max(tm, key=lambda k_: tm[k_])
        step_bytes = sum(nb.values())
        out['roofline'] = {'bound': 'hbm', 'kernel': domk, 'achieved': per_kernel[domk]['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': per_kernel[domk]['frac'], 'traffic': None, 'kernels': per_kernel, 'row_counts': dedup['per_batch_mean'],
                           'avg_launch_ms': tm[domk], 'timed_launches': n_timed, 'whole_step_bytes': step_bytes,
                           'whole_step_frac': step_bytes / step_s / 1e9 / HBM_PEAK_GBS, 'hbm_copy_achievable': copy_gbs,
                           'definition': 'row layout at world 1: strictly necessary HBM bytes of its three big kernels (bench.py) / HIP-event time between '
                                         'the library\'s launches / 8 TB/s; `k_shard_apply` time includes the (empty, with the self-bypass) gradient exchange'}
        dedup = None
    if dedup is not None:
        # `frac` = dedup-aware algorithmic HBM bytes / measured launch time / peak: a fraction by construction (what has to
        # cross the HBM interface at least once; everything re-read is assumed cached).  `model_*` = SURVEY 8d's
        # per-OCCURRENCE model (charges a read-modify-write per touch: the kernel merges touches first, so it can exceed 1).
        per_kernel = {}
        bpl = dedup['bytes_per_launch']
        for kn in ('k_sampled_fwd_bwd', 'k_seg_reduce'):
            byt, occ, req, ms_ = bpl['necessary_' + kn], bpl[kn], bpl['requested_' + kn], ms_of[kn]
            kname = FWD if kn != 'k_seg_reduce' else RED
            rate = lambda b_: b_ / (ms_ * 1e-3) / 1e9
            # cache level: the rows the kernel REQUESTS (one per occurrence / per touch) against ONE bound — the guide's two gather
            # rates mixed by the L2 hit rate the counters measured on this kernel (TCC_HIT / TCC_MISS, the same profile script):
            # hit x 17 TB/s (rows served by the XCDs' L2) + (1 - hit) x 8.6 TB/s (rows served by the Infinity Cache).  Quoted only with
            # a profile of these very kernel sources; r04 printed the two rates as two "peaks", one of which the kernel exceeded.
            hit = hit_of.get(kname)
            bound = (hit * L2_GATHER_GBS + (1.0 - hit) * MALL_GATHER_GBS) if hit is not None else None
            ref_ms = 1e3 * ((req - byt) / (MB_GATHER_GBS * 1e9) + byt / (MB_RMW_COLD_GBS * 1e9))
            per_kernel[kname] = {'bytes_per_launch': byt, 'avg_launch_ms': ms_, 'achieved': rate(byt), 'frac': rate(byt) / HBM_PEAK_GBS,
                                 'per_occurrence_bytes': occ, 'per_occurrence_frac': rate(occ) / HBM_PEAK_GBS,
                                 'traffic': traffic_of.get(kname),
                                 'traffic_frac': (rate(traffic_of[kname]) / HBM_PEAK_GBS) if kname in traffic_of else None,
                                 'requested_bytes': req, 'requested_GBs': rate(req), 'l2_hit_rate': hit, 'cache_bound_GBs': bound,
                                 'requested_frac_of_cache_bound': (rate(req) / bound) if bound else None,
                                 # context: the kernel's bytes priced at what micro-benchmarks of the two access patterns delivered
                                 'measured_reference': {'cache_served_bytes': req - byt, 'hbm_bytes': byt, 'reference_ms': ref_ms,
                                                        'time_over_reference': ms_ / ref_ms if ref_ms > 0 else None}}
        step_bytes = bpl['necessary_k_sampled_fwd_bwd'] + bpl['necessary_k_seg_reduce']
        step_occ = bpl['k_sampled_fwd_bwd'] + bpl['k_seg_reduce']
        step_req = bpl['requested_k_sampled_fwd_bwd'] + bpl['requested_k_seg_reduce']
        step_traffic = sum(traffic_of.get(kn, 0.0) for kn in (FWD, RED, 'k_span_planned')) or None
        # the step's cache-level bound: its kernels' bounds weighted by the time each would need for its requested rows
        step_bound = None
        if all(per_kernel[kn_]['cache_bound_GBs'] for kn_ in per_kernel):
            step_bound = step_req / sum(pk_['requested_bytes'] / pk_['cache_bound_GBs'] for pk_ in per_kernel.values())
        dk = per_kernel[dom]
        out['roofline'] = {
            'bound': 'hbm', 'kernel': dom, 'achieved': dk['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': dk['frac'],
            'traffic': traffic, 'traffic_source': traffic_note, 'kernel_source_hash': kernel_hash,
            'traffic_rate': (traffic / (dom_ms * 1e-3) / 1e9) if traffic else None,
            'traffic_frac': (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            'bytes_per_launch': dk['bytes_per_launch'], 'avg_launch_ms': float(dom_ms), 'timed_launches': n_timed,
            'kernels': per_kernel, 'row_counts': dedup['per_batch_mean'], 'batches_counted': dedup['batches_counted'],
            'sole_toucher_w_rows_fused': dedup['sole_toucher_w_rows_fused'],
            'cache_bytes_k_seg_reduce': dedup['bytes_per_launch']['cache_bytes_k_seg_reduce'],
            'cache_bytes_k_sampled_fwd_bwd': dedup['bytes_per_launch']['cache_bytes_k_sampled_fwd_bwd'],
            'whole_step_bytes': step_bytes, 'whole_step_achieved': step_bytes / step_s / 1e9,
            'whole_step_frac': step_bytes / step_s / 1e9 / HBM_PEAK_GBS,
            'whole_step_per_occurrence_frac': step_occ / step_s / 1e9 / HBM_PEAK_GBS,
            'cache_resident': bpl['cache_resident'],
            'cache_level': {'requested_bytes_per_step': step_req, 'requested_GBs': step_req / step_s / 1e9,
                            'bound_GBs': step_bound, 'frac_of_bound': (step_req / step_s / 1e9 / step_bound) if step_bound else None,
                            'note': 'rows the kernels request from L2 / Infinity Cache (one per occurrence, one gradient row per touch) over the '
                                    'whole step; for a cache-resident model (MovieLens shapes) this, not HBM, is the binding traffic.  ONE bound '
                                    'per kernel: L2 hit rate (PMC, same kernel sources) x 17 TB/s + (1 - hit) x 8.6 TB/s (MI355X_MICROARCH.md '
                                    '"Indexed rows"); the step\'s = its kernels\' weighted by the time their rows need; null without a profile'},
            'whole_step_traffic': step_traffic,
            'whole_step_traffic_frac': (step_traffic / step_s / 1e9 / HBM_PEAK_GBS) if step_traffic else None,
            # (SURVEY 8d's per-occurrence read-modify-write model: the kernels merge occurrences first, so it is not a fraction — r01 - r04
            # printed it beside the headline; kept for comparison with those rounds only)
            'legacy': {'model_bytes_per_launch': dom_alg, 'model_frac': achieved / HBM_PEAK_GBS,
                       'model_whole_step_frac': step_alg / step_s / 1e9 / HBM_PEAK_GBS},
            'hbm_copy_achievable': copy_gbs,
            'definition': 'frac = STRICTLY NECESSARY HBM bytes (bench.py:byte_model: every gathered row once per DISTINCT row, dz1/g2 written once, one '
                          'read-modify-write of parameter + slots per DISTINCT row, 8 B per touch; everything re-read assumed cached) / HIP-event '
                          'launch time / 8 TB/s — never above traffic_frac; per_occurrence_* = one gather read per occurrence where the table '
                          'exceeds the 256 MiB Infinity Cache (the r01-r03 headline); traffic = PMC bytes per launch (FETCH_SIZE x2 + WRITE_SIZE; '
                          'counts Infinity-Cache hits); requested_* = cache-level rows against cache_bound_GBs (PMC L2 hit rate); measured_reference = '
                          'the bytes at the rates scripts/mb/*.hip measured on this chip (gather 7.6 TB/s, cold read-modify-write 4.8 TB/s, no overlap)'}
    elif out.get('roofline') is None:
        out['roofline'] = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': None,
                           'traffic': None, 'model_bytes_per_launch': dom_alg, 'model_frac': achieved / HBM_PEAK_GBS,
                           'avg_launch_ms': float(dom_ms), 'timed_launches': n_timed, 'hbm_copy_achievable': copy_gbs,
                           'definition': 'row-sharded path: only the per-occurrence model (SURVEY 8d) is evaluated here'}
    out['cpu_baseline'] = out['cpu_baseline_all_cores'] = None
    if world == 1 and not args.no_cpu_baseline:
        from bench_cpu import cpu_baseline
        uid, iid, y, keep_off, seed = batches[0]
        one, many = cpu_baseline(eng, indptr, indices, (uid, iid, y, keep_off), seed, Q, LR, REG, q_threshold, budget_s=args.cpu_budget_s,
                                 n_cpu=args.cpu_triples, optimizer=args.optimizer, all_cores=not args.no_all_cores)
        out['cpu_baseline'], out['cpu_baseline_all_cores'] = one, many
    return out


def worker_main(args):
    """One rank of a measurement (world 1: the whole bench; world > 1: one layout, in a child process of a coordinator)."""
    global K
    if args.k:
        K = args.k
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    debug_gloo = os.environ.get('DRX_BENCH_BACKEND') == 'gloo'     # debugging aid: N ranks sharing ONE GPU, host-staged exchange
    dev_index = local_rank % max(1, torch.cuda.device_count()) if debug_gloo else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    import datetime
    import torch.distributed as dist
    rccl1 = world == 1 and os.environ.get('DRX_BENCH_RCCL1') == '1'     # debugging aid: 1-rank RCCL communicator, real collectives
    tmo = datetime.timedelta(seconds=300)
    if rccl1:
        _init_group(dist, 'nccl', 0, 1, device_id=dev, timeout=tmo)
    if world > 1:
        if debug_gloo:
            _init_group(dist, 'gloo', rank, world, timeout=tmo)
        else:
            _init_group(dist, 'nccl', rank, world, device_id=dev, timeout=tmo)
    layout = args.child_layout or ('columns' if args.force_columns else 'rows')
    if layout == 'columns' and (world > 1 or args.force_columns):
        out = run_columns(args, rank, world, dev, dist, debug_gloo, rccl1)
    else:
        out = run_direct(args, rank, world, dev, dist, debug_gloo, rccl1)
    if rank == 0 and world == 1 and not args.force_columns and not args.force_sharded:
        out['hr_at_10'] = hr_at_10(dev, with_cpu=not args.no_cpu_baseline) if (not args.no_hr and not args.users) else None
        out['cpu_baseline_reference_mode'] = (out['hr_at_10'] or {}).get('cpu_baseline_reference_mode')
        if args.force_configs or (not args.no_configs and not args.users and args.workload == 'synth-10m' and K == 128):
            from bench_configs import configs_block
            out['configs'] = configs_block(dev, run_direct, args, with_cpu=not args.no_cpu_baseline)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or rccl1:
        dist.barrier()                      # rank 0 prints (and times a device copy) after the others are done
        dist.destroy_process_group()
    return 0


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.child_layout == 'selftest':
        selftest_child()
        return 0
    if args.child_layout is None and (args.gpus > 1 or args.launch_dry_run):
        # N GPUs: this process only starts children and merges their lines — it must not touch the GPU (and does not import anything
        # that does): the children are fresh processes, never an exec of one that has initialised HIP
        return launch_or_coordinate(args, argv)
    if args.child_layout is None:
        assert int(os.environ.get('WORLD_SIZE', 1)) == 1, 'WORLD_SIZE > 1 needs --gpus N'
    return worker_main(args)


if __name__ == '__main__':
    sys.exit(main())
