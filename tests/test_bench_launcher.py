"""bench.py --gpus N from a plain command line: the starting process launches the ranks itself — one fresh child
per rank and sharding layout — without touching the GPU, relays ONE merged JSON line and returns the children's verdict.  Checked
here on the CPU: the dry run's command lines, and the whole launcher path over gloo with the children's measurement replaced by a
rendezvous + all-reduce (`--launch-selftest`), both started plainly and under torch.distributed.run (the driver's way)."""
import json
import os
import pytest
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')
ENV = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}


def test_dry_run_prints_one_command_line_per_rank_and_layout():
    out = subprocess.run([sys.executable, BENCH, '--gpus', '4', '--steps', '20', '--warmup', '5', '--launch-dry-run'], cwd=ROOT, env=ENV,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('[')]
    assert len(lines) == 12                                       # 4 ranks x (rows, its second chance, columns)
    retry = [l for l in lines if l.startswith('[rows_torch_transport (only if rows fails)]')]
    assert len(retry) == 4 and all(l.rstrip().endswith('--transport torch --child-layout rows') for l in retry)
    for lay in ('rows', 'columns'):
        mine = [l for l in lines if l.startswith(f'[{lay}]')]
        assert sorted(int(l.split('RANK=')[1].split()[0]) for l in mine) == [0, 1, 2, 3]
        stores = {l.split('DRX_RDZV=')[1].split()[0] for l in mine}
        assert len(stores) == 1 and next(iter(stores)).startswith('file://')    # the ranks of a layout meet in one FILE store ...
        for l in mine:
            assert 'WORLD_SIZE=4' in l and 'MASTER_PORT' not in l and l.rstrip().endswith(f'--child-layout {lay}')
            assert '--gpus 4 --steps 20 --warmup 5' in l and '--launch-dry-run' not in l
    assert len({l.split('DRX_RDZV=')[1].split()[0] for l in lines}) == 3         # ... and every attempt in a store of its own


def _one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_launcher_path_over_gloo_world_2():
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--launch-selftest'], cwd=ROOT, env=ENV, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _one_line(out.stdout)
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['selftest'] is True and 'selftest' in d['layouts']


def test_launcher_path_under_torch_distributed_run():
    """The driver's command shape: every worker coordinates its own rank; rank 0 prints the one line."""
    import socket
    with socket.socket() as so:               # torch.distributed.run itself wants a port: one the kernel just handed out, not a formula
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), BENCH, '--gpus', '2', '--launch-selftest'], cwd=ROOT, env=ENV, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _one_line(out.stdout)
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2


def test_a_failing_layout_is_reported_not_fatal(tmp_path):
    """A layout whose children die leaves an `error` entry; the line and the exit code come from the layouts that survived — here
    none does (no GPU in the CPU test box: the real children raise), so the launcher reports the failure and returns non-zero."""
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1', '--layout', 'columns',
                          '--layout-timeout-s', '120'], cwd=ROOT, env=dict(ENV, HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES=''),
                         capture_output=True, text=True, timeout=400)
    d = _one_line(out.stdout)
    assert out.returncode != 0 and d['value'] is None and 'error' in d['layouts']['columns']


def test_a_failing_row_layout_gets_a_second_chance_through_torch_distributed():
    """The row layout's exchanges go through the library's own RCCL communicator by default; when that attempt fails on ANY rank every
    coordinator starts the layout once more with --transport torch (here both die: no GPU) — and an explicit --transport torch has no
    second attempt."""
    env = dict(ENV, HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1', '--layout', 'rows', '--layout-timeout-s', '120'],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    d = _one_line(out.stdout)
    assert out.returncode != 0 and d['value'] is None and set(d['layouts']) == {'rows', 'rows_torch_transport'}, d
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1', '--layout', 'rows', '--transport', 'torch',
                          '--layout-timeout-s', '120'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert set(_one_line(out.stdout)['layouts']) == {'rows'}


def test_row_statistics_of_the_shared_form_match_a_brute_force_count():
    """bench.batch_row_stats(share=True) — the byte model's touches / row loads / work items of DRX_BATCH_SHARE_USERS lists — against a
    plain-Python count of the rule of csrc/drx_prep.hpp (k_tp_item_* / k_tp_expand<SHARE>): the samples of a user, ascending, in work
    items of 16; per (work item, history position) one touch of the item's summed row + one per dropper where 1 + droppers < keepers,
    else one per keeper."""
    import numpy as np
    import torch
    import bench
    from helpers import hash_u32, q_threshold
    rng = np.random.default_rng(3)
    U, N, B, q, seed = 9, 40, 300, 0.3, 77
    deg = rng.integers(1, 12, size=U)
    indptr = np.zeros(U + 1, np.int64)
    indptr[1:] = np.cumsum(deg)
    indices = np.concatenate([np.sort(rng.choice(N, size=d, replace=False)) for d in deg]).astype(np.int32)
    uid = rng.integers(0, U, size=B)
    uid[:60] = 4                                                   # one user with several work items
    iid = rng.integers(0, N, size=B)
    keep_off = np.zeros(B + 1, np.int64)
    keep_off[1:] = np.cumsum(deg[uid])
    st = bench.batch_row_stats(torch.as_tensor(indptr), torch.as_tensor(indices), torch.as_tensor(uid, dtype=torch.int32),
                               torch.as_tensor(iid, dtype=torch.int32), torch.as_tensor(keep_off, dtype=torch.int32), seed, q, share=True)
    touches = item_rows = items = 0
    thr = q_threshold(q)
    for u in range(U):
        samples = [b for b in range(B) if uid[b] == u]
        for p0 in range(0, len(samples), bench.SHARE_TRIPLES):
            piece = samples[p0:p0 + bench.SHARE_TRIPLES]
            items += 1
            for j in range(deg[u]):
                k = sum(int(hash_u32(seed, np.asarray([b]), np.asarray([j]))[0] >= thr) for b in piece)
                item_rows += 1
                touches += (1 + len(piece) - k) if (len(piece) > 1 and 1 + len(piece) - k < k) else k
    assert (st['share_touches'], st['share_item_rows'], st['share_items']) == (touches, item_rows, items)
    assert st['share_touches'] <= st['occ_W']                      # never more touches than the plain list holds
    bm = bench.byte_model({k_: float(v) for k_, v in st.items()}, 128, 1.0, fused_solo=True, n_users=U, n_items=N)
    assert bm['necessary_k_sampled_fwd_bwd'] <= bm['requested_k_sampled_fwd_bwd'] and bm['necessary_k_seg_reduce'] <= bm['requested_k_seg_reduce']


def test_byte_model_against_hand_computed_bytes():
    """bench.byte_model — the function every roofline fraction of the line is priced on — on a batch small enough to count by hand:
    B = 10 triples, K = 128 (rows of 512 B), Adagrad (S = 1), tables larger than the Infinity Cache; 40 history items of which 30 are
    kept (22 distinct, 16 of them touched once); 9 distinct users (8 sole), 7 distinct output items (5 sole)."""
    import bench
    st = {'B': 10, 'history_items': 40, 'occ_W': 30, 'dist_W': 22, 'solo_W': 16, 'dist_V': 9, 'solo_V': 8, 'dist_O': 7, 'solo_O': 5}
    bm = bench.byte_model(st, 128, 1.0, fused_solo=True)
    row = 512.0
    # reduction: one read-modify-write of parameter + slot (4 rows' worth) per distinct row that reaches the list — every distinct W row,
    # the V / W2T rows more than one triple touches — and 8 bytes per touch (kept W occurrences + the non-sole V / W2T touches)
    rows = 22 + (9 - 8) + (7 - 5)
    touches = 30 + (10 - 8) + (10 - 5)
    assert bm['k_seg_reduce'] == bm['necessary_k_seg_reduce'] == row * rows * 4 + 8.0 * touches
    assert bm['cache_bytes_k_seg_reduce'] == row * touches and bm['requested_k_seg_reduce'] == bm['k_seg_reduce'] + row * touches
    # forward, per-occurrence reading: one gathered row per kept occurrence + V + W2T row per triple, dz1 per triple and g2 where the
    # output row is shared, sole-toucher V / W2T rows updated in place (1 slot read + parameter and slot written), indices, 40 B ids
    fwd = row * (30 + 10 + 10) + row * (10 + (10 - 5)) + row * (8 + 5) * 3 + 4.0 * 40 + 40.0 * 10
    assert bm['k_sampled_fwd_bwd'] == fwd
    # strictly necessary: every gathered row once per DISTINCT row
    assert bm['necessary_k_sampled_fwd_bwd'] == fwd - row * ((30 + 10 + 10) - (22 + 9 + 7))
    assert bm['requested_k_sampled_fwd_bwd'] == fwd and bm['cache_resident'] is False
    # a model that fits the Infinity Cache (MovieLens shapes): gathers are counted once per distinct row in the first place
    small = bench.byte_model(st, 128, 1.0, fused_solo=True, n_users=1000, n_items=1000)
    assert small['cache_resident'] is True and small['k_sampled_fwd_bwd'] == small['necessary_k_sampled_fwd_bwd']
    # Adam keeps two slots per parameter: (2 + 2 S) rows per read-modify-write
    assert bench.byte_model(st, 128, 2.0, fused_solo=True)['k_seg_reduce'] == row * rows * 6 + 8.0 * touches
