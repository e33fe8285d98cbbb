"""bench.py --gpus N from a plain command line: the starting process launches the ranks itself — one fresh child
per rank and sharding layout — without touching the GPU, relays ONE merged JSON line and returns the children's verdict.  Checked
here on the CPU: the dry run's command lines, and the whole launcher path over gloo with the children's measurement replaced by a
rendezvous + all-reduce (`--launch-selftest`), both started plainly and under torch.distributed.run (the driver's way)."""
import json
import os
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
    assert len(lines) == 8                                        # 4 ranks x (rows, columns)
    for lay in ('rows', 'columns'):
        mine = [l for l in lines if l.startswith(f'[{lay}]')]
        assert sorted(int(l.split('RANK=')[1].split()[0]) for l in mine) == [0, 1, 2, 3]
        stores = {l.split('DRX_RDZV=')[1].split()[0] for l in mine}
        assert len(stores) == 1 and next(iter(stores)).startswith('file://')    # the ranks of a layout meet in one FILE store ...
        for l in mine:
            assert 'WORLD_SIZE=4' in l and 'MASTER_PORT' not in l and l.rstrip().endswith(f'--child-layout {lay}')
            assert '--gpus 4 --steps 20 --warmup 5' in l and '--launch-dry-run' not in l
    assert len({l.split('DRX_RDZV=')[1].split()[0] for l in lines}) == 2         # ... and the two layouts in different ones


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


def test_pmc_traffic_profile_was_taken_on_these_kernel_sources():
    """bench.py quotes profiles/pmc_traffic.json only when its kernel-source hash equals the tree's; a stale file means the driver's line
    carries `traffic: null` (it happened in r03: the passes were re-taken on the GPU box and only the per-round copy was committed).
    Re-run scripts/profile_round.sh and copy gpurun_out/<tag>/pmc_traffic.json to profiles/pmc_traffic.json when this fails."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    with open(os.path.join(root, 'profiles', 'pmc_traffic.json')) as f:
        meta = json.load(f)['_meta']
    assert meta['kernel_source_hash'] == bench.kernel_source_hash()
