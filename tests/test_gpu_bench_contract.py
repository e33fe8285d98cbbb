"""The one JSON line bench.py prints (the driver's contract): fields, types and internal consistency, on a short run of the small
workload (a child process; the 10M x 1M default is what the driver itself times)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contracted_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'ml-100k', '--batch', '4096', '--steps', '20',
                          '--warmup', '5', '--no-hr', '--force-configs', '--cpu-budget-s', '3'], cwd=ROOT, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['metric'].startswith('training samples/sec') and d['unit'] == 'samples/s' and d['higher_is_better'] is True
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5 and d['scaling'] == 'weak' and d['data'] == 'synthetic'
    assert d['vs_baseline'] is None and d['dtype'] == 'f32' and 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 4096 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6          # whole-job samples over the timed steps
    # five back-to-back windows of exactly `steps` steps; the line's numbers are the median window's
    assert d['windows'] == 5 and len(d['window_ms']) == 5 and d['window_ms_min'] <= d['ms_per_step'] * 20 * (1 + 1e-9) <= d['window_ms_max'] * (1 + 1e-6)
    assert abs(sorted(d['window_ms'])[2] - d['ms_per_step'] * 20) < 1e-3 and d['rccl_ranks'] == 1
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert 0.0 < r['frac'] <= 1.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert r['traffic'] is None or r['traffic'] > 0                                       # (PMC passes are of the 10M x 1M workload)
    # (MovieLens shapes: the lists are in the shared form — DRX_BATCH_SHARE_USERS — and the forward kernel is k_items_fwd_bwd)
    assert 'k_items_fwd_bwd' in r['kernels'] and r['row_counts']['share_touches'] < r['row_counts']['occ_W']
    for k in ('k_items_fwd_bwd', 'k_seg_reduce_planned'):
        kr = r['kernels'][k]
        assert 0.0 < kr['frac'] <= 1.0 and kr['avg_launch_ms'] > 0
        # the headline fraction is priced on the strictly necessary bytes: never above the per-occurrence reading, never above what
        # the counters support, never above the rows the kernel requests at cache level
        assert kr['frac'] <= kr['per_occurrence_frac'] * (1 + 1e-9) and kr['bytes_per_launch'] <= kr['requested_bytes']
        assert kr['traffic_frac'] is None or kr['frac'] <= kr['traffic_frac'] * 1.05
        # ONE cache-level bound per kernel, built from the L2 hit rate the counters measured on these kernel sources (none for this
        # small workload: null, never a guess); where it exists the requested rows stay under it
        assert kr['requested_GBs'] > 0 and (kr['cache_bound_GBs'] is None) == (kr['l2_hit_rate'] is None)
        assert kr['requested_frac_of_cache_bound'] is None or 0.0 < kr['requested_frac_of_cache_bound'] <= 1.0
        assert kr['measured_reference']['reference_ms'] > 0
    # the reduction's strictly necessary bytes are what the formula says, recomputed from the line's own row counts:
    # one read-modify-write of parameter + slot (4 x 4K bytes) per distinct row that reaches the list, 8 bytes per touch
    rc, Kb = r['row_counts'], 4.0 * 128
    rows = (rc['dist_W'] - 0) + (rc['dist_V'] - rc['solo_V']) + (rc['dist_O'] - rc['solo_O'])
    touches = rc.get('share_touches', rc['occ_W']) + (rc['B'] - rc['solo_V']) + (rc['B'] - rc['solo_O'])
    want = 4.0 * Kb * rows + 8.0 * touches
    assert abs(r["kernels"]["k_seg_reduce_planned"]["bytes_per_launch"] - want) <= 1e-4 * want, (r['kernels']['k_seg_reduce_planned']['bytes_per_launch'], want)
    assert r['whole_step_frac'] <= r['whole_step_per_occurrence_frac'] * (1 + 1e-9)
    assert r['cache_resident'] is True and r['cache_level']['requested_GBs'] > r['whole_step_achieved']      # ml-100k shape: all in cache
    assert r['cache_level']['frac_of_bound'] is None or r['cache_level']['frac_of_bound'] <= 1.0
    # the per-occurrence read-modify-write model of SURVEY 8d is not a fraction: under `legacy`, nowhere near the headline
    assert 'model_frac' not in r and 'model_whole_step_frac' not in r and 'model_frac' in r['legacy']
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['value'] > 0 and c['unit'] == 'samples/s' and c['sample']
    a = d['cpu_baseline_all_cores']
    assert a['kind'] == 'port' and a['cores'] >= 1 and a['value'] > 0 and a['unit'] == 'samples/s' and 'worker processes' in a['sample']
    # BASELINE configurations 2, 3 and 5 in the same line
    g = d['configs']
    c2 = g['cfg2_cdae_ml1m_sampled']
    assert c2['value'] > 0 and c2['unit'] == 'samples/s' and 0.0 < c2['roofline']['frac'] <= 1.0 and c2['cpu_baseline']['value'] > 0
    for k in ('k_items_fwd_bwd', 'k_seg_reduce_planned'):
        assert 0.0 < c2['roofline']['kernels'][k]['frac'] <= 1.0
    # the ml-1m model is cache-resident: its line carries the cache-level reading (rows requested from L2 / Infinity Cache)
    assert c2['roofline']['cache_resident'] is True and c2['roofline']['cache_level']['requested_GBs'] > 0
    for k in ('k_items_fwd_bwd', 'k_seg_reduce_planned'):
        f = c2['roofline']['kernels'][k]['requested_frac_of_cache_bound']        # (quoted only with a PMC profile of these kernel sources)
        assert f is None or 0.0 < f <= 1.0, (k, f)
    fb = c2['roofline']['cache_level']['frac_of_bound']
    assert fb is None or 0.0 < fb <= 1.0
    c3 = g['cfg3_dmf_ml1m']
    for name in ('DMF_B256', 'DMF_B4096', 'ModifiedDMF_B256', 'ModifiedDMF_B4096'):
        assert c3[name]['step_ms'] > 0 and c3[name]['fit_samples_per_s'] > 0
    m = c3['mfma_scorer']
    assert m['users'] == 2048 and m['achieved_write_GBs'] > 0 and 0.0 < m['frac_of_hbm_peak'] <= 1.0
    c5 = g['cfg5_caser_ml1m']['Caser_B4096']
    assert c5['step_ms'] > 0 and c5['fit_windows_per_s'] > 0 and c5['fit_steady_ms_per_step'] > 0
    # a steady fit() rate is the median of fenced windows inside ONE long fit: it cannot undercut the device step by more than noise
    for blk, names_ in ((c3, ('DMF_B4096', 'ModifiedDMF_B4096')), (g['cfg5_caser_ml1m'], ('Caser_B4096',))):
        for name in names_:
            assert blk[name]['fit_windows']['windows'] >= 5 and blk[name]['fit_steady_ms_per_step'] >= 0.8 * blk[name]['step_ms'], (name, blk[name])
    # r06: configurations 3 and 5 carry a roofline (necessary HBM bytes AND the cache-level reading: their tables are L2-resident) and a
    # cpu_baseline (the oracle's step on the host, one core) like configuration 2
    for blk, keys, unit in ((c3, ('B256', 'B4096'), 'samples/s'), (g['cfg5_caser_ml1m'], ('B4096',), 'windows/s')):
        for k in keys:
            rl = blk['roofline'][k]
            assert rl['bound'] == 'hbm' and rl['peak'] == 8000.0 and 0.0 < rl['frac'] <= 1.0 and abs(rl['frac'] - rl['achieved'] / rl['peak']) < 1e-9
            cl = rl['cache_level']
            assert cl['requested_bytes_per_step'] > rl['bytes_per_launch'] * 0 and 0.0 < cl['frac_of_bound'] <= 1.0 and cl['bound_GBs'] == 17000.0
            assert rl['avg_launch_ms'] >= cl['time_at_bound_ms'] > 0          # (no step is faster than its bytes at the two bounds)
        cb = blk['cpu_baseline']
        assert cb['kind'] == 'port' and cb['cores'] == 1 and cb['value'] > 0 and cb['unit'] == unit and cb['sample']
    c5d = g['cfg5_caser_ml1m']['Caser_B4096_device_sampler']
    assert c5d['sampler'].startswith('device') and 0 < c5d['fit_steady_ms_per_step'] < c5['fit_steady_ms_per_step']
