"""End-of-round gate (its own file, named to run LAST under -x): the committed PMC profiles belong to the tree's kernel sources."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('name', ['pmc_traffic.json', 'pmc_traffic_ml-1m.json'])
def test_pmc_traffic_profile_was_taken_on_these_kernel_sources(name):
    """bench.py quotes profiles/pmc_traffic*.json only when their kernel-source hash equals the tree's (bench.SAMPLED_STEP_SOURCES); a
    stale file means the driver's line carries `traffic: null`, `l2_hit_rate: null` and no cache bound (it happened in r03).  A FAILURE,
    not a skip: re-run scripts/profile_round.sh on the GPU box and copy gpurun_out/<tag>/pmc_traffic*.json to profiles/.
    The developer loop between two GPU-box calls sets DRX_ALLOW_STALE_PMC=1."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, 'profiles', name)) as f:
        meta = json.load(f)['_meta']
    if meta['kernel_source_hash'] != bench.kernel_source_hash():
        msg = ('profiles/%s is stale: taken on kernel sources %s, the tree is %s — re-run scripts/profile_round.sh'
               % (name, meta['kernel_source_hash'], bench.kernel_source_hash()))
        if os.environ.get('DRX_ALLOW_STALE_PMC') == '1':
            pytest.skip(msg)
        pytest.fail(msg)
