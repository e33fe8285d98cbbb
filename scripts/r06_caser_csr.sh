#!/bin/bash
set -u
TAG=${1:-r06bg}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_caser.py tests/test_gpu_baseline_shapes.py -q -m gpu -x > $OUT/pytest.log 2>&1; grep -n "passed\|failed\|Error" $OUT/pytest.log | tail -4
for i in 1 2; do timeout -k 5 600 python scripts/caser_fit_rate.py > $OUT/caser_fit_rate.json 2> $OUT/caser_fit_rate.err; python - <<PY
import json
d = json.loads([l for l in open('$OUT/caser_fit_rate.json') if l.startswith('{')][-1])
for k, v in d.items():
    if isinstance(v, dict) and ('fit_steady_ms_per_step' in v or 'step_ms' in v):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ('step_ms', 'fit_steady_ms_per_step', 'fit_windows_per_s')})
PY
done
python scripts/r06_caser_host_profile.py 2>&1 | tail -50 > $OUT/caser_host_profile.txt; head -14 $OUT/caser_host_profile.txt | cut -c1-150
