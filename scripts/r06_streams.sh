#!/bin/bash
set -u
TAG=${1:-r06am}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
{
echo "# three models in one process (each fit takes its run-ahead stream from the probed process-wide pool)"
python scripts/r06_dmf_dev256.py DMF,ModifiedDMF,DMF,ModifiedDMF 256 2>&1 | grep steady | cut -c1-60
echo "# K dummy high-priority streams used first, then the model"
for K in 2 6 10; do python scripts/r06_stream_parity.py $K -1 use 2>&1 | tail -3 | cut -c1-200; done
echo "# the same with the probe off (DRX_STREAM_PROBE=0)"
for K in 6 10; do DRX_STREAM_PROBE=0 python scripts/r06_stream_parity.py $K -1 use 2>&1 | tail -1 | cut -c1-200; done
} > $OUT/streams.log 2>&1
cat $OUT/streams.log
timeout -k 5 1500 python -m pytest tests/test_gpu_caser.py tests/test_gpu_dmf.py tests/test_gpu_shard.py tests/test_gpu_fit.py -q -m gpu -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout -k 5 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1])
print('headline', round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), 'ms; frac', round(d['roofline']['frac'], 3))
c = d.get('configs', {})
for k in ('cfg3_dmf_ml1m', 'cfg5_caser_ml1m'):
    for n, v in c.get(k, {}).items():
        if isinstance(v, dict) and 'fit_steady_ms_per_step' in v:
            print(k, n, round(v['fit_steady_ms_per_step'], 4), 'ms/step steady', round(v.get('step_ms', 0), 4))
print('cfg2', round(c['cfg2_cdae_ml1m_sampled']['value'] / 1e6, 1), 'M/s')
PY
