#!/bin/bash
# one GPU-box call: the row layout's tests, then the one-GPU proxy with the exchanges issued by the library's phase calls against call by call
set -u
TAG=${1:-r06ac}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 5 1200 python -m pytest tests/test_gpu_shard.py -q -m gpu -x > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
COMMON="--steps 100 --warmup 10 --windows 3 --no-cpu-baseline --no-hr --no-configs"
for C in 1 2 4; do
  DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --no-self-bypass --chunks $C > $OUT/remote_c${C}_phases.json 2> $OUT/remote_c${C}_phases.err
  DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --no-self-bypass --chunks $C --no-phases > $OUT/remote_c${C}_python.json 2> $OUT/remote_c${C}_python.err
done
for C in 1 2; do
  DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --chunks $C > $OUT/bypass_c${C}_phases.json 2> $OUT/bypass_c${C}_phases.err
  DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --chunks $C --no-phases > $OUT/bypass_c${C}_python.json 2> $OUT/bypass_c${C}_python.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), 'ms', d.get('host_issue_ms_per_step'), d['config'].get('exchanges_issued_by'), [round(v, 4) for v in (d.get('phases_ms') or {}).values()])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-1500:])
PY
