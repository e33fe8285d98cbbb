#!/bin/bash
# r06 item 1(b): the chunked exchange schedule — parity of the shard kernels (pytest) and the all-remote one-GPU proxy at 1 / 2 / 4 / 8 chunks.
# Usage: gpurun -- bash scripts/r06_chunks.sh <tag> [pytest-selection]
set -u
TAG=${1:-r06b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 5 1200 python -m pytest tests/test_gpu_shard.py ${2:-} -x -q -m gpu > $OUT/pytest_shard.log 2>&1
tail -15 $OUT/pytest_shard.log
COMMON="--steps 100 --warmup 10 --windows 3 --no-cpu-baseline --no-hr --no-configs"
for C in 1 2 4 8; do
  DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --no-self-bypass --chunks $C > $OUT/remote_c$C.json 2> $OUT/remote_c$C.err
done
DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --no-self-bypass --chunks 1 --transport torch > $OUT/remote_c1_torch.json 2> $OUT/remote_c1_torch.err
DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --no-self-bypass --chunks 4 --transport torch > $OUT/remote_c4_torch.json 2> $OUT/remote_c4_torch.err
DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --chunks 4 > $OUT/bypass_c4.json 2> $OUT/bypass_c4.err
DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --chunks 1 > $OUT/bypass_c1.json 2> $OUT/bypass_c1.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), 'ms', d.get('host_issue_ms_per_step'), [round(v, 4) for v in (d.get('phases_ms') or {}).values()])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-1500:])
PY
