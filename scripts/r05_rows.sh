#!/bin/bash
# row layout at world 1 through a 1-rank RCCL communicator: parity tests, then the bench lines (own rows bypassing / every row through the communicator)
TAG=${1:-r05rows}
mkdir -p gpurun_out/$TAG
timeout -k 5 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/$TAG/tests.log
for extra in "" "--no-self-bypass"; do
  DRX_BENCH_RCCL1=1 python bench.py --force-sharded --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs $extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('rows world1 [$extra]', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k[:14]: round(x,3) for k,x in d['phases_ms'].items()}, 'host_issue', d.get('host_issue_ms_per_step'))
json.dump(d, open('gpurun_out/$TAG/rows_world1$extra.json','w'))"
done
