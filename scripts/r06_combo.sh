#!/bin/bash
# one GPU-box call: placement A/B (+ lookup-only diagnostic), chunk sweep with the threaded transport, three tests, prep-alone kernel table
set -u
TAG=${1:-r06h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash scripts/r06_ab_place.sh $TAG/ab > $OUT/ab.log 2>&1; tail -8 $OUT/ab.log
timeout -k 5 900 python -m pytest "tests/test_gpu_fullsize.py::test_one_full_size_step_matches_the_oracle_on_compacted_tables" tests/test_gpu_kshard.py tests/test_gpu_shard.py -q -m gpu > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
COMMON="--steps 100 --warmup 10 --windows 3 --no-cpu-baseline --no-hr --no-configs"
for C in 1 2 4; do
  DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --no-self-bypass --chunks $C > $OUT/remote_c$C.json 2> $OUT/remote_c$C.err
done
DRX_BENCH_RCCL1=1 timeout -k 5 300 python bench.py $COMMON --force-sharded --chunks 2 > $OUT/bypass_c2.json 2> $OUT/bypass_c2.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), 'ms', d.get('host_issue_ms_per_step'), [round(v, 4) for v in (d.get('phases_ms') or {}).values()])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-1500:])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prep -o prep -- python3 $GRAFT_REPO_ROOT/scripts/prep_alone.py synth-10m 65536 > $GRAFT_REPO_ROOT/$OUT/prep_alone.txt 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $OUT/prep -name '*kernel_stats.csv' | head -1) $OUT/prep_alone_kernel_stats.csv 2>/dev/null
find $OUT/prep -name '*kernel_trace.csv' -delete
head -25 $OUT/prep_alone_kernel_stats.csv; tail -2 $OUT/prep_alone.txt
