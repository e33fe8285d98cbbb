#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03c}
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kshard.py tests/test_gpu_cdae.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
( time python bench.py ) > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -4 $OUT/bench_full.err
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('value', round(d['value'] / 1e6, 1), 'ms', round(d['ms_per_step'], 4), 'win', d.get('window_ms'), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
print('frac', d['roofline']['frac'], 'whole', d['roofline']['whole_step_frac'])
print('cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None, 'all', (d.get('cpu_baseline_all_cores') or {}).get('value'), (d.get('cpu_baseline_all_cores') or {}).get('cores'))
print('configs', json.dumps(d.get('configs'), indent=0)[:3000])
PY
ROOT=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/kt -o kt -- python3 $ROOT/bench.py --steps 100 --warmup 10 --windows 2 --no-cpu-baseline --no-hr --no-configs > $ROOT/$OUT/kt_bench.json 2> $ROOT/$OUT/kt.err
cd $ROOT
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -size +4M -delete
head -12 $OUT/kernel_stats.csv
DRX_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --windows 2 --users 200000 --batch 8192 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo "2-rank rc=$?"; tail -3 $OUT/bench_2rank_gloo.err; cut -c1-1500 $OUT/bench_2rank_gloo.json
