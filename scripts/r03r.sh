#!/bin/bash
OUT=gpurun_out/r03r; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD
python -m pytest tests/test_gpu_cdae.py tests/test_gpu_fullsize.py tests/test_gpu_fit.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_$rep.json 2>> $OUT/bench.err; done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
cd /tmp
stress() { name=$1; envs=$2; n=$3; f=0
  for i in $(seq 1 $n); do
    ( export $envs; timeout 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_s -o f -- python3 $ROOT/bench.py --steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-hr --no-configs > /dev/null 2> $ROOT/$OUT/s.err )
    if grep -q "Memory access fault" $ROOT/$OUT/s.err; then f=$((f+1)); fi
    rm -rf $ROOT/$OUT/pmc_s
  done; echo "$name: $f faults in $n runs"; }
stress default_plain X=0 12
