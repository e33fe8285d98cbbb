#!/bin/bash
OUT=gpurun_out/r03aj; mkdir -p $OUT
for rep in 1 2; do for n in 1 2; do
  DRX_SIDE_STREAMS=$n python bench.py --workload ml-1m --steps 30 --warmup 5 --windows 3 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_ml1m_n${n}_$rep.json 2>> $OUT/bench.err
  DRX_SIDE_STREAMS=$n python bench.py --workload ml-1m --batch 4096 --steps 100 --warmup 10 --windows 3 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_ml1mB4096_n${n}_$rep.json 2>> $OUT/bench.err
done; done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
tail -2 $OUT/bench.err
