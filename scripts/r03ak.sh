#!/bin/bash
OUT=gpurun_out/r03ak; mkdir -p $OUT
for rep in 1 2; do for q in 4 8; do for n in 2 3; do
  GPU_MAX_HW_QUEUES=$q DRX_SIDE_STREAMS=$n python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_q${q}_n${n}_$rep.json 2>> $OUT/bench.err
done; done; done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
