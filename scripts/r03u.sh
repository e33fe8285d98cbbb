#!/bin/bash
# after the sampler-scratch fix: A/B of the forward kernels, the pipelined multi-process tests repeated WITHOUT the retry, full GPU suite
OUT=gpurun_out/r03u; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do for pf in 0 1; do
  DRX_FWD_PF=$pf python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hr --no-configs > $OUT/bench_pf${pf}_$rep.json 2>> $OUT/bench.err
done; done
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M/s', round(d['ms_per_step'], 4), [round(v * 1e3, 1) for v in d['phases_ms'].values()])
PY
fails=0; runs=0
for rep in $(seq 1 ${REPS:-6}); do
  DRX_TEST_NO_RETRY=1 timeout -k 5 400 python -m pytest tests/test_gpu_kshard.py -q -m gpu -k "pipelines_of_several or in_turns or in_parts" -p no:cacheprovider > $OUT/flake_$rep.log 2>&1 || fails=$((fails+1))
  runs=$((runs+1)); tail -1 $OUT/flake_$rep.log
done
echo "multi-process repetitions without retry: $fails failing of $runs"
timeout -k 5 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/gpu_tests.log 2>&1; echo "full suite rc=$?"; tail -2 $OUT/gpu_tests.log
