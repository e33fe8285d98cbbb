#!/bin/bash
# How much the phase events of bench.py cost: the step rate with events recorded on every 4th / 8th / 16th step.
for e in 4 8 16; do for i in 1 2; do
  DRX_BENCH_EVENTS_EVERY=$e python bench.py --no-hr --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(sys.argv[1], round(r['value']/1e6,2), r['roofline']['timed_launches'], round(r['roofline']['avg_launch_ms'],4))" $e
done; done
