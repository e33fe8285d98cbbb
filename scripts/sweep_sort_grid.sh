for g in 16 32 64 128 512; do
  echo "== DRX_SORT_GRID=$g"
  DRX_SORT_GRID=$g python scripts/bench_sort.py 2>&1 | head -2 | tail -1
  DRX_SORT_GRID=$g python bench.py --no-cpu-baseline --no-hr 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step'],4), {k[:12]:round(v,4) for k,v in d['phases_ms'].items()})"
done
