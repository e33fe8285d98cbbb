#!/bin/bash
# rows layout at world 1 through a 1-rank RCCL communicator: bench lines (own rows bypass the collectives / every row through the communicator),
# their kernel stats, and the full GPU suite a second time in this lease
mkdir -p gpurun_out/r04_rows
export DRX_BENCH_RCCL1=1
python bench.py --force-sharded --no-cpu-baseline --no-hr --no-configs > gpurun_out/r04_rows/rows_layout_world1_rccl.json 2> gpurun_out/r04_rows/rows1.err
python bench.py --force-sharded --no-self-bypass --no-cpu-baseline --no-hr --no-configs > gpurun_out/r04_rows/rows_layout_world1_rccl_all_remote.json 2> gpurun_out/r04_rows/rows2.err
unset DRX_BENCH_RCCL1
bash scripts/prof_rows.sh r04_rows_bypass > gpurun_out/r04_rows/prof_bypass.log 2>&1
bash scripts/prof_rows.sh r04_rows_allremote --no-self-bypass > gpurun_out/r04_rows/prof_allremote.log 2>&1
timeout -k 5 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r04_rows/gpu_tests_run2.log 2>&1; echo "suite rc=$?"; tail -1 gpurun_out/r04_rows/gpu_tests_run2.log
for f in gpurun_out/r04_rows/rows_layout_world1_rccl.json gpurun_out/r04_rows/rows_layout_world1_rccl_all_remote.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', round(d['value']/1e6,1), round(d['ms_per_step'],4), d['phases_ms'])"; done
