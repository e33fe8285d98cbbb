#!/bin/bash
# One GPU-box call: bench line + kernel-trace stats + the two PMC passes (separate runs, per the microarch guide).
# Usage (through gpurun): bash scripts/profile_round.sh <tag>; afterwards, HERE: cp gpurun_out/<tag>/pmc_traffic*.json profiles/
# (the copy made below lands on the GPU box only; tests/test_bench_launcher.py checks the committed file against the kernel sources)
# Every rocprofv3 call runs under `timeout -k`: r03 lost 50 GPU-minutes to a profiler that kept waiting after its child had faulted.
set -u
TAG=${1:-r01f}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --steps 200 --warmup 10 --windows 2 --no-cpu-baseline --no-hr --no-configs > $OUT/kt_bench.json 2> $OUT/kt.err
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-hr --no-configs > /dev/null 2> $OUT/pmc_fetch.err
timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-hr --no-configs > /dev/null 2> $OUT/pmc_write.err
timeout -k 5 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o l2 -- python3 $ROOT/bench.py --steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-hr --no-configs > /dev/null 2> $OUT/pmc_l2.err
# the ml-1m-shaped sampled run of the `configs` block (BASELINE configuration 2): kernel stats of its own
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_ml1m -o kt -- python3 $ROOT/bench.py --workload ml-1m --steps 100 --warmup 10 --windows 2 --no-cpu-baseline --no-hr --no-configs > $OUT/kt_ml1m_bench.json 2> $OUT/kt_ml1m.err
cp $(find $OUT/kt_ml1m -name '*kernel_stats.csv' | head -1) $OUT/ml1m_sampled_kernel_stats.csv
# ... and its two PMC passes (`configs.cfg2...traffic` was null)
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_ml1m -o f -- python3 $ROOT/bench.py --workload ml-1m --steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-hr --no-configs > /dev/null 2> $OUT/pmc_fetch_ml1m.err
timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_ml1m -o w -- python3 $ROOT/bench.py --workload ml-1m --steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-hr --no-configs > /dev/null 2> $OUT/pmc_write_ml1m.err
timeout -k 5 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2_ml1m -o l2 -- python3 $ROOT/bench.py --workload ml-1m --steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-hr --no-configs > /dev/null 2> $OUT/pmc_l2_ml1m.err
cd $ROOT
find $OUT -name '*.csv' | head -20
python profiles/pmc_l2.py $(find $OUT/pmc_l2 -name '*counter_collection.csv' | head -1) $OUT/pmc_l2_hit_rate.json
python profiles/pmc_summary.py $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) $OUT/pmc_traffic_raw.json
python profiles/pmc_make_traffic.py $OUT/pmc_traffic_raw.json $OUT/pmc_traffic.json $TAG synth-10m 65536 adagrad $OUT/pmc_l2_hit_rate.json
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
python profiles/pmc_summary.py $(find $OUT/pmc_fetch_ml1m -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_write_ml1m -name '*counter_collection.csv' | head -1) $OUT/pmc_traffic_raw_ml1m.json
python profiles/pmc_l2.py $(find $OUT/pmc_l2_ml1m -name '*counter_collection.csv' | head -1) $OUT/pmc_l2_hit_rate_ml1m.json
python profiles/pmc_make_traffic.py $OUT/pmc_traffic_raw_ml1m.json $OUT/pmc_traffic_ml-1m.json $TAG ml-1m 65536 adagrad $OUT/pmc_l2_hit_rate_ml1m.json
cp $OUT/pmc_traffic_ml-1m.json profiles/pmc_traffic_ml-1m.json
python bench.py > $OUT/bench_with_traffic.json 2>> $OUT/bench.err     # the line as the driver will see it, quoting the PMC passes just taken
grep -h "drx::" $(find $OUT/kt -name '*kernel_stats.csv' | head -1) | head -40 > $OUT/kernel_stats_drx.csv
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
# keep only the small summaries
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -size +4M -delete
tail -5 $OUT/kt.err
du -sh $ROOT/gpurun_out
ls -la $OUT
