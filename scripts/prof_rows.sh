# kernel stats of the row-sharded layout at world 1 through a 1-rank RCCL communicator:  bash scripts/prof_rows.sh <tag> [extra bench flags]
TAG=${1:-rows}; shift
export TMPDIR=/tmp; R=$(pwd); mkdir -p $R/gpurun_out/$TAG; cd /tmp; rm -rf /tmp/pr_$TAG
export DRX_BENCH_RCCL1=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$TAG -o kt -- python3 $R/bench.py --force-sharded --steps 60 --warmup 10 --windows 1 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/prof_bench.json 2> $R/gpurun_out/$TAG/prof_bench.err
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pr_$TAG/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
out=open("$R/gpurun_out/$TAG/kernel_stats.csv","w")
out.write("Name,Calls,AverageUs,TotalMs\n")
for r in rows[:40]:
    line="%s,%s,%.1f,%.2f"%(r["Name"][:110].replace(",",";"), r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6)
    out.write(line+"\n"); print(line)
PY
