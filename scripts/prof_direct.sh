# kernel stats of the single-GPU sampled step:  bash scripts/prof_direct.sh <tag> [bench flags, e.g. --workload ml-1m]
TAG=${1:-direct}; shift
export TMPDIR=/tmp; R=$(pwd); mkdir -p $R/gpurun_out/$TAG; cd /tmp; rm -rf /tmp/pd_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd_$TAG -o kt -- python3 $R/bench.py --steps 60 --warmup 10 --windows 1 --no-cpu-baseline --no-hr --no-configs "$@" > $R/gpurun_out/$TAG/prof_bench.json 2> $R/gpurun_out/$TAG/prof_bench.err
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pd_$TAG/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
out=open("$R/gpurun_out/$TAG/kernel_stats.csv","w")
out.write("Name,Calls,AverageUs,TotalMs\n")
for r in rows[:30]:
    line="%s,%s,%.1f,%.2f"%(r["Name"][:110].replace(",",";"), r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6)
    out.write(line+"\n"); print(line)
PY
