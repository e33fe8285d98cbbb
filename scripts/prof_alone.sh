export TMPDIR=/tmp; R=$(pwd); cd /tmp; rm -rf /tmp/pa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o kt -- python3 $R/scripts/exp_events.py none > /tmp/pa.txt 2>&1
tail -2 /tmp/pa.txt
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pa/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_sampled_fwd" in r["Name"] or "k_seg_reduce" in r["Name"] or "tail" in r["Name"]:
        print(r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3)
PY
