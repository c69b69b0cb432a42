export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/ktb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktb -o r -- python3 tools/ba_latency.py > /tmp/ba.out 2>&1
tail -3 /tmp/ba.out
python3 - <<EOF
import csv,glob
f=glob.glob("/tmp/ktb/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-40s %6s %10.1f us avg %10.1f total ms %6s%%" % (r["Name"].split("(")[0][:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
EOF
