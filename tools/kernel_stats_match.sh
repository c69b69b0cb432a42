cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktm
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktm -o r -- python3 $GRAFT_REPO_ROOT/tools/match_latency.py > /tmp/ktm.out 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ktm/**/*kernel_stats.csv", recursive=True)[0]
print("%-28s %6s %10s %10s" % ("kernel", "calls", "avg us", "min us"))
for r in csv.DictReader(open(f)):
    name = r["Name"].split("(")[0].replace("void ", "")
    print("%-28s %6s %10.1f %10.1f" % (name[:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/ktm/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in ("k_bow_queries", "k_bow_resolve<0>", "k_bow_resolve<1>", "k_topk_lists_n<8>"):
    if k in d:
        print(k, "durations (us) in call order:", " ".join("%.0f" % x for x in d[k]))
PY
