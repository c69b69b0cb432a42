#!/bin/bash
# Per-kernel average / minimum duration of one bench run under rocprofv3 --kernel-trace --stats (timed steps overlap
# their streams, so the averages include contention; the minimum is close to the isolated time).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs "$@" > /dev/null 2>/tmp/kt.err
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
print("%-16s %6s %12s %12s %7s" % ("kernel", "calls", "avg us", "min us", "%"))
for r in csv.DictReader(open(f)):
    name = r["Name"].split("(")[0].replace("void ", "").split("::")[-1]
    if name.startswith("k_"):
        print("%-16s %6s %12.1f %12.1f %7s" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Percentage"]))
PY
