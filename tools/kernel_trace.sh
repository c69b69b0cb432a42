#!/bin/bash
# Per-launch kernel durations (median / min per kernel and grid size) from a rocprofv3 kernel trace of a short bench run;
# shows e.g. the seven resize levels separately.  Run on the GPU box from the repository root.
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-match --batch 256 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    k = row['Kernel_Name'].split('(')[0].replace('void ', '')
    if k.startswith('k_'):
        acc[(k, row['Grid_Size_X'] if 'Grid_Size_X' in row else row.get('Grid_Size'))].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
for (k, g), v in sorted(acc.items()):
    v = sorted(v)
    print("%-20s grid %-10s n=%3d  median %8.1f us  min %8.1f" % (k, g, len(v), v[len(v)//2] / 1e3, v[0] / 1e3))
PY
