#!/bin/bash
# Device timeline of ONE single-frame operator() call (the last of tools/latency.py's): start / end of every kernel and
# copy relative to the first, from a rocprofv3 kernel + memory-copy trace.  Run on the GPU box from the repository root.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o r -- python3 $GRAFT_REPO_ROOT/tools/latency.py "$@" > /tmp/tl.out 2>&1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        ev.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), row['Kernel_Name'].split('(')[0].replace('void ', ''), row.get('Queue_Id', '')))
for f in glob.glob('/tmp/tl/**/*memory_copy_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        ev.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), 'copy ' + row.get('Direction', ''), ''))
ev.sort()
# one call = from a host-to-device copy up to the next one; take the 40th call (steady state, stage timing still off)
starts = [i for i, e in enumerate(ev) if e[2].startswith('copy') and 'HOST_TO_DEVICE' in e[2].upper()]
i0 = starts[40]; i1 = starts[41]
t0 = ev[i0][0]
for s, e, n, q in ev[i0:i1]:
    print("%8.1f -> %8.1f us  (%6.1f)  %-18s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n, q))
PY
