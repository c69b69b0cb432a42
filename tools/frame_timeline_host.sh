#!/bin/bash
# Host API calls and device activity of ONE single-frame operator() call on one time axis (rocprofv3 kernel, memory-copy and
# HIP runtime traces): when each launch was ISSUED against when its kernel ran.  GPU box, from the repository root:
#   bash tools/frame_timeline_host.sh W H N_FEATURES [name=value ...]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlh
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d /tmp/tlh -o r -- python3 $GRAFT_REPO_ROOT/tools/latency.py "$@" > /tmp/tlh.out 2>&1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob('/tmp/tlh/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        ev.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), 'GPU  ' + row['Kernel_Name'].split('(')[0].replace('void ', ''), row.get('Queue_Id', '')))
copies = []
for f in glob.glob('/tmp/tlh/**/*memory_copy_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        ev.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), 'GPU  copy ' + row.get('Direction', ''), ''))
for f in glob.glob('/tmp/tlh/**/*hip_api_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        ev.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), 'host ' + row['Function'], ''))
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2].startswith('GPU  copy') and 'HOST_TO_DEVICE' in e[2].upper()]
i0 = starts[40]; i1 = starts[41]
t0 = ev[i0][0]
# host calls that belong to the call start a little before the device copy: back up to the hipMemcpyAsync that issued it
j = i0
while j > 0 and not (ev[j][2].startswith('host hipMemcpy')): j -= 1
for s, e, n, q in ev[j:i1]:
    if n.startswith('host hipGetLastError') or n.startswith('host hipSetDevice') or n.startswith('host __hip'): continue
    print("%8.1f -> %8.1f us  (%6.1f)  %-40s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n, q))
PY
