#!/usr/bin/env python3
"""Timeline of the kernels of the last few bench steps from a rocprofv3 --kernel-trace csv:
   python tools/step_timeline.py <dir with *kernel_trace.csv> [n_steps]
prints, for the last n steps, every kernel's start / end (us, relative to the step's first resize) and the queue it ran on,
plus the time during which 0 / 1 / 2 / 3+ kernels were in flight."""
import csv, glob, sys
root = sys.argv[1]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip().split("::")[-1]
        if name.startswith("k_"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
# a step starts at the first pyramid launch after a descriptor kernel
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_resize") and (i == 0 or not any(q[2].startswith("k_resize") for q in rows[max(0, i - 3):i]))]
starts = [i for i in starts if i == 0 or True]
sel = starts[-(n_steps + 1):]
t0 = rows[sel[0]][0]
seg = rows[sel[0]:sel[-1]]
for s, e, n, q in seg:
    print("%9.1f %9.1f  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
ev = sorted([(s, 1) for s, e, n, q in seg] + [(e, -1) for s, e, n, q in seg])
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
    depth += d
    last = t
tot = sum(hist.values())
print("span %.1f us over %d step(s): " % (tot / 1e3, len(sel) - 1) + ", ".join("%d%s kernels %.1f %%" % (k, "+" if k == 3 else "", 100. * v / tot) for k, v in sorted(hist.items())))
