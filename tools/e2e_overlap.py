#!/usr/bin/env python3
"""Do the PCIe copies of bench.py's end-to-end phase overlap its kernels?  From a rocprofv3 --kernel-trace --memory-copy-trace csv
directory: every large host-to-device copy (one resident batch) with its duration, rate, the engine / agent columns the trace has,
and the fraction of its interval during which an extractor kernel was running; plus any runtime copy kernels (blits) in the trace.
   python tools/e2e_overlap.py <dir> [h2d_bytes]"""
import csv
import glob
import sys

root = sys.argv[1]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 512 * 1242 * 375
copies, kernels, blits = [], [], {}
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        copies.append(r)
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        short = n.split("(")[0].replace("void ", "").split("<")[0].strip().split("::")[-1]
        if short.startswith("k_"):
            kernels.append((s, e, short))
        elif "copy" in n.lower() or "fill" in n.lower() or "rocclr" in n.lower():
            b = blits.setdefault(short, [0, 0])
            b[0] += 1
            b[1] += e - s
kernels.sort()
if copies:
    print("memory-copy trace columns:", ", ".join(copies[0].keys()))


def busy(s, e):
    """ns of [s, e) covered by at least one extractor kernel"""
    t, last = 0, s
    for ks, ke, _ in kernels:
        if ke <= last or ks >= e:
            continue
        a, b = max(ks, last), min(ke, e)
        if b > a:
            t += b - a
            last = b
    return t


# (the trace carries no byte counts: a batch's host-to-device copy is told by its length, several milliseconds)
by_dir = {}
for r in copies:
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    by_dir.setdefault(r.get("Direction", "?"), []).append((s_, e_))
for d, v in sorted(by_dir.items()):
    durs = sorted(e_ - s_ for s_, e_ in v)
    print("%-28s %5d copies, %9.3f ms in all, median %.3f ms, longest %.3f ms" % (d, len(v), sum(durs) / 1e6, durs[len(durs) // 2] / 1e6, durs[-1] / 1e6))
big = sorted((x for v in by_dir.values() for x in v if x[1] - x[0] > 2e6), key=lambda x: x[0])
print("%d copies longer than 2 ms (a resident batch of %d bytes, host to device): those of the end-to-end steps, then those of the copies-only measurement" % (len(big), want))
for s_, e_ in big:
    print("  %8.3f ms  %6.1f GB/s  extractor kernels running during %5.1f %% of it" % ((e_ - s_) / 1e6, want / (e_ - s_), 100.0 * busy(s_, e_) / (e_ - s_)))
if blits:
    print("runtime copy / fill kernels in the kernel trace (a copy done by a kernel competes with the extractor for CUs):")
    for n, (c, t) in sorted(blits.items(), key=lambda kv: -kv[1][1]):
        print("  %-60s %6d launches %10.3f ms" % (n, c, t / 1e6))
else:
    print("no runtime copy kernels in the kernel trace: the copies ran on the DMA engines")
