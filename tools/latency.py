#!/usr/bin/env python3
"""Single-call latency of the drop-in path (host pointers in, host pointers out) and per-stage device times
at batch 1.  Usage: python tools/latency.py [W H [NF]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from monoorbslam3_amd import synth  # noqa: E402
from monoorbslam3_amd.extractor import ORBExtractor  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 2 else 1242
H = int(sys.argv[2]) if len(sys.argv) > 2 else 375
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
img = synth.make_frames(1, W, H)[0]
variants = {}
for kv in sys.argv[4:]:  # kernel-choice switches (monoorbslam3_amd/extractor.py VARIANTS), e.g. split_level0=0
    k, v = kv.split("=")
    variants[k] = int(v) if v.lstrip("-").isdigit() else v
ex = ORBExtractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, variants=variants)
for _ in range(5):
    ex(img)
t0 = time.perf_counter()
N = 50
for _ in range(N):
    k, d = ex(img)
dt = (time.perf_counter() - t0) / N
print("%dx%d nf=%d: host-to-host operator() %.1f us/frame (%d keypoints)" % (W, H, NF, dt * 1e6, len(k)))
ex.set_stage_timing(True)
acc = {}
for it in range(11):
    ex(img)
    t = ex.stage_times_ms()
    if it:
        for kk, v in t.items():
            acc[kk] = acc.get(kk, 0) + v / 10
print("   device stages (us):", {kk: round(v * 1e3, 1) for kk, v in acc.items()}, "sum %.1f" % (sum(acc.values()) * 1e3))
