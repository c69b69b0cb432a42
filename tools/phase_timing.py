#!/usr/bin/env python3
"""Timing experiment: run the extractor with phases of the FAST kernel disabled (results are wrong on
purpose) and print the stage times.  Usage: python tools/phase_timing.py [batch] [flags...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from monoorbslam3_amd import _lib, synth  # noqa: E402
from monoorbslam3_amd.extractor import ORBExtractor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
FLAGS = [int(x) for x in sys.argv[2:]] or [0, 1, 2, 4, 8, 6, 14]
W, H = 1242, 375
base = synth.make_frames(32, W, H)
frames = torch.from_numpy(base).cuda().repeat((B + 31) // 32, 1, 1)[:B].contiguous()
ex = ORBExtractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
cap = ex.max_keypoints(W, H)
kp = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
n = torch.zeros(B, dtype=torch.int32, device="cuda")
L = _lib.lib()
ex.set_stage_timing(True)
for flags in FLAGS:
    L.orbx_debug_set_flags(flags)
    acc = {}
    for it in range(4):
        ex.extract_batch_device(frames.data_ptr(), B, W, H, W, W * H, kp.data_ptr(), desc.data_ptr(), cap,
                                n.data_ptr(), None)
        t = ex.stage_times_ms()
        if it:
            for k, v in t.items():
                acc[k] = acc.get(k, 0) + v / 3
    print("flags", flags, {k: round(v, 3) for k, v in acc.items()}, "kp", float(n.float().mean()), flush=True)
L.orbx_debug_set_flags(0)
