#!/usr/bin/env python3
"""Per-stage device times (HIP events on the launch stream) of the extractor on a resident batch.
Usage: python tools/phase_timing.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from monoorbslam3_amd import _lib, synth  # noqa: E402
from monoorbslam3_amd.extractor import ORBExtractor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = 1242, 375
base = synth.make_frames(32, W, H)
frames = torch.from_numpy(base).cuda().repeat((B + 31) // 32, 1, 1)[:B].contiguous()
ex = ORBExtractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
cap = ex.max_keypoints(W, H)
kp = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
n = torch.zeros(B, dtype=torch.int32, device="cuda")
ex.set_stage_timing(True)
for _rep in range(1):
    acc = {}
    for it in range(4):
        ex.extract_batch_device(frames.data_ptr(), B, W, H, W, W * H, kp.data_ptr(), desc.data_ptr(), cap,
                                n.data_ptr(), None)
        t = ex.stage_times_ms()
        if it:
            for k, v in t.items():
                acc[k] = acc.get(k, 0) + v / 3
    print("stages_ms", {k: round(v, 3) for k, v in acc.items()}, "kp", float(n.float().mean()), flush=True)
