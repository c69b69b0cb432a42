#!/usr/bin/env python3
"""What each batch stage of the extractor costs when a co-resident partner leaves it only part of a CU: the stage is run ALONE
(orbx_set_stage_timing(1): every kernel on one stream, HIP events around each stage) with its occupancy capped by extra dynamic
LDS per workgroup (orbx_dev_set_lds_pad, a development hook).  The numbers price profiles/r06_pipeline_budget.md.
   python tools/occupancy_sweep.py [frames]         (default 512 frames 1242 x 375 / 2000 features, as bench.py)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from monoorbslam3_amd import _lib, synth  # noqa: E402
from monoorbslam3_amd.extractor import ORBExtractor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ONLY = sys.argv[2] if len(sys.argv) > 2 else None   # e.g. "fast=17,16,15": one stage, these caps, three passes
W, H, NF = 1242, 375, 2000
LDS_CU = 160 * 1024
dev = torch.device("cuda", 0)
base = synth.make_frames(32, W, H, seed=synth.DEFAULT_SEED)
frames = torch.from_numpy(base).to(dev).repeat((B + 31) // 32, 1, 1)[:B].contiguous()
ex = ORBExtractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=0)
cap = ex.max_keypoints(W, H)
d_kp = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
L = _lib.lib()
L.orbx_dev_set_lds_pad.argtypes = [__import__("ctypes").c_int, __import__("ctypes").c_int]


def stage_ms(n=4):
    ex.set_stage_timing(True)
    acc = {}
    for it in range(n + 1):
        ex.extract_batch_device(frames.data_ptr(), B, W, H, W, W * H, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(), st.cuda_stream)
        if it:
            for k, v in ex.stage_times_ms().items():
                acc[k] = acc.get(k, 0.0) + v / n
    torch.cuda.synchronize()
    ex.set_stage_timing(False)
    return acc


# (stage id of the hook, the key of stage_times_ms it shows in, LDS bytes per workgroup as shipped, waves per workgroup,
#  workgroups per CU as shipped, the caps to try as workgroups per CU)
STAGES = [
    (1, "fast", 8768, 1, 18, [16, 14, 12, 9]),
    (2, "octree", 34 * 1024, 4, 4, [3, 2, 1]),
    (3, "orient", 1024, 4, 8, [6, 5, 4, 3, 2]),
    (4, "desc", 28160, 4, 5, [4, 3, 2]),
    (0, "resize", 15456, 4, 8, [6, 4, 3]),
]
if ONLY:
    key_only, _, caps_only = ONLY.partition("=")
    STAGES = [(a, k, c, w_, g, [int(x) for x in caps_only.split(",")] * 3) for a, k, c, w_, g, _ in STAGES if k == key_only]
stage_ms(12)          # clocks and caches warm: the first passes on a fresh box run 5-10 % slower
ref = stage_ms()
print("%d frames %dx%d / %d features; stages alone, ms per launch group (as shipped):" % (B, W, H, NF))
print("   " + "  ".join("%s %.4f" % (k, v) for k, v in ref.items()))
ref_n = d_n.cpu().numpy().copy()
print("%-8s %14s %16s %10s %8s" % ("stage", "workgroups/CU", "waves per SIMD", "ms", "x"))
for sid, key, lds, waves, wg0, caps in STAGES:
    ref = stage_ms()   # as shipped, measured again next to its own caps
    print("%-8s %14d %16.2f %10.4f %8s   (as shipped)" % (key, wg0, wg0 * waves / 4.0, ref[key], "1.00"))
    for wg in caps:
        # the largest workgroup size (a multiple of 256 B) of which exactly `wg` fit a CU's LDS
        pad = max(min((LDS_CU // wg) // 256 * 256, 150 * 1024) - lds, 0)
        assert LDS_CU // (lds + pad) == wg, (key, wg, pad)
        assert L.orbx_dev_set_lds_pad(sid, pad) == 0
        got = stage_ms()
        assert np.array_equal(d_n.cpu().numpy(), ref_n)
        print("%-8s %14d %16.2f %10.4f %8.2f   (pad %d B: %d workgroups fit %d KB)" % (
            key, LDS_CU // (lds + pad), LDS_CU // (lds + pad) * waves / 4.0, got[key], got[key] / ref[key], pad, LDS_CU // (lds + pad), LDS_CU // 1024))
    assert L.orbx_dev_set_lds_pad(sid, 0) == 0
