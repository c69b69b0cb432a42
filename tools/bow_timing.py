#!/usr/bin/env python3
"""Device time of Frame::computeBow (orbv_transform_device) at ORBvoc scale: k = 10, L = 6 (1.1 M nodes), a batch of
frames with 2000 descriptors each, levelsup = 4.  CPU oracle timed beside it on one frame.
Usage: python tools/bow_timing.py [batch] [L]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from monoorbslam3_amd import synth  # noqa: E402
from monoorbslam3_amd.vocabulary import ORBVocabulary  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6
N, CAP = 2000, 2048
t0 = time.time()
voc = synth.make_vocabulary(10, L, seed=1, p_early_leaf=0.0, p_stop=0.0)
print("vocabulary: %d nodes, %d words (%.1f s to synthesise)" % (len(voc["parent"]), voc["is_leaf"].sum(), time.time() - t0),
      flush=True)
V = ORBVocabulary.from_arrays(voc)
base = synth.make_descriptors_near_words(voc, 8 * N, seed=2).reshape(8, N, 32)
desc = torch.zeros((B, CAP, 32), dtype=torch.uint8, device="cuda")
desc[:, :N] = torch.from_numpy(base).cuda().repeat((B + 7) // 8, 1, 1)[:B]
n = torch.full((B,), N, dtype=torch.int32, device="cuda")
out = dict(bow_ids=torch.zeros((B, CAP), dtype=torch.int32, device="cuda"),
           bow_vals=torch.zeros((B, CAP), dtype=torch.float64, device="cuda"),
           n_words=torch.zeros(B, dtype=torch.int32, device="cuda"),
           fv_nodes=torch.zeros((B, CAP), dtype=torch.int32, device="cuda"),
           fv_off=torch.zeros((B, CAP + 1), dtype=torch.int32, device="cuda"),
           fv_idx=torch.zeros((B, CAP), dtype=torch.int32, device="cuda"),
           n_fv=torch.zeros(B, dtype=torch.int32, device="cuda"))
s = torch.cuda.Stream()


def run():
    V.transform_device(B, desc.data_ptr(), n.data_ptr(), CAP, 4, out["bow_ids"].data_ptr(), out["bow_vals"].data_ptr(),
                       out["n_words"].data_ptr(), out["fv_nodes"].data_ptr(), out["fv_off"].data_ptr(),
                       out["fv_idx"].data_ptr(), out["n_fv"].data_ptr(), s.cuda_stream)


with torch.cuda.stream(s):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10):
        run()
    e1.record(s)
s.synchronize()
ms = e0.elapsed_time(e1) / 10
print("orbv_transform_device: %.3f ms per batch of %d frames = %.2f us/frame, %.0f frames/s" % (ms, B, ms * 1e3 / B, B / ms * 1e3))
print("words/frame %.0f, fv nodes/frame %.0f" % (out["n_words"].float().mean().item(), out["n_fv"].float().mean().item()))
# single-frame host API latency
d1 = base[0]
for _ in range(3):
    V.transform(d1)
t0 = time.perf_counter()
for _ in range(20):
    V.transform(d1)
print("orbv_transform (host pointers, one frame): %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
from oracle import orb_ref_py  # noqa: E402  (CPU baseline only)
R = orb_ref_py.Vocabulary(voc)
t0 = time.perf_counter()
for _ in range(3):
    R.transform(d1)
print("CPU oracle transform, one thread: %.3f ms/frame" % ((time.perf_counter() - t0) / 3 * 1e3))
