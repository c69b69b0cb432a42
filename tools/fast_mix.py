#!/usr/bin/env python3
"""Stage counts of k_fast_strip over one 512-frame launch pair (probe build: make -C monoorbslam3_amd/csrc prof).
GPU box:  python tools/fast_mix.py [batch]   ->  gpurun_out/fast_stage_counts.json  (input of tools/isa_mix.py)"""
import ctypes as C
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(ROOT, "monoorbslam3_amd", "lib", "liborbx_prof.so")
if not os.path.exists(prof):
    sys.exit("build it first: make -C monoorbslam3_amd/csrc prof")
tmp = "/tmp/fastprof"
shutil.rmtree(tmp, ignore_errors=True)
shutil.copytree(os.path.join(ROOT, "monoorbslam3_amd"), os.path.join(tmp, "monoorbslam3_amd"))
shutil.copy(prof, os.path.join(tmp, "monoorbslam3_amd", "lib", "liborbx.so"))
sys.path.insert(0, tmp)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from monoorbslam3_amd import _lib, synth  # noqa: E402
from monoorbslam3_amd.extractor import ORBExtractor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N_SHAPES = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None   # corner density of the synthetic frames (bench.py's density sweep: 50 / 150)
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1242, 375)   # (other shapes: counts only, the ISA mix file stays KITTI's)
base = synth.make_frames(32, W, H, seed=synth.DEFAULT_SEED, n_shapes=N_SHAPES)          # the bench's batch: 32 distinct frames + per-copy noise
frames = torch.from_numpy(base).cuda().repeat((B + 31) // 32, 1, 1)[:B].contiguous()
g = torch.Generator(device="cpu").manual_seed(1234)
noise = torch.randint(-2, 3, frames.shape, generator=g, dtype=torch.int16).cuda()
noise[:32] = 0
frames = (frames.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8).contiguous()
ex = ORBExtractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
cap = ex.max_keypoints(W, H)
kp = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
de = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
n = torch.zeros((B,), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
L = _lib.lib()
L.orbx_dev_fast_prof.restype = C.c_int
L.orbx_dev_fast_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros(16, np.uint64)
ex.extract_batch_device(frames.data_ptr(), B, W, H, W, W * H, kp.data_ptr(), de.data_ptr(), cap, n.data_ptr(), st)
_lib.check(L.orbx_dev_fast_prof(ex._h, out.ctypes.data, 1))           # warm-up discarded
ex.extract_batch_device(frames.data_ptr(), B, W, H, W, W * H, kp.data_ptr(), de.data_ptr(), cap, n.data_ptr(), st)
_lib.check(L.orbx_dev_fast_prof(ex._h, out.ctypes.data, 1))
names = ["strips", "tile_loads", "compass_steps", "arc_batches", "score_batches", "nms_batches", "passes", "items", "pixels",
         "passes_overflowed", "narrow_passes", "retry_passes"]
res = {k: int(v) for k, v in zip(names, out)}
res["frames"] = B
print(res)
print("per strip: %.2f tile loads, %.1f compass steps, %.2f 16-point batches (%.1f items each), %.2f strength batches (%.1f pixels each)"
      % (res["tile_loads"] / res["strips"], res["compass_steps"] / res["strips"], res["arc_batches"] / res["strips"],
         res["items"] / max(res["arc_batches"], 1), res["score_batches"] / res["strips"], res["pixels"] / max(res["score_batches"], 1)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
if N_SHAPES is None and (W, H) == (1242, 375):
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "fast_stage_counts.json"), "w"))
