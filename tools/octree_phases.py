#!/usr/bin/env python3
"""Phase time stamps of k_octree_lds for one frame (development build: make -C monoorbslam3_amd/csrc prof).
Usage on the GPU box: python tools/octree_phases.py [batch [W H [name=value ...]]]"""
import ctypes as C
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_dir = os.path.join(ROOT, "monoorbslam3_amd", "lib")
prof = os.path.join(lib_dir, "liborbx_prof.so")
if not os.path.exists(prof):
    sys.exit("build it first: make -C monoorbslam3_amd/csrc prof")
# the package loads lib/liborbx.so: run from a scratch copy of the package with the probe build in its place
tmp = "/tmp/octprof"
shutil.rmtree(tmp, ignore_errors=True)
shutil.copytree(os.path.join(ROOT, "monoorbslam3_amd"), os.path.join(tmp, "monoorbslam3_amd"))
shutil.copy(prof, os.path.join(tmp, "monoorbslam3_amd", "lib", "liborbx.so"))
sys.path.insert(0, tmp)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from monoorbslam3_amd import _lib, synth  # noqa: E402
from monoorbslam3_amd.extractor import ORBExtractor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1242, 375)
imgs = synth.make_frames(B, W, H)
variants = {}
for kv in sys.argv[4:]:  # kernel-choice switches, e.g. split_level0=0 (one chain: the quadtree runs with nothing beside it)
    k, v = kv.split("=")
    variants[k] = int(v) if v.lstrip("-").isdigit() else v
ex = ORBExtractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, variants=variants)
NAMES = {0: "start", 1: "codes + initial nodes", 2: "ranks", 3: "child counts", 4: "apply -> size", 5: "sort (K keys)", 6: "child counts",
         7: "stop index", 8: "apply -> size", 9: "rounds done", 10: "strongest + out (n cand)"}
for it in range(3):
    if B == 1:
        ex(imgs[0])
    else:
        ex.extract_batch(imgs)
L = _lib.lib()
L.orbx_dev_octree_phases.restype = C.c_int
L.orbx_dev_octree_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros((8, 64), np.uint64)
_lib.check(L.orbx_dev_octree_phases(ex._h, out.ctypes.data, 8))
for lvl in range(8):
    n = int(out[lvl, 63])
    t0 = int(out[lvl, 0])
    print("level %d" % lvl)
    prev = t0
    for k in range(n):
        t, tag = int(out[lvl, 2 * k]), int(out[lvl, 2 * k + 1])
        print("   +%7.2f us (%6.2f)  %-26s %s" % ((t - t0) / 100.0, (t - prev) / 100.0, NAMES.get(tag & 255, "?"), (tag >> 8) or ""))
        prev = t
