#!/usr/bin/env python3
"""When each cell's workgroup of k_fast_cells_wave started and ended in one single-frame call (development build:
make -C monoorbslam3_amd/csrc prof).  GPU box: python tools/fast_cell_times.py [W H [name=value ...]]"""
import ctypes as C
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(ROOT, "monoorbslam3_amd", "lib", "liborbx_prof.so")
if not os.path.exists(prof):
    sys.exit("build it first: make -C monoorbslam3_amd/csrc prof")
tmp = "/tmp/fcprof"
shutil.rmtree(tmp, ignore_errors=True)
shutil.copytree(os.path.join(ROOT, "monoorbslam3_amd"), os.path.join(tmp, "monoorbslam3_amd"))
shutil.copy(prof, os.path.join(tmp, "monoorbslam3_amd", "lib", "liborbx.so"))
sys.path.insert(0, tmp)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from monoorbslam3_amd import _lib, synth  # noqa: E402
from monoorbslam3_amd.extractor import ORBExtractor  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
variants = {}
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    variants[k] = int(v) if v.lstrip("-").isdigit() else v
img = synth.make_frames(1, W, H)[0]
ex = ORBExtractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, variants=variants)
L = _lib.lib()
L.orbx_dev_fast_cell_times.restype = C.c_int
L.orbx_dev_fast_cell_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
N = 16384
out = np.zeros((N, 3), np.uint64)
for it in range(4):
    ex(img)
    _lib.check(L.orbx_dev_fast_cell_times(ex._h, out.ctypes.data, N, 1))
L.orbx_dev_octree_phases.restype = C.c_int
L.orbx_dev_octree_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
oct = np.zeros((8, 64), np.uint64)
_lib.check(L.orbx_dev_octree_phases(ex._h, oct.ctypes.data, 8))
used = out[out[:, 1] > 0]
t0 = int(used[:, 0].min())
start = (used[:, 0].astype(np.int64) - t0) / 100.0
end = (used[:, 1].astype(np.int64) - t0) / 100.0
work = (used[:, 2].astype(np.int64) - t0) / 100.0   # the cell's own work done, before the group's reservation in the candidate list
print("%d cells; times in us from the first start" % len(used))
# the two launches of a split call (level 0 on the side stream, the rest on the main one) are told apart by their start times
gap = np.sort(start)
cut = gap[np.argmax(np.diff(gap)) + 1] if len(gap) > 1 and np.diff(gap).max() > 3.0 else None
groups = (("first launch", start < cut), ("second launch", start >= cut)) if cut is not None else (("the launch", start >= 0),)
for grp, sel in groups:
    if not sel.any():
        continue
    s, e, w = start[sel], end[sel], work[sel]
    d = w - s
    print("%-13s %5d cells: first start %6.1f  last start %6.1f  last work done %6.1f  last end (reserved + written) %6.1f | work per cell median %5.1f  p90 %5.1f  max %5.1f"
          % (grp, sel.sum(), s.min(), s.max(), w.max(), e.max(), np.median(d), np.percentile(d, 90), d.max()))
for lvl in (0, 1):  # the quadtree kernels' own stamps (same counter): when the next kernel of each chain really started
    n = int(oct[lvl, 63])
    print("quadtree level %d: start %6.1f  end %6.1f" % (lvl, (int(oct[lvl, 0]) - t0) / 100.0, (int(oct[lvl, 2 * (n - 1)]) - t0) / 100.0))
