#!/usr/bin/env python3
"""Write tests/golden/extract_320x240_n300.npz: a small seeded frame and the ORACLE's output for it.

The reference itself cannot run here (OpenCV/Eigen absent), so this fixture is
oracle-generated: it pins the oracle and the HIP path against drift, not against the original.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monoorbslam3_amd import synth  # noqa: E402
from oracle import orb_ref_py  # noqa: E402

img = synth.make_frames(1, 320, 240, seed=424242)[0]
o = orb_ref_py.Oracle(300, 1.2, 8, 20, 7)
kps, desc, counts = o.extract(img)
np.savez_compressed(os.path.join(ROOT, "tests/golden/extract_320x240_n300.npz"), image=img, desc=desc,
                    counts=np.array(counts, np.int32), **{f: kps[f] for f in ("x", "y", "angle", "response", "octave")})
print(len(kps), counts)

# ---- records fixture: the steps either side of the extractor on the frame above (oracle-generated drift guard)
voc = synth.make_vocabulary(6, 3, seed=77, flip_bits=50)
V = orb_ref_py.Vocabulary(voc)
bow_ids, bow_vals, (fv_nodes, fv_off, fv_idx) = V.transform(desc, 1)
cam = dict(width=320, height=240, fx=190.0, fy=190.0, cx=160.0, cy=120.0)
dist = (-0.28, 0.07, 2e-4, 2e-5)
raw, un, cell_start, cell_items = orb_ref_py.frame_post(**cam, dist=dist, kps=kps)
rng = np.random.RandomState(5)
sizes = rng.randint(1, 9, 40)
g_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
g_desc = desc[rng.randint(0, len(desc), int(g_off[-1]))]
medoid = np.array([orb_ref_py.distinctive_descriptor(g_desc[g_off[i]:g_off[i + 1]]) for i in range(len(sizes))], np.int32)
np.savez_compressed(os.path.join(ROOT, "tests/golden/records_320x240.npz"),
                    voc_parent=voc["parent"], voc_leaf=voc["is_leaf"], voc_desc=voc["desc"], voc_weight=voc["weight"],
                    voc_hdr=np.array([voc["k"], voc["L"], voc["scoring"], voc["weighting"]], np.int32),
                    bow_ids=bow_ids, bow_vals=bow_vals, fv_nodes=fv_nodes, fv_off=fv_off, fv_idx=fv_idx,
                    cam=np.array([cam["width"], cam["height"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]], np.float64),
                    dist=np.array(dist, np.float64), un_x=un["x"], un_y=un["y"], cell_start=cell_start, cell_items=cell_items,
                    g_off=g_off, g_desc=g_desc, medoid=medoid)
print("records fixture:", len(bow_ids), "words,", len(fv_nodes), "nodes,", int(cell_start[-1]), "grid items")
