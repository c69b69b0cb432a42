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
