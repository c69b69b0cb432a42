#!/usr/bin/env python3
"""Host-to-host latency of the matcher entry points on BASELINE config 3 (2000 x 2000 descriptors)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from monoorbslam3_amd import synth  # noqa: E402
from monoorbslam3_amd.matcher import ORBMatcher  # noqa: E402

n = 2000
a, b, _ = synth.make_descriptor_pair(n, seed=1)
rng = np.random.RandomState(0)
ang1 = rng.uniform(0, 360, n).astype(np.float32)
ang2 = rng.uniform(0, 360, n).astype(np.float32)
ok = np.ones(n, np.uint8)
mp0 = np.full(n, -1, np.int32)
m = ORBMatcher(0.7, True)


def timeit(fn, reps=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


for bits, label in ((0, "dense: one node, 2000x2000 pairs"), (4, "16 nodes (~125/node)"), (10, "~1000 nodes (~2/node)")):
    fv1, fv2 = synth.feature_vector_by_prefix(a, bits), synth.feature_vector_by_prefix(b, bits)
    ms = timeit(lambda: m.SearchByBow(a, ang1, ok, fv1, b, ang2, mp0, fv2))
    nm = m.SearchByBow(a, ang1, ok, fv1, b, ang2, mp0, fv2)[0]
    ms2 = timeit(lambda: m.SearchForTriangulation(a, ang1, 1 - ok, fv1, b, ang2, np.zeros(n, np.uint8), fv2))
    print("%-36s SearchByBow %.3f ms (%d matches)   SearchForTriangulation %.3f ms" % (label, ms, nm, ms2))
print("hamming_matrix 2000x2000 (8 MB out): %.3f ms;  best2 2000x2000: %.3f ms" % (
    timeit(lambda: ORBMatcher.hamming_matrix(a, b)), timeit(lambda: ORBMatcher.best2(a, b))))
