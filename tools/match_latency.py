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


def timeit(fn, reps=21):
    """median of `reps` calls in ms (single calls are occasionally hit by a host-side hiccup of tens of ms)"""
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


for bits, label in ((0, "dense: one node, 2000x2000 pairs"), (4, "16 nodes (~125/node)"), (10, "~1000 nodes (~2/node)")):
    fv1, fv2 = synth.feature_vector_by_prefix(a, bits), synth.feature_vector_by_prefix(b, bits)
    ms = timeit(lambda: m.SearchByBow(a, ang1, ok, fv1, b, ang2, mp0, fv2))
    nm = m.SearchByBow(a, ang1, ok, fv1, b, ang2, mp0, fv2)[0]
    ms2 = timeit(lambda: m.SearchForTriangulation(a, ang1, 1 - ok, fv1, b, ang2, np.zeros(n, np.uint8), fv2))
    print("%-36s SearchByBow %.3f ms (%d matches)   SearchForTriangulation %.3f ms" % (label, ms, nm, ms2))
# ---- the same searches on device-resident records: device time of the enqueued chain (HIP events), nothing read back
import torch  # noqa: E402
from monoorbslam3_amd.extractor import KP_DTYPE  # noqa: E402
dev = torch.device("cuda", 0)
up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
k1r, k2r = np.zeros(n, KP_DTYPE), np.zeros(n, KP_DTYPE)
k1r["angle"], k2r["angle"] = ang1, ang2
kp = lambda k: torch.from_numpy(np.frombuffer(k.tobytes(), np.uint8).copy()).to(dev)  # noqa: E731


def dev_fv(fv):
    nodes, off, idx = fv
    pad = lambda x, dt, m: torch.from_numpy(np.concatenate([np.asarray(x, dt), np.zeros(max(m - len(x), 0), dt)])).to(dev)  # noqa: E731
    return (pad(nodes, np.uint32, n).view(torch.int32), pad(off, np.int32, n + 1), pad(idx, np.uint32, n).view(torch.int32),
            torch.tensor([len(nodes)], dtype=torch.int32, device=dev))


for bits, label in ((0, "dense: one node, 2000x2000 pairs"), (4, "16 nodes (~125/node)"), (10, "~1000 nodes (~2/node)")):
    fv1, fv2 = synth.feature_vector_by_prefix(a, bits), synth.feature_vector_by_prefix(b, bits)
    d = dict(desc1=up(a), kps1=kp(k1r), kf_mp_ok=up(ok), fv1=dev_fv(fv1), desc2=up(b), kps2=kp(k2r), frame_mp=up(mp0), fv2=dev_fv(fv2),
             result=torch.zeros(8, dtype=torch.int32, device=dev), has_mp1=up(1 - ok), has_mp2=up(np.zeros(n, np.uint8)),
             matches12=torch.zeros(n, dtype=torch.int32, device=dev))
    # a stream of its own (a NULL stream is stream 0 itself for every device entry point, include/orbx.h "Streams"): the events
    # below are recorded on the stream the chain is enqueued on
    ts = torch.cuda.Stream()
    st = ts.cuda_stream
    assert st != 0
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tb, tt = [], []
    for _ in range(12):
        d["frame_mp"].fill_(-1)
        torch.cuda.synchronize()
        with torch.cuda.stream(ts):
            e0.record()
            m.SearchByBowDevice(d, n, n, stream=st)
            e1.record()
            m.SearchForTriangulationDevice(d, n, n, stream=st)
            e2.record()
        torch.cuda.synchronize()
        tb.append(e0.elapsed_time(e1)); tt.append(e1.elapsed_time(e2))
    r = d["result"].cpu().numpy()
    print("%-36s SearchByBow on the device %.3f ms   SearchForTriangulation %.3f ms (device time, median; %d sweeps at most)"
          % (label, sorted(tb)[6], sorted(tt)[6], r[2]))
print("hamming_matrix 2000x2000 (8 MB out): %.3f ms;  best2 2000x2000: %.3f ms" % (
    timeit(lambda: ORBMatcher.hamming_matrix(a, b)), timeit(lambda: ORBMatcher.best2(a, b))))

# ---- window searches on two extracted views (752x480, 2000 features), as the tracking thread calls them every frame
import torch  # noqa: E402,F401  (loads the HIP runtime the library shares)
from monoorbslam3_amd.extractor import ORBExtractor  # noqa: E402

w, h = 752, 480
canvas = synth.make_canvas(w + 40, h + 20, seed=909)
f1 = np.ascontiguousarray(canvas[5:5 + h, 10:10 + w])
f2 = np.ascontiguousarray(canvas[9:9 + h, 16:16 + w])
ex = ORBExtractor(2000, 1.2, 8, 20, 7)
k1, d1 = ex(f1)
k2, d2 = ex(f2)
n1, n2 = len(k1), len(k2)
q_xy = np.stack([k1["x"] - 6.0 + rng.normal(0, 1.5, n1), k1["y"] - 4.0 + rng.normal(0, 1.5, n1)], axis=1).astype(np.float32)
q_ok = np.ones(n1, np.uint8)
mp = np.full(n2, -1, np.int32)
m9 = ORBMatcher(0.9, True)
ms = timeit(lambda: m9.SearchByProjectionFrame(d1, q_xy, (7.0 * k1["size"]).astype(np.float32), k1["octave"], k1["angle"], q_ok,
                                              k2, d2, w, h, mp))
nm = m9.SearchByProjectionFrame(d1, q_xy, (7.0 * k1["size"]).astype(np.float32), k1["octave"], k1["angle"], q_ok, k2, d2, w, h, mp)[0]
print("SearchByProjection(last frame -> frame), %d queries, radius 7*size: %.3f ms (%d matches)" % (n1, ms, nm))
q_level = k1["octave"].astype(np.int32)
q_radius = (4.0 * 1.2 ** q_level).astype(np.float32)
ms = timeit(lambda: ORBMatcher(0.8, True).SearchByProjectionPoints(d1, q_xy, q_radius, q_level, q_ok, k2, d2, w, h, mp))
print("SearchByProjection(map points -> frame), %d queries: %.3f ms" % (n1, ms))
pre = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
ms = timeit(lambda: m9.SearchForInitialization(k1, d1, k2, d2, w, h, pre.copy(), 100))
print("SearchForInitialization (window 100): %.3f ms" % ms)
# ---- SearchForInitialization resolved on the device (one workgroup, k_init_resolve): device time by HIP events on the call's
# stream for two extracted views, a crowded scene of near-duplicates and the worst case the limits allow -- every level-0 feature of
# both frames inside ONE 100-px window, descriptors from 8 clusters (each list holds every candidate, long claimant chains)
from monoorbslam3_amd.frame import FramePost  # noqa: E402
post = FramePost(w, h, 460.0, 460.0, w / 2.0, h / 2.0)


def crowd(nn, seed, x0, x1, y0, y1, n_clusters, p_flip, centres):
    r = np.random.RandomState(seed)
    k = np.zeros(nn, KP_DTYPE)
    k["x"] = r.uniform(x0, x1, nn).astype(np.float32); k["y"] = r.uniform(y0, y1, nn).astype(np.float32)
    k["size"] = 1.0; k["angle"] = (r.normal(40, 25, nn) % 360).astype(np.float32); k["class_id"] = -1
    dd = centres[r.randint(0, n_clusters, nn)] ^ np.packbits(r.uniform(size=(nn, 256)) < p_flip, axis=1, bitorder="little")
    return k, dd.astype(np.uint8)


c40 = np.random.RandomState(2026).randint(0, 256, (40, 32)).astype(np.uint8)
init_cases = [("two extracted views", k1, d1, k2, d2, 1024),
              ("crowded: 900 x 850 in 260 x 200 px", *crowd(900, 1, 200, 460, 100, 300, 40, 0.03, c40), *crowd(850, 2, 200, 460, 100, 300, 40, 0.03, c40), 1024),
              ("worst case: 2000 x 2000 in 90 x 90 px", *crowd(2000, 3, 300, 390, 200, 290, 8, 0.02, c40), *crowd(2000, 4, 300, 390, 200, 290, 8, 0.02, c40), 2048)]
ts_i = torch.cuda.Stream()
for label, ka, da, kb, db_, lcap in init_cases:
    na, nb = len(ka), len(kb)
    pre0 = np.stack([ka["x"], ka["y"]], axis=1).astype(np.float32)
    _, kbu, start, items = post(kb)
    kpb = lambda k: torch.from_numpy(np.frombuffer(np.ascontiguousarray(k).tobytes(), np.uint8).copy()).to(dev)  # noqa: E731
    di = dict(kps1=kpb(ka), desc1=up(da), kps2=kpb(kbu), desc2=up(db_), cell_start=up(start.astype(np.int32)),
              cell_items=up(np.concatenate([items, np.zeros(1, items.dtype)]).astype(np.int32)), pre=up(pre0),
              matches12=torch.zeros(na, dtype=torch.int32, device=dev), result=torch.zeros(8, dtype=torch.int32, device=dev))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    by_lanes = {}
    for lanes in (0, 1, 4, 16, 64):   # ORBM_VAR_INIT_LANES: lanes that share a query's list; 0 = the kernel's own choice
        m9._hd.set_variant("init_lanes", lanes)
        tms = []
        for _ in range(9):
            di["pre"].copy_(up(pre0))
            torch.cuda.synchronize()
            with torch.cuda.stream(ts_i):
                ev0.record()
                m9.SearchForInitializationDevice(di, na, nb, post.cols, post.rows, window=100, list_cap=lcap, stream=ts_i.cuda_stream)
                ev1.record()
            torch.cuda.synchronize()
            tms.append(ev0.elapsed_time(ev1))
        by_lanes[lanes] = sorted(tms)[4]
    m9._hd.set_variant("init_lanes", 0)
    r = di["result"].cpu().numpy()
    host_ms = timeit(lambda: m9.SearchForInitialization(ka, da, kb, db_, w, h, pre0.copy(), 100), reps=5)
    print("SearchForInitialization on the device, %-38s %.3f ms device time (median of 9; %d matches, overflow flag %d, %d sweeps, %d list entries); "
          "with 1 / 4 / 16 / 64 lanes per query %.3f / %.3f / %.3f / %.3f ms; host entry point %.3f ms"
          % (label + ":", by_lanes[0], r[0], r[1], r[2], r[3], by_lanes[1], by_lanes[4], by_lanes[16], by_lanes[64], host_ms))
# map-point descriptors: 3000 points x 2..12 observations
sizes = rng.randint(2, 13, 3000)
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
obs = rng.randint(0, 256, (int(off[-1]), 32)).astype(np.uint8)
print("ComputeDistinctiveDescriptors, 3000 points / %d observations: %.3f ms" % (off[-1],
      timeit(lambda: ORBMatcher.ComputeDistinctiveDescriptors(obs, off))))
