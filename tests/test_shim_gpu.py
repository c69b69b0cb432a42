"""The C++ shims with the reference's class signatures, run end to end on the GPU and compared with the
ctypes path and the oracle (same bits expected everywhere)."""
import os
import subprocess

import numpy as np
import pytest

from monoorbslam3_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_shim_end_to_end(oracle_mod, tmp_path):
    from monoorbslam3_amd.extractor import KP_DTYPE
    from monoorbslam3_amd.matcher import ORBMatcher
    exe = str(tmp_path / "shim_smoke")
    lib = os.path.join(ROOT, "monoorbslam3_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "monoorbslam3_amd", "compat"),
                           os.path.join(ROOT, "tests", "cpp", "shim_smoke.cpp"), "-o", exe,
                           "-L", lib, "-lorbx", "-Wl,-rpath," + lib])
    w, h = 752, 480
    canvas = synth.make_canvas(w + 40, h + 20, seed=1234)
    f1 = np.ascontiguousarray(canvas[5:5 + h, 10:10 + w])
    f2 = np.ascontiguousarray(canvas[8:8 + h, 19:19 + w])
    raw = tmp_path / "in.raw"
    with open(raw, "wb") as f:
        f.write(b"%d %d\n" % (w, h))
        f.write(f1.tobytes())
        f.write(f2.tobytes())
    out = tmp_path / "out.bin"
    print(subprocess.check_output([exe, str(raw), str(out)], text=True))
    buf = open(out, "rb").read()
    n1, n2, n_ini, n_bow, n_tri, levels = np.frombuffer(buf, np.int32, 6)
    o = 24
    k1 = np.frombuffer(buf, KP_DTYPE, n1, o); o += 28 * n1
    d1 = np.frombuffer(buf, np.uint8, 32 * n1, o).reshape(n1, 32); o += 32 * n1
    k2 = np.frombuffer(buf, KP_DTYPE, n2, o); o += 28 * n2
    d2 = np.frombuffer(buf, np.uint8, 32 * n2, o).reshape(n2, 32); o += 32 * n2
    m_ini = np.frombuffer(buf, np.int32, n1, o); o += 4 * n1
    bow = np.frombuffer(buf, np.int32, n2, o); o += 4 * n2
    m_tri = np.frombuffer(buf, np.int32, n1, o); o += 4 * n1
    n_fuse, n3, n_fp = np.frombuffer(buf, np.int32, 3, o); o += 12
    fuse_slot = np.frombuffer(buf, np.int32, n3, o); o += 4 * n3
    fuse_bad = np.frombuffer(buf, np.int32, n_fp, o)
    assert levels == 8
    # extraction: the 2N "initial" extractor of a 1000-feature one == oracle with re-quota
    orc = oracle_mod.Oracle(1000, 1.2, 8, 20, 7)
    orc.requota(2000)
    for img, k, d in ((f1, k1, d1), (f2, k2, d2)):
        ok, od, _ = orc.extract(img)
        assert len(ok) == len(k)
        for fld in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(k[fld], ok[fld]), fld
        assert np.array_equal(d, od)
    # matchers vs oracle on the same data
    pre = np.stack([k1["x"], k1["y"]], axis=1)
    r_n, r_m, _ = oracle_mod.search_for_initialization(0.9, True, k1, d1, k2, d2, w, h, pre, 100)
    assert (n_ini, m_ini.tolist()) == (r_n, r_m.tolist()) and n_ini > 50
    fv1 = synth.feature_vector_by_prefix(d1, 6)
    fv2 = synth.feature_vector_by_prefix(d2, 6)
    ok1 = (np.arange(n1) % 3 != 0).astype(np.uint8)
    r_n, r_mp = oracle_mod.search_by_bow(0.7, True, d1, k1["angle"], ok1, fv1, d2, k2["angle"], np.full(n2, -1, np.int32), fv2)
    assert n_bow == r_n and np.array_equal(bow, r_mp) and n_bow > 20
    r_n, r_m = oracle_mod.search_for_triangulation(False, d1, k1["angle"], ok1, fv1, d2, k2["angle"], np.zeros(n2, np.uint8), fv2)
    assert n_tri == r_n and np.array_equal(m_tri, r_m)
    assert ORBMatcher.DescriptorDistance(d1[0], d2[0]) == int(np.unpackbits(d1[0] ^ d2[0]).sum())
    # ---- the static fuse (ORBMatcher.cpp:524-592): the C++ shim on mirror MapPoints against a Python replay of the same
    # scene over the oracle's per-point search (the scene construction mirrors tests/cpp/shim_smoke.cpp)
    f32 = np.float32
    assert n3 == n1
    uid = iter(range(1, 1 << 30))
    pts = []          # per listed point: None, or dict(i = feature, bad, obs = {key frame id: slot}); "K" = the key frame
    for i in range(n1):
        if i % 11 == 3:
            pts.append(None)
            continue
        mp = dict(i=i, bad=(i % 29 == 7), obs={next(uid): 0 for _ in range(i % 4)})
        pts.append(mp)
        if i % 17 == 5:
            pts.append(mp)
    assert len(pts) == n_fp
    slots = {}        # key-frame slot -> the point object it holds
    for i in range(n1):
        if i % 6 == 1:
            mp = dict(i=-1, bad=(i % 5 == 0), obs={next(uid): 0 for _ in range((i // 6) % 4)}, foreign=True)
            mp["obs"]["K"] = i
            slots[i] = mp
    nq = len(pts)
    q_desc = np.zeros((nq, 32), np.uint8); q_xy = np.zeros((nq, 2), f32); q_radius = np.zeros(nq, f32)
    q_level = np.zeros(nq, np.int32); q_ok = np.zeros(nq, np.uint8)
    sf = np.ones(8, f32)
    for l in range(1, 8):
        sf[l] = sf[l - 1] * f32(1.2)
    for k, mp in enumerate(pts):
        if mp is None:
            continue
        i = mp["i"]
        u = np.floor(k1["x"][i] * f32(8)) / f32(8) + f32((i * 37) % 17 - 8) / f32(8)
        v = np.floor(k1["y"][i] * f32(8)) / f32(8) + f32((i * 53) % 13 - 6) / f32(8)
        if not (0 <= u < w and 0 <= v < h):
            continue
        lvl = min(int(k1["octave"][i]) + (1 if i % 5 == 0 else 0), 7)
        d = d1[i].copy()
        d[i % 32] ^= 1 << (i % 8)
        d[(i * 7) % 32] ^= 1 << ((i // 3) % 8)
        q_desc[k], q_xy[k], q_level[k], q_radius[k], q_ok[k] = d, (u, v), lvl, f32(3) * sf[lvl], 1
    r_bi, r_bd, _ = oracle_mod.search_fuse(q_desc, q_xy, q_radius, q_level, q_ok, k1, d1, w, h, sf * sf)

    def replace(loser, winner):           # MapPoint::replace (MapPoint.cpp:233-264): observations move to the winner
        obs, loser["obs"], loser["bad"] = loser["obs"], {}, True
        for kf, idx in obs.items():
            if kf not in winner["obs"]:
                winner["obs"][kf] = idx
                if kf == "K":
                    slots[idx] = winner
            elif kf == "K":
                del slots[idx]

    num = 0
    for k, mp in enumerate(pts):
        if mp is None or mp["bad"] or "K" in mp["obs"]:      # :534, live
            continue
        if not q_ok[k] or r_bi[k] < 0:
            continue
        b = int(r_bi[k])
        mp1 = slots.get(b)
        if mp1 is None:
            mp["obs"]["K"] = b
            slots[b] = mp
        elif not mp1["bad"]:
            if len(mp1["obs"]) > len(mp["obs"]):
                replace(mp, mp1)
            else:
                replace(mp1, mp)
        num += 1
    want_slot = np.full(n1, -1, np.int32)
    for s_, holder in slots.items():
        want_slot[s_] = -2 if holder.get("foreign") else next(k for k, q in enumerate(pts) if q is holder)
    want_bad = np.array([-1 if q is None else int(q["bad"]) for q in pts], np.int32)
    assert n_fuse == num and n_fuse > 300
    assert np.array_equal(fuse_slot, want_slot) and np.array_equal(fuse_bad, want_bad)


def test_cpp_frame_records_end_to_end(oracle_mod, tmp_path):
    """compat/FramePost.h + compat/ORBVocabulary.h driven like Frame::Frame / Frame::computeBow, then SearchByBow on
    the FeatureVectors they produce; every array equals the oracle's."""
    from monoorbslam3_amd.extractor import KP_DTYPE
    exe = str(tmp_path / "records_smoke")
    lib = os.path.join(ROOT, "monoorbslam3_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "monoorbslam3_amd", "compat"),
                           os.path.join(ROOT, "tests", "cpp", "records_smoke.cpp"), "-o", exe,
                           "-L", lib, "-lorbx", "-Wl,-rpath," + lib])
    w, h = 752, 480
    canvas = synth.make_canvas(w + 40, h + 20, seed=4321)
    frames = [np.ascontiguousarray(canvas[5:5 + h, 10:10 + w]), np.ascontiguousarray(canvas[7:7 + h, 14:14 + w])]
    raw = tmp_path / "in.raw"
    with open(raw, "wb") as f:
        f.write(b"%d %d\n" % (w, h))
        for fr in frames:
            f.write(fr.tobytes())
    voc = synth.make_vocabulary(10, 4, seed=21, flip_bits=60)
    vpath = tmp_path / "voc.txt"
    synth.write_vocabulary_text(voc, str(vpath))
    out = tmp_path / "out.bin"
    print(subprocess.check_output([exe, str(raw), str(vpath), str(out)], text=True))
    buf = open(out, "rb").read()
    orc = oracle_mod.Oracle(1000, 1.2, 8, 20, 7)
    R = oracle_mod.Vocabulary(oracle_mod.parse_vocabulary_text(str(vpath)))
    cam = dict(width=w, height=h, fx=458.654, fy=457.296, cx=367.215, cy=248.375)
    dist = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
    o, recs = 0, []
    for k in range(2):
        n, nw, nf, cols, rows, n_bow = np.frombuffer(buf, np.int32, 6, o); o += 24
        kraw = np.frombuffer(buf, KP_DTYPE, n, o); o += 28 * n
        kun = np.frombuffer(buf, KP_DTYPE, n, o); o += 28 * n
        desc = np.frombuffer(buf, np.uint8, 32 * n, o).reshape(n, 32); o += 32 * n
        cells = []
        for _ in range(cols * rows):
            cnt = int(np.frombuffer(buf, np.int32, 1, o)[0]); o += 4
            cells.append(np.frombuffer(buf, np.int32, cnt, o)); o += 4 * cnt
        bow = np.frombuffer(buf, np.dtype([("id", "<u4"), ("v", "<f8")]), nw, o); o += 12 * nw
        fv_nodes, fv_lists = [], []
        for _ in range(nf):
            node, cnt = np.frombuffer(buf, np.uint32, 2, o); o += 8
            fv_nodes.append(node); fv_lists.append(np.frombuffer(buf, np.uint32, int(cnt), o)); o += 4 * int(cnt)
        # extraction and Frame.cpp:24-51
        ok, od, _ = orc.extract(frames[k])
        assert n == len(ok) and np.array_equal(desc, od)
        w_raw, w_un, w_start, w_items = oracle_mod.frame_post(**cam, dist=dist, kps=ok)
        assert kraw.tobytes() == w_raw.tobytes() and kun.tobytes() == w_un.tobytes()
        assert (cols, rows) == (19, 12)
        for c in range(cols * rows):
            assert np.array_equal(cells[c], w_items[w_start[c]:w_start[c + 1]])
        # Frame::computeBow
        wi, wv, (wn, wo, wx) = R.transform(od, 2)
        assert np.array_equal(bow["id"], wi) and bow["v"].tobytes() == wv.tobytes()
        assert np.array_equal(np.array(fv_nodes, np.uint32), wn)
        for r in range(nf):
            assert np.array_equal(fv_lists[r], wx[wo[r]:wo[r + 1]])
        recs.append((kun, od, (wn, wo, wx), n_bow))
    (k1, d1, fv1, n_bow), (k2, d2, fv2, _) = recs
    r_n, _ = oracle_mod.search_by_bow(0.7, True, d1, k1["angle"], np.ones(len(k1), np.uint8), fv1, d2, k2["angle"],
                                      np.full(len(k2), -1, np.int32), fv2)
    assert n_bow == r_n and n_bow > 30


def test_cpp_caller_on_the_null_stream(tmp_path):
    """include/orbx.h "Streams": a C++ caller that fills on stream 0, calls *_device(..., NULL) and reads on stream 0 needs no
    synchronisation of its own -- one handle behind fills still in flight, and two handles chained
    (tests/cpp/null_stream.cpp; expected values from a popcount loop in that file)."""
    exe = str(tmp_path / "null_stream")
    lib = os.path.join(ROOT, "monoorbslam3_amd", "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "null_stream.cpp"), "-o", exe,
                           "-L", lib, "-lorbx", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + lib])
    out = subprocess.run([exe], text=True, capture_output=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "null stream ok" in out.stdout


def test_two_host_threads_through_the_c_abi(tmp_path):
    """include/orbx.h "Streams and threads" / SURVEY 8b "Threading": the reference runs Tracking and LocalMapping on two threads
    (System.cpp:55; LocalMapping.cpp:45-52, 168, 282, 301) that share one vocabulary.  tests/cpp/two_threads.cpp: thread T loops
    orbx_extract + orbv_transform + orbm_search_by_projection_frame + orbba_pose_optimize_batch, thread M loops orbv_transform +
    orbm_search_for_triangulation + orbm_search_fuse + orbba_local_bundle_adjustment, each on its own handles; every output of
    every iteration equals the single-thread answer byte for byte.  Then both threads run DEVICE chains, each on a non-blocking
    stream of its own (orbx_extract_batch_device -> orbm_best2_device beside orbv_transform_device x2 ->
    orbm_search_for_triangulation_device): equal to the single-thread chains and to the host entry points."""
    exe = str(tmp_path / "two_threads")
    lib = os.path.join(ROOT, "monoorbslam3_amd", "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "two_threads.cpp"), "-o", exe,
                           "-L", lib, "-lorbx", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-pthread", "-Wl,-rpath," + lib])
    out = subprocess.run([exe, "--iters", "12"], text=True, capture_output=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "two threads ok" in out.stdout
