"""GPU parity of the matcher cores against the CPU oracle, through the C ABI (bit-exact integers)."""
import numpy as np
import pytest

from monoorbslam3_amd import synth

pytestmark = pytest.mark.gpu


def _popcount_ref(a, b):
    return np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2).astype(np.uint16)


def test_descriptor_distance_kats():
    from monoorbslam3_amd.matcher import ORBMatcher
    x = np.arange(32, dtype=np.uint8)
    assert ORBMatcher.DescriptorDistance(x, x) == 0
    assert ORBMatcher.DescriptorDistance(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256
    y = x.copy()
    y[5] ^= 0b10110000
    assert ORBMatcher.DescriptorDistance(x, y) == 3


@pytest.mark.parametrize("na,nb", [(1, 1), (63, 65), (500, 777), (2000, 2000)])
def test_hamming_matrix(na, nb):
    from monoorbslam3_amd.matcher import ORBMatcher
    rng = np.random.RandomState(na * 7 + nb)
    a = rng.randint(0, 256, (na, 32)).astype(np.uint8)
    b = rng.randint(0, 256, (nb, 32)).astype(np.uint8)
    d = ORBMatcher.hamming_matrix(a, b)
    assert np.array_equal(d, _popcount_ref(a, b))


def _best2_ref(d, row_ok, col_ok):
    na, nb = d.shape
    bi = np.full(na, -1, np.int32)
    bd = np.full(na, 256, np.uint16)
    sd = np.full(na, 256, np.uint16)
    for i in range(na):
        if row_ok is not None and not row_ok[i]:
            continue
        best, second, idx = 256, 256, -1
        for j in range(nb):
            if col_ok is not None and not col_ok[j]:
                continue
            v = int(d[i, j])
            if v < best:
                second, best, idx = best, v, j
            elif v < second:
                second = v
        bi[i], bd[i], sd[i] = idx, best, second
    return bi, bd, sd


def test_best2_masked_and_ties():
    from monoorbslam3_amd.matcher import ORBMatcher
    a, b, _ = synth.make_descriptor_pair(300, seed=3)
    b[17] = b[5]            # exact duplicate candidate: first index must win
    b[40] = a[40] ^ 255     # a 256-distance candidate
    rng = np.random.RandomState(1)
    row_ok = (rng.uniform(size=300) > 0.2).astype(np.uint8)
    col_ok = (rng.uniform(size=300) > 0.3).astype(np.uint8)
    d = _popcount_ref(a, b)
    for ro, co in ((None, None), (row_ok, col_ok)):
        got = ORBMatcher.best2(a, b, ro, co)
        ref = _best2_ref(d, ro, co)
        for g, r in zip(got, ref):
            assert np.array_equal(g, r)
    # no candidates at all
    bi, bd, sd = ORBMatcher.best2(a[:5], b[:0])
    assert bi.tolist() == [-1] * 5 and bd.tolist() == [256] * 5 and sd.tolist() == [256] * 5


def test_hamming_csr():
    from monoorbslam3_amd.matcher import ORBMatcher
    a, b, _ = synth.make_descriptor_pair(200, seed=9)
    rng = np.random.RandomState(2)
    q_idx = rng.randint(0, 200, 50)
    lens = rng.randint(0, 90, 50)
    off = np.concatenate([[0], np.cumsum(lens)])
    c_idx = rng.randint(0, 200, off[-1])
    out = ORBMatcher.hamming_csr(a, b, q_idx, off, c_idx)
    d = _popcount_ref(a, b)
    ref = np.concatenate([d[q_idx[q], c_idx[off[q]:off[q + 1]]] for q in range(50)])
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("bits,n", [(10, 2000), (4, 2000), (0, 600)])
@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_bow(oracle_mod, bits, n, check_ori):
    from monoorbslam3_amd.matcher import ORBMatcher
    a, b, _ = synth.make_descriptor_pair(n, seed=bits + 1)
    rng = np.random.RandomState(bits)
    ang1 = rng.uniform(0, 360, n).astype(np.float32)
    ang2 = ((ang1[rng.permutation(n)] + rng.normal(0, 20, n)) % 360).astype(np.float32)
    ok = (rng.uniform(size=n) > 0.25).astype(np.uint8)
    mp0 = np.where(rng.uniform(size=n) > 0.9, 7, -1).astype(np.int32)
    fv1 = synth.feature_vector_by_prefix(a, bits)
    fv2 = synth.feature_vector_by_prefix(b, bits)
    m = ORBMatcher(0.7, check_ori)
    n_got, mp_got = m.SearchByBow(a, ang1, ok, fv1, b, ang2, mp0, fv2)
    n_ref, mp_ref = oracle_mod.search_by_bow(0.7, check_ori, a, ang1, ok, fv1, b, ang2, mp0, fv2)
    assert n_got == n_ref and np.array_equal(mp_got, mp_ref)
    assert n_got > 0


@pytest.mark.parametrize("bits", [8, 3])
@pytest.mark.parametrize("check_ori", [False, True])
def test_search_for_triangulation(oracle_mod, bits, check_ori):
    from monoorbslam3_amd.matcher import ORBMatcher
    n = 1500
    a, b, _ = synth.make_descriptor_pair(n, seed=bits + 20, flip_p=0.06)
    rng = np.random.RandomState(bits)
    ang1 = rng.uniform(0, 360, n).astype(np.float32)
    ang2 = rng.uniform(0, 360, n).astype(np.float32)
    h1 = (rng.uniform(size=n) > 0.6).astype(np.uint8)
    h2 = (rng.uniform(size=n) > 0.6).astype(np.uint8)
    fv1 = synth.feature_vector_by_prefix(a, bits)
    fv2 = synth.feature_vector_by_prefix(b, bits)
    m = ORBMatcher(0.6, check_ori)
    n_got, m_got = m.SearchForTriangulation(a, ang1, h1, fv1, b, ang2, h2, fv2)
    n_ref, m_ref = oracle_mod.search_for_triangulation(check_ori, a, ang1, h1, fv1, b, ang2, h2, fv2)
    assert n_got == n_ref and np.array_equal(m_got, m_ref)
    assert (m_got == 0).sum() == 0  # reference quirk: index 0 is never matched (ORBMatcher.cpp:484)


def _clustered_pair(seed, n_centres=40, per_a=15, per_b=6, flip=3):
    """Many near-duplicate descriptors: several key-frame features compete for the same few frame features, so the
    sequential 'already matched' rule (ORBMatcher.cpp:150 / :466) decides most rows and the device top-K runs dry."""
    rng = np.random.RandomState(seed)
    cen = rng.randint(0, 256, size=(n_centres, 32)).astype(np.uint8)

    def copies(k):
        out = np.repeat(cen, k, axis=0)
        for row in out:
            for bit in rng.choice(256, size=rng.randint(0, flip + 1), replace=False):
                row[bit >> 3] ^= 1 << (bit & 7)
        return out[rng.permutation(len(out))]

    return np.ascontiguousarray(copies(per_a)), np.ascontiguousarray(copies(per_b))


@pytest.mark.parametrize("bits", [0, 2])
@pytest.mark.parametrize("check_ori", [False, True])
def test_search_by_bow_collisions(oracle_mod, bits, check_ori):
    from monoorbslam3_amd.matcher import ORBMatcher
    a, b = _clustered_pair(31 + bits)
    rng = np.random.RandomState(5)
    ang1 = rng.uniform(0, 360, len(a)).astype(np.float32)
    ang2 = rng.uniform(0, 360, len(b)).astype(np.float32)
    ok = np.ones(len(a), np.uint8)
    mp0 = np.where(rng.uniform(size=len(b)) > 0.9, 3, -1).astype(np.int32)
    fv1 = synth.feature_vector_by_prefix(a, bits)
    fv2 = synth.feature_vector_by_prefix(b, bits)
    m = ORBMatcher(0.99, check_ori)
    n_got, mp_got = m.SearchByBow(a, ang1, ok, fv1, b, ang2, mp0, fv2)
    n_ref, mp_ref = oracle_mod.search_by_bow(0.99, check_ori, a, ang1, ok, fv1, b, ang2, mp0, fv2)
    assert n_got == n_ref and np.array_equal(mp_got, mp_ref)
    assert n_got > 10


@pytest.mark.parametrize("bits", [0, 2])
def test_search_for_triangulation_collisions(oracle_mod, bits):
    from monoorbslam3_amd.matcher import ORBMatcher
    a, b = _clustered_pair(77 + bits)
    rng = np.random.RandomState(6)
    ang1 = rng.uniform(0, 360, len(a)).astype(np.float32)
    ang2 = rng.uniform(0, 360, len(b)).astype(np.float32)
    h1 = np.zeros(len(a), np.uint8)
    h2 = (rng.uniform(size=len(b)) > 0.85).astype(np.uint8)
    fv1 = synth.feature_vector_by_prefix(a, bits)
    fv2 = synth.feature_vector_by_prefix(b, bits)
    m = ORBMatcher(0.6, False)
    n_got, m_got = m.SearchForTriangulation(a, ang1, h1, fv1, b, ang2, h2, fv2)
    n_ref, m_ref = oracle_mod.search_for_triangulation(False, a, ang1, h1, fv1, b, ang2, h2, fv2)
    assert n_got == n_ref and np.array_equal(m_got, m_ref)
    assert n_got > 100  # every frame feature without a map point ends up taken


def test_search_for_initialization_on_extracted_frames(oracle_mod):
    """two shifted views of one scene through the GPU extractor, then the initialisation matcher"""
    from monoorbslam3_amd.extractor import ORBExtractor
    from monoorbslam3_amd.matcher import ORBMatcher
    w, h = 752, 480
    canvas = synth.make_canvas(w + 40, h + 20, seed=77)
    f1 = np.ascontiguousarray(canvas[5:5 + h, 10:10 + w])
    f2 = np.ascontiguousarray(canvas[9:9 + h, 22:22 + w])
    ex = ORBExtractor(2000, 1.2, 8, 20, 7)
    k1, d1 = ex(f1)
    k2, d2 = ex(f2)
    pre = np.stack([k1["x"], k1["y"]], axis=1)
    for ori in (True, False):
        m = ORBMatcher(0.9, ori)
        n_got, m_got, pre_got = m.SearchForInitialization(k1, d1, k2, d2, w, h, pre, 100)
        n_ref, m_ref, pre_ref = oracle_mod.search_for_initialization(0.9, ori, k1, d1, k2, d2, w, h, pre, 100)
        assert n_got == n_ref and np.array_equal(m_got, m_ref) and np.array_equal(pre_got, pre_ref)
        assert n_got > 50


@pytest.mark.parametrize("ori", [True, False])
def test_search_for_initialization_on_the_device(oracle_mod, ori):
    """orbm_search_for_initialization_device (ORBMatcher.cpp:33-116 with the window lists, the stealing rule of :63 / :75-81, the
    rotation histogram that keeps robbed queries, ComputeThreeMaxima and the vecPreMatched update on the device): matches12, the
    count and the updated pre-matches equal the oracle's sequential loop and the host entry point -- on two extracted views, and on a
    crowded synthetic scene (near-duplicate descriptors in a small area: many features want the same candidate, several are robbed,
    the fixed point needs more than two sweeps)."""
    import torch
    from monoorbslam3_amd.extractor import ORBExtractor, KP_DTYPE
    from monoorbslam3_amd.frame import FramePost
    from monoorbslam3_amd.matcher import ORBMatcher
    dev = torch.device("cuda", 0)
    w, h = 752, 480
    canvas = synth.make_canvas(w + 40, h + 20, seed=77)
    f1 = np.ascontiguousarray(canvas[5:5 + h, 10:10 + w])
    f2 = np.ascontiguousarray(canvas[9:9 + h, 22:22 + w])
    ex = ORBExtractor(2000, 1.2, 8, 20, 7)
    cases = [("views",) + ex(f1) + ex(f2)]
    # crowded: 900 level-0 features of either frame inside a 260 x 200 px area, descriptors from 40 clusters with a few flipped bits
    rng = np.random.RandomState(2026)
    centres = rng.randint(0, 256, (40, 32)).astype(np.uint8)
    def crowd(n, seed):
        r = np.random.RandomState(seed)
        k = np.zeros(n, KP_DTYPE)
        k["x"] = r.uniform(200, 460, n).astype(np.float32); k["y"] = r.uniform(100, 300, n).astype(np.float32)
        k["size"] = 1.0; k["angle"] = (r.normal(40, 25, n) % 360).astype(np.float32); k["octave"] = (r.uniform(size=n) > 0.85).astype(np.int32)
        k["class_id"] = -1
        d = centres[r.randint(0, 40, n)] ^ np.packbits(r.uniform(size=(n, 256)) < 0.03, axis=1, bitorder="little")
        return k, d.astype(np.uint8)
    cases.append(("crowded",) + crowd(900, 1) + crowd(850, 2))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kp = lambda k: torch.from_numpy(np.frombuffer(np.ascontiguousarray(k).tobytes(), np.uint8).copy()).to(dev)  # noqa: E731
    post = FramePost(w, h, 460.0, 460.0, w / 2.0, h / 2.0)
    m = ORBMatcher(0.9, ori)
    for name, k1, d1, k2, d2 in cases:
        n1, n2 = len(k1), len(k2)
        pre = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
        n_ref, m_ref, pre_ref = oracle_mod.search_for_initialization(0.9, ori, k1, d1, k2, d2, w, h, pre, 100)
        n_host, m_host, pre_host = m.SearchForInitialization(k1, d1, k2, d2, w, h, pre, 100)
        _, k2u, start, items = post(k2)
        assert k2u.tobytes() == np.ascontiguousarray(k2).tobytes()   # no distortion: the record IS the key points
        d = dict(kps1=kp(k1), desc1=up(d1), kps2=kp(k2u), desc2=up(d2), cell_start=up(start.astype(np.int32)),
                 cell_items=up(np.concatenate([items, np.zeros(1, items.dtype)]).astype(np.int32)), pre=up(pre),
                 matches12=torch.full((n1,), 7, dtype=torch.int32, device=dev), result=torch.zeros(8, dtype=torch.int32, device=dev))
        m.SearchForInitializationDevice(d, n1, n2, post.cols, post.rows, window=100, list_cap=1024)
        res = d["result"].cpu().numpy()
        print(name, "device result", res.tolist(), "oracle", n_ref, "host", n_host)
        assert res[1] == 0 and res[0] == n_ref == n_host, (name, res, n_ref, n_host)
        assert np.array_equal(d["matches12"].cpu().numpy(), m_ref) and np.array_equal(m_host, m_ref), name
        assert np.array_equal(d["pre"].cpu().numpy(), pre_ref) and np.array_equal(pre_host, pre_ref), name
        assert n_ref > 50
        if name == "crowded":
            assert res[2] > 2   # the fixed point had chains to follow
        # ORBM_VAR_INIT_LANES: 1, 4, 16 or 64 lanes share a query's window list in the resolve kernel -- the same answer from each
        for lanes in (1, 4, 16, 64):
            m._hd.set_variant("init_lanes", lanes)
            d["pre"].copy_(up(pre)); d["matches12"].fill_(7)
            m.SearchForInitializationDevice(d, n1, n2, post.cols, post.rows, window=100, list_cap=1024)
            res_l = d["result"].cpu().numpy()
            assert res_l[1] == 0 and res_l[0] == n_ref and res_l[2] == res[2], (name, lanes, res_l)
            assert np.array_equal(d["matches12"].cpu().numpy(), m_ref) and np.array_equal(d["pre"].cpu().numpy(), pre_ref), (name, lanes)
        m._hd.set_variant("init_lanes", 0)
        # a pool too small for the lists: reported, matches12 all -1, pre untouched
        d["pre"].copy_(up(pre)); d["matches12"].fill_(7)
        m.SearchForInitializationDevice(d, n1, n2, post.cols, post.rows, window=100, list_cap=2)
        res = d["result"].cpu().numpy()
        assert res[1] == 1 and res[0] == 0 and (d["matches12"].cpu().numpy() == -1).all() and np.array_equal(d["pre"].cpu().numpy(), pre)
    # nothing to match against
    d["matches12"].fill_(7)
    m.SearchForInitializationDevice(d, n1, 0, post.cols, post.rows)
    assert (d["matches12"].cpu().numpy() == -1).all() and int(d["result"][0]) == 0


def _two_views(w=752, h=480, nf=2000):
    from monoorbslam3_amd.extractor import ORBExtractor
    canvas = synth.make_canvas(w + 40, h + 20, seed=909)
    f1 = np.ascontiguousarray(canvas[5:5 + h, 10:10 + w])
    f2 = np.ascontiguousarray(canvas[9:9 + h, 16:16 + w])
    ex = ORBExtractor(nf, 1.2, 8, 20, 7)
    k1, d1 = ex(f1)
    k2, d2 = ex(f2)
    return w, h, k1, d1, k2, d2


@pytest.mark.parametrize("ori", [True, False])
def test_search_by_projection_frame(oracle_mod, ori):
    """last frame -> current frame (ORBMatcher.cpp:203-348): queries carry the (here: shifted) projections"""
    from monoorbslam3_amd.matcher import ORBMatcher
    w, h, k1, d1, k2, d2 = _two_views()
    rng = np.random.RandomState(4)
    n1, n2 = len(k1), len(k2)
    q_xy = np.stack([k1["x"] - 6.0 + rng.normal(0, 1.5, n1), k1["y"] - 4.0 + rng.normal(0, 1.5, n1)], axis=1).astype(np.float32)
    q_ok = (rng.uniform(size=n1) > 0.3).astype(np.uint8)
    q_radius = (7.0 * k1["size"]).astype(np.float32)          # th * key_points[i].size
    mp0 = np.where(rng.uniform(size=n2) > 0.95, 12345, -1).astype(np.int32)
    m = ORBMatcher(0.9, ori)
    got = m.SearchByProjectionFrame(d1, q_xy, q_radius, k1["octave"], k1["angle"], q_ok, k2, d2, w, h, mp0)
    ref = oracle_mod.search_by_projection_frame(ori, d1, q_xy, q_radius, k1["octave"], k1["angle"], q_ok, k2, d2, w, h, mp0)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and got[0] > 100


def test_search_by_projection_points(oracle_mod):
    """local map points -> frame (ORBMatcher.cpp:350-415), with the level-dependent ratio test and the counters"""
    from monoorbslam3_amd.matcher import ORBMatcher
    w, h, k1, d1, k2, d2 = _two_views()
    rng = np.random.RandomState(5)
    n1, n2 = len(k1), len(k2)
    q_xy = np.stack([k1["x"] - 6.0 + rng.normal(0, 2.0, n1), k1["y"] - 4.0 + rng.normal(0, 2.0, n1)], axis=1).astype(np.float32)
    q_ok = (rng.uniform(size=n1) > 0.2).astype(np.uint8)
    q_level = np.clip(k1["octave"] + rng.randint(0, 2, n1), 0, 7).astype(np.int32)
    q_radius = (np.where(rng.uniform(size=n1) > 0.5, 2.5, 4.0) * 3.0 * (1.2 ** q_level)).astype(np.float32)
    mp0 = np.where(rng.uniform(size=n2) > 0.9, 777, -1).astype(np.int32)
    m = ORBMatcher(0.8, True)
    got = m.SearchByProjectionPoints(d1, q_xy, q_radius, q_level, q_ok, k2, d2, w, h, mp0)
    ref = oracle_mod.search_by_projection_points(0.8, d1, q_xy, q_radius, q_level, q_ok, k2, d2, w, h, mp0)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and got[2] == ref[2]
    assert got[0] > 100 and got[2][0] == int((q_ok == 0).sum())


def test_distinctive_descriptors(oracle_mod):
    """MapPoint::computeDescriptor for a batch of map points (MapPoint.cpp:103-152)."""
    from monoorbslam3_amd.matcher import ORBMatcher
    rng = np.random.RandomState(12)
    sizes = [0, 1, 2, 3, 4, 7, 64, 65, 130, 1024] + list(rng.randint(1, 40, 300))
    groups = []
    for k, n in enumerate(sizes):
        if k % 3 == 0:  # near-duplicates: many equal medians, the first index must win
            base = rng.randint(0, 256, (1, 32)).astype(np.uint8)
            g = np.repeat(base, n, axis=0)
            for r in g[n // 2:]:
                r[rng.randint(32)] ^= 1 << rng.randint(8)
        else:
            g = rng.randint(0, 256, (n, 32)).astype(np.uint8)
        groups.append(g)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    got = ORBMatcher.ComputeDistinctiveDescriptors(np.concatenate(groups), off)
    want = [oracle_mod.distinctive_descriptor(g) for g in groups]
    assert list(got) == want
    with pytest.raises(Exception):
        ORBMatcher.ComputeDistinctiveDescriptors(np.zeros((1025, 32), np.uint8), np.array([0, 1025], np.int32))


def _set_best2(mh, variant):
    """"fp4" | "i8" | "valu", or "fp4-resident<k>": the FP4 kernel as k workgroups per CU walking the query blocks"""
    kernel, _, res = variant.partition("-resident")
    mh.set_variant("best2", kernel)
    mh.set_variant("best2_resident", int(res) if res else 0)


@pytest.mark.parametrize("variant", ["fp4", "i8", "valu", "fp4-resident1"])
def test_best2_device_many_problems(oracle_mod, variant):
    """orbm_best2_device the way bench.py drives it: several (A, B) problems in one launch, per-problem counts below
    the strides, device pointers, through each of the three kernels that ship (ORBM_VAR_BEST2: the FP4 matrix path, the i8
    matrix path, the VALU twin; a fresh handle per variant).  Every row equals the oracle's sequential strict-'<' scan
    (ORBMatcher.cpp:148-162); includes a query whose only candidates are at distance 256, empty problems, masked rows and
    candidates, and a row mask WITHOUT a candidate mask (the matrix kernels' own masked-row path)."""
    import torch
    from monoorbslam3_amd import _lib
    from monoorbslam3_amd.matcher import MatcherHandle, _mlib
    rng = np.random.RandomState(2026)
    n_pairs, a_stride, b_stride = 5, 2051, 2100
    na = np.array([2000, 1, 0, 777, 2051], np.int32)
    nb = np.array([1999, 300, 500, 0, 2100], np.int32)
    A = rng.randint(0, 256, (n_pairs, a_stride, 32)).astype(np.uint8)
    B = rng.randint(0, 256, (n_pairs, b_stride, 32)).astype(np.uint8)
    for p in range(n_pairs):  # correlated pairs so that best and second differ a lot, plus exact duplicates (ties)
        m = min(na[p], nb[p])
        if m:
            flips = np.packbits(rng.uniform(size=(m, 256)) < 0.08, axis=1, bitorder="little")
            B[p, :m] = (A[p, :m] ^ flips)[rng.permutation(m)]
        if nb[p] > 20:
            B[p, 17] = B[p, 5]
    B[1, :300] = A[1, 0] ^ 255          # problem 1: every candidate at distance 256 from the only query
    row_ok = np.ones((n_pairs, a_stride), np.uint8)
    col_ok = np.ones((n_pairs, b_stride), np.uint8)
    row_ok[0, ::7] = 0
    col_ok[0, ::5] = 0
    col_ok[4, 100:1500] = 0
    dev = torch.device("cuda", 0)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    dA, dB, dna, dnb, drow, dcol = t(A), t(B), t(na), t(nb), t(row_ok), t(col_ok)
    mh = MatcherHandle(device=0)
    _set_best2(mh, variant)
    L = _mlib()
    for masks in (False, True, "rows"):
        d_bi = torch.full((n_pairs, a_stride), -7, dtype=torch.int32, device=dev)
        d_bd = torch.full((n_pairs, a_stride), -7, dtype=torch.int16, device=dev)
        d_sd = torch.full((n_pairs, a_stride), -7, dtype=torch.int16, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(L.orbm_best2_device(mh._h, n_pairs, dA.data_ptr(), a_stride, dna.data_ptr(), a_stride, dB.data_ptr(),
                                       b_stride, dnb.data_ptr(), b_stride, drow.data_ptr() if masks else None,
                                       dcol.data_ptr() if masks is True else None, d_bi.data_ptr(), d_bd.data_ptr(),
                                       d_sd.data_ptr(), st))
        torch.cuda.synchronize()
        bi, bd, sd = d_bi.cpu().numpy(), d_bd.cpu().numpy().view(np.uint16), d_sd.cpu().numpy().view(np.uint16)
        for p in range(n_pairs):
            cand = np.arange(nb[p]) if masks is not True else np.nonzero(col_ok[p, :nb[p]])[0]
            r_bi, r_bd, r_sd = oracle_mod.best2(A[p, :na[p]], B[p][cand])
            r_bi = np.where(r_bi >= 0, cand[np.maximum(r_bi, 0)] if len(cand) else -1, -1)
            if masks:
                dead = row_ok[p, :na[p]] == 0
                r_bi[dead], r_bd[dead], r_sd[dead] = -1, 256, 256
            assert np.array_equal(bi[p, :na[p]], r_bi), (masks, p)
            assert np.array_equal(bd[p, :na[p]], r_bd), (masks, p)
            assert np.array_equal(sd[p, :na[p]], r_sd), (masks, p)
            # rows between the count and the stride are written as "no candidate", never left stale
            assert (bi[p, na[p]:] == -1).all() and (bd[p, na[p]:] == 256).all() and (sd[p, na[p]:] == 256).all()
    assert bi[1, 0] == -1 and bd[1, 0] == 256  # a 256-distance candidate never beats the initial 256


@pytest.mark.parametrize("resident", [1, 2])
def test_best2_resident_grid_walks_every_block(oracle_mod, resident):
    """ORBM_VAR_BEST2_RESIDENT: k_best2_fp4 as 1 or 2 workgroups per CU that walk the (problem, 512-query block) list -- more
    blocks than the grid holds (700 problems of up to 2 blocks each against 256 or 512 workgroups), ragged counts, so that
    workgroups take two or three blocks of different problems one after the other; every row equals the oracle's scan."""
    import torch
    from monoorbslam3_amd import _lib
    from monoorbslam3_amd.matcher import MatcherHandle, _mlib
    rng = np.random.RandomState(77 + resident)
    n_pairs, a_stride, b_stride = 700, 600, 160
    na = rng.randint(0, a_stride + 1, n_pairs).astype(np.int32)
    nb = rng.randint(0, b_stride + 1, n_pairs).astype(np.int32)
    na[:3], nb[:3] = (a_stride, 0, 513), (b_stride, 40, 0)
    A = rng.randint(0, 256, (n_pairs, a_stride, 32)).astype(np.uint8)
    B = rng.randint(0, 256, (n_pairs, b_stride, 32)).astype(np.uint8)
    B[:, :32] = A[:, 7:39] ^ np.packbits(rng.uniform(size=(n_pairs, 32, 256)) < 0.05, axis=2, bitorder="little")
    dev = torch.device("cuda", 0)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    dA, dB, dna, dnb = t(A), t(B), t(na), t(nb)
    mh = MatcherHandle(device=0)
    _set_best2(mh, "fp4-resident%d" % resident)
    d_bi = torch.full((n_pairs, a_stride), -7, dtype=torch.int32, device=dev)
    d_bd = torch.full((n_pairs, a_stride), -7, dtype=torch.int16, device=dev)
    d_sd = torch.full((n_pairs, a_stride), -7, dtype=torch.int16, device=dev)
    _lib.check(_mlib().orbm_best2_device(mh._h, n_pairs, dA.data_ptr(), a_stride, dna.data_ptr(), a_stride, dB.data_ptr(), b_stride,
                                         dnb.data_ptr(), b_stride, None, None, d_bi.data_ptr(), d_bd.data_ptr(), d_sd.data_ptr(), None))
    bi, bd, sd = d_bi.cpu().numpy(), d_bd.cpu().numpy().view(np.uint16), d_sd.cpu().numpy().view(np.uint16)
    for p in range(n_pairs):
        r_bi, r_bd, r_sd = oracle_mod.best2(A[p, :na[p]], B[p, :nb[p]])
        assert np.array_equal(bi[p, :na[p]], r_bi) and np.array_equal(bd[p, :na[p]], r_bd) and np.array_equal(sd[p, :na[p]], r_sd), p
        assert (bi[p, na[p]:] == -1).all() and (bd[p, na[p]:] == 256).all() and (sd[p, na[p]:] == 256).all()


@pytest.mark.parametrize("variant", ["fp4", "i8", "valu", "fp4-resident2"])
def test_best2_dense_2000x2000_every_kernel(oracle_mod, variant):
    """BASELINE config 3's shape (2000 x 2000 256-bit descriptors, no masks) through each dense best / second-best kernel:
    index, best and second distance of every row equal the oracle's scan."""
    from monoorbslam3_amd.matcher import MatcherHandle, ORBMatcher
    a, b, _ = synth.make_descriptor_pair(2000, seed=77)
    b[1234] = b[77]  # a duplicate candidate: the first index wins
    mh = MatcherHandle()
    _set_best2(mh, variant)
    bi, bd, sd = ORBMatcher.best2(a, b, handle=mh)
    r_bi, r_bd, r_sd = oracle_mod.best2(a, b)
    assert np.array_equal(bi, r_bi) and np.array_equal(bd, r_bd) and np.array_equal(sd, r_sd)


@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_bow_dense_2000x2000(oracle_mod, check_ori):
    """BASELINE config 3: one vocabulary node holding all 2000 x 2000 descriptors (the dense Hamming brute force)."""
    from monoorbslam3_amd.matcher import ORBMatcher
    a, b, _ = synth.make_descriptor_pair(2000, seed=31)
    rng = np.random.RandomState(8)
    ang1 = rng.uniform(0, 360, len(a)).astype(np.float32)
    ang2 = (ang1[rng.permutation(len(a))] + rng.normal(0, 20, len(a))).astype(np.float32) % 360
    ok = (rng.uniform(size=len(a)) > 0.1).astype(np.uint8)
    mp0 = np.where(rng.uniform(size=len(b)) > 0.95, 9, -1).astype(np.int32)
    fv1 = synth.feature_vector_by_prefix(a, 0)
    fv2 = synth.feature_vector_by_prefix(b, 0)
    assert len(fv1[0]) == 1 and fv1[1][-1] == 2000
    m = ORBMatcher(0.7, check_ori)
    n_got, mp_got = m.SearchByBow(a, ang1, ok, fv1, b, ang2, mp0, fv2)
    n_ref, mp_ref = oracle_mod.search_by_bow(0.7, check_ori, a, ang1, ok, fv1, b, ang2, mp0, fv2)
    assert n_got == n_ref and np.array_equal(mp_got, mp_ref)
    assert n_got > (100 if check_ori else 1000)


@pytest.mark.parametrize("check_ori", [True, False])
def test_search_for_triangulation_dense_2000x2000(oracle_mod, check_ori):
    from monoorbslam3_amd.matcher import ORBMatcher
    a, b, _ = synth.make_descriptor_pair(2000, seed=32)
    rng = np.random.RandomState(9)
    ang1 = rng.uniform(0, 360, len(a)).astype(np.float32)
    ang2 = rng.uniform(0, 360, len(b)).astype(np.float32)
    h1 = (rng.uniform(size=len(a)) > 0.8).astype(np.uint8)
    h2 = (rng.uniform(size=len(b)) > 0.8).astype(np.uint8)
    fv1 = synth.feature_vector_by_prefix(a, 0)
    fv2 = synth.feature_vector_by_prefix(b, 0)
    m = ORBMatcher(0.6, check_ori)
    n_got, m_got = m.SearchForTriangulation(a, ang1, h1, fv1, b, ang2, h2, fv2)
    n_ref, m_ref = oracle_mod.search_for_triangulation(check_ori, a, ang1, h1, fv1, b, ang2, h2, fv2)
    assert n_got == n_ref and np.array_equal(m_got, m_ref)
    assert n_got > (50 if check_ori else 500)


def _dev_fv(fv, torch, dev, pad):
    """A FeatureVector as orbv_transform_device leaves it: node ids, CSR offsets, feature indices and the node COUNT on the device
    (arrays padded to the capacity, as the per-frame rows of a batch are)."""
    nodes, off, idx = fv
    t = lambda a, dt, n: torch.from_numpy(np.concatenate([np.asarray(a, dt), np.zeros(max(n - len(a), 0), dt)])).to(dev)  # noqa: E731
    return (t(nodes, np.uint32, pad).view(torch.int32), t(off, np.int32, pad + 1), t(idx, np.uint32, pad).view(torch.int32),
            torch.tensor([len(nodes)], dtype=torch.int32, device=dev))


def _bow_cases():
    out = []
    for bits, n in ((10, 2000), (4, 2000), (0, 600)):          # ~1000 nodes of ~2 features, 16 nodes of ~125, one node
        a, b, _ = synth.make_descriptor_pair(n, seed=bits + 1)
        out.append(("prefix%d" % bits, a, b, bits, 0.7))
    for bits in (0, 2):                                        # clustered near-duplicates: the top-8 lists run dry
        a, b = _clustered_pair(31 + bits)
        out.append(("clustered%d" % bits, a, b, bits, 0.99))
    a, b, _ = synth.make_descriptor_pair(2000, seed=31)        # BASELINE config 3: dense 2000 x 2000 in one node
    out.append(("dense2000", a, b, 0, 0.7))
    return out


@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_bow_and_triangulation_on_the_device(oracle_mod, check_ori):
    """orbm_search_by_bow_device / orbm_search_for_triangulation_device (ORBMatcher.cpp:118-201, :417-522 with the node join,
    the greedy pass, the rotation histogram and ComputeThreeMaxima on the device): frame_mp / matches12 and the counts equal
    the oracle's sequential loops and the host entry points on every node granularity of profiles/r03_match_latency.txt --
    ~1000 nodes, 16 nodes, one node --, on clustered descriptors (many key-frame features want the same frame features, lists
    used up, rows rescanned on the device) and on the dense 2000 x 2000 case."""
    import torch
    from monoorbslam3_amd.matcher import MatcherHandle, ORBMatcher
    dev = torch.device("cuda", 0)
    mh = MatcherHandle(device=0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    for name, a, b, bits, ratio in _bow_cases():
        n1, n2 = len(a), len(b)
        rng = np.random.RandomState(len(name) + bits)
        ang1 = rng.uniform(0, 360, n1).astype(np.float32)
        ang2 = rng.uniform(0, 360, n2).astype(np.float32)
        if n1 == n2:
            ang2 = ((ang1[rng.permutation(n1)] + rng.normal(0, 20, n1)) % 360).astype(np.float32)
        ok = (rng.uniform(size=n1) > 0.2).astype(np.uint8)
        mp0 = np.where(rng.uniform(size=n2) > 0.92, 7, -1).astype(np.int32)
        h1 = (rng.uniform(size=n1) > 0.7).astype(np.uint8)
        h2 = (rng.uniform(size=n2) > 0.8).astype(np.uint8)
        fv1, fv2 = synth.feature_vector_by_prefix(a, bits), synth.feature_vector_by_prefix(b, bits)
        m = ORBMatcher(ratio, check_ori, handle=mh)
        from monoorbslam3_amd.extractor import KP_DTYPE
        k1, k2 = np.zeros(n1, KP_DTYPE), np.zeros(n2, KP_DTYPE)
        k1["angle"], k2["angle"] = ang1, ang2
        kp = lambda k: torch.from_numpy(np.frombuffer(k.tobytes(), np.uint8).copy()).to(dev)  # noqa: E731
        d = dict(desc1=up(a), kps1=kp(k1), kf_mp_ok=up(ok), fv1=_dev_fv(fv1, torch, dev, n1), desc2=up(b), kps2=kp(k2),
                 frame_mp=up(mp0), fv2=_dev_fv(fv2, torch, dev, n2), result=torch.zeros(8, dtype=torch.int32, device=dev),
                 has_mp1=up(h1), has_mp2=up(h2), matches12=torch.full((n1,), -5, dtype=torch.int32, device=dev))
        # -- SearchByBow
        n_ref, mp_ref = oracle_mod.search_by_bow(ratio, check_ori, a, ang1, ok, fv1, b, ang2, mp0, fv2)
        n_host, mp_host = m.SearchByBow(a, ang1, ok, fv1, b, ang2, mp0, fv2)
        m.SearchByBowDevice(d, n1, n2, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        res = d["result"].cpu().numpy()
        assert res[1] == 0, name
        assert res[0] == n_ref == n_host, (name, res, n_ref)
        assert np.array_equal(d["frame_mp"].cpu().numpy(), mp_ref) and np.array_equal(mp_host, mp_ref), name
        assert n_ref > 0
        # -- SearchForTriangulation
        n_ref, m_ref = oracle_mod.search_for_triangulation(check_ori, a, ang1, h1, fv1, b, ang2, h2, fv2)
        m.SearchForTriangulationDevice(d, n1, n2, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        res = d["result"].cpu().numpy()
        assert res[1] == 0 and res[0] == n_ref, (name, res, n_ref)
        assert np.array_equal(d["matches12"].cpu().numpy(), m_ref), name
    # a vocabulary node with more features than the resolve kernel keeps state for: reported, that node untouched
    big = 4500
    a, b, _ = synth.make_descriptor_pair(big, seed=99)
    kb = np.zeros(big, KP_DTYPE)
    fvb = synth.feature_vector_by_prefix(a, 0)
    db = dict(desc1=up(a), kps1=kp(kb), kf_mp_ok=up(np.ones(big, np.uint8)), fv1=_dev_fv(fvb, torch, dev, big), desc2=up(b), kps2=kp(kb),
              frame_mp=torch.full((big,), -1, dtype=torch.int32, device=dev), fv2=_dev_fv(synth.feature_vector_by_prefix(b, 0), torch, dev, big),
              result=torch.zeros(8, dtype=torch.int32, device=dev))
    m.SearchByBowDevice(db, big, big, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert db["result"].cpu().numpy()[1] == 1 and db["result"].cpu().numpy()[0] == 0 and (db["frame_mp"].cpu().numpy() == -1).all()
    # ... and NOTHING of the call may have been written when one node is too large among small ones that could have been
    # resolved: frame_mp comes back exactly as passed (some slots taken), so the host entry point can take over on it (orbm.h)
    big2 = 4400 + 600
    rng2 = np.random.RandomState(777)
    a2 = rng2.randint(0, 256, (big2, 32)).astype(np.uint8)
    b2 = a2 ^ np.packbits(rng2.uniform(size=(big2, 256)) < 0.08, axis=1, bitorder="little")   # b2[i] is a noisy a2[i]
    node = np.r_[np.zeros(4400, np.uint8), (1 + np.arange(600) % 15).astype(np.uint8)]     # 4400 features of either side in node 0
    a2[:, 0] = (a2[:, 0] & 0xF0) | node                                                       # (4 prefix bits), 600 over 15 nodes
    b2[:, 0] = (b2[:, 0] & 0xF0) | node
    mp_in = np.full(big2, -1, np.int32)
    mp_in[::7] = 5
    kb2 = np.zeros(big2, KP_DTYPE)
    db2 = dict(desc1=up(a2), kps1=kp(kb2), kf_mp_ok=up(np.ones(big2, np.uint8)), fv1=_dev_fv(synth.feature_vector_by_prefix(a2, 4), torch, dev, big2),
               desc2=up(b2), kps2=kp(kb2), frame_mp=up(mp_in), fv2=_dev_fv(synth.feature_vector_by_prefix(b2, 4), torch, dev, big2),
               result=torch.zeros(8, dtype=torch.int32, device=dev), has_mp1=up(np.zeros(big2, np.uint8)), has_mp2=up(np.zeros(big2, np.uint8)),
               matches12=torch.full((big2,), 7, dtype=torch.int32, device=dev))
    m.SearchByBowDevice(db2, big2, big2)
    r2 = db2["result"].cpu().numpy()
    assert r2[1] == 1 and r2[0] == 0 and r2[3] == 0 and np.array_equal(db2["frame_mp"].cpu().numpy(), mp_in)
    m.SearchForTriangulationDevice(db2, big2, big2)
    r2 = db2["result"].cpu().numpy()
    assert r2[1] == 1 and r2[0] == 0 and (db2["matches12"].cpu().numpy() == -1).all()
    # the same data with the large node just inside the limit resolves (the small nodes do find matches)
    keep = np.r_[0:4000, 4400:big2]
    a3, b3, n3 = a2[keep], b2[keep], len(keep)
    db3 = dict(desc1=up(a3), kps1=kp(kb2[:n3]), kf_mp_ok=up(np.ones(n3, np.uint8)), fv1=_dev_fv(synth.feature_vector_by_prefix(a3, 4), torch, dev, n3),
               desc2=up(b3), kps2=kp(kb2[:n3]), frame_mp=torch.full((n3,), -1, dtype=torch.int32, device=dev),
               fv2=_dev_fv(synth.feature_vector_by_prefix(b3, 4), torch, dev, n3), result=torch.zeros(8, dtype=torch.int32, device=dev))
    m.SearchByBowDevice(db3, n3, n3)
    r3 = db3["result"].cpu().numpy()
    n_ref3, mp_ref3 = oracle_mod.search_by_bow(ratio, check_ori, a3, np.zeros(n3, np.float32), np.ones(n3, np.uint8), synth.feature_vector_by_prefix(a3, 4),
                                               b3, np.zeros(n3, np.float32), np.full(n3, -1, np.int32), synth.feature_vector_by_prefix(b3, 4))
    assert r3[1] == 0 and r3[0] == n_ref3 > 100 and np.array_equal(db3["frame_mp"].cpu().numpy(), mp_ref3)
    # nothing to do: empty sides leave frame_mp alone and report no match
    d["result"].fill_(9)
    m.SearchByBowDevice(d, 0, n2, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert d["result"].cpu().numpy()[0] == 0


def test_bow_searches_on_the_device_with_entry_less_nodes(oracle_mod):
    """A caller-built FeatureVector may hold a node without entries (orbv_transform_device never writes one).  k_bow_queries gives
    EVERY node a first-query index, so k_bow_resolve's extents [begin[p], begin[p + 1]) never read the scratch an earlier call left
    behind: empty nodes at the front, in the middle (two in a row) and at the end of side 1, after a call of another shape has
    filled the handle's scratch; results equal the oracle's loops (ORBMatcher.cpp:136-185, :448-506 skip such a node)."""
    import torch
    from monoorbslam3_amd.extractor import KP_DTYPE
    from monoorbslam3_amd.matcher import MatcherHandle, ORBMatcher
    dev = torch.device("cuda", 0)
    mh = MatcherHandle(device=0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kp = lambda k: torch.from_numpy(np.frombuffer(k.tobytes(), np.uint8).copy()).to(dev)  # noqa: E731
    m = ORBMatcher(0.8, False, handle=mh)

    def run(a, b, fv1, fv2):
        n1, n2 = len(a), len(b)
        k1, k2 = np.zeros(n1, KP_DTYPE), np.zeros(n2, KP_DTYPE)
        ok, mp0 = np.ones(n1, np.uint8), np.full(n2, -1, np.int32)
        h1, h2 = np.zeros(n1, np.uint8), np.zeros(n2, np.uint8)
        d = dict(desc1=up(a), kps1=kp(k1), kf_mp_ok=up(ok), fv1=_dev_fv(fv1, torch, dev, n1 + 8), desc2=up(b), kps2=kp(k2), frame_mp=up(mp0),
                 fv2=_dev_fv(fv2, torch, dev, n2 + 8), result=torch.zeros(8, dtype=torch.int32, device=dev), has_mp1=up(h1), has_mp2=up(h2),
                 matches12=torch.full((n1,), -5, dtype=torch.int32, device=dev))
        z1, z2 = np.zeros(n1, np.float32), np.zeros(n2, np.float32)
        n_ref, mp_ref = oracle_mod.search_by_bow(0.8, False, a, z1, ok, fv1, b, z2, mp0, fv2)
        m.SearchByBowDevice(d, n1, n2)
        res = d["result"].cpu().numpy()
        assert res[1] == 0 and res[0] == n_ref and np.array_equal(d["frame_mp"].cpu().numpy(), mp_ref)
        t_ref, m_ref = oracle_mod.search_for_triangulation(False, a, z1, h1, fv1, b, z2, h2, fv2)
        m.SearchForTriangulationDevice(d, n1, n2)
        res = d["result"].cpu().numpy()
        assert res[1] == 0 and res[0] == t_ref and np.array_equal(d["matches12"].cpu().numpy(), m_ref)
        return n_ref, t_ref

    # a first call of another shape leaves its begins in the scratch
    a0, b0, _ = synth.make_descriptor_pair(1500, seed=3)
    run(a0, b0, synth.feature_vector_by_prefix(a0, 6), synth.feature_vector_by_prefix(b0, 6))
    a, b, _ = synth.make_descriptor_pair(900, seed=8)
    pa, pb = a[:, 0] & 15, b[:, 0] & 15                      # feature_vector_by_prefix(., 4) buckets by these bits
    nodes1, off1, idx1 = synth.feature_vector_by_prefix(a, 4)
    nodes2, off2, idx2 = synth.feature_vector_by_prefix(b, 4)
    assert len(nodes1) == 16 and len(nodes2) == 16 and sorted(set(pa)) == list(range(16)) and sorted(set(pb)) == list(range(16))
    # side 1 loses the entries of nodes 0, 7, 8 and 15 but keeps the nodes
    lists = [list(idx1[off1[i]:off1[i + 1]]) for i in range(16)]
    for i in (0, 7, 8, 15):
        lists[i] = []
    off_e = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
    idx_e = np.array([j for x in lists for j in x], np.uint32)
    n_ref, t_ref = run(a, b, (np.asarray(nodes1, np.uint32), off_e, idx_e), (nodes2, off2, idx2))
    assert n_ref > 100 and t_ref > 100


def test_search_fuse(oracle_mod):
    """Per-point core of the static fuse SearchByProjection(keyFrame, mapPoints, Map*, th) (ORBMatcher.cpp:524-592):
    KeyFrame window with the strict test, chi-square gate, closest descriptor at distance <= TH_LOW."""
    from monoorbslam3_amd.matcher import ORBMatcher
    w, h, k1, d1, k2, d2 = _two_views()
    rng = np.random.RandomState(15)
    n1 = len(k1)
    sigma2 = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32) ** 2
    # map points = the key frame's own features seen again: jittered projection, a few descriptor bits flipped
    q_xy = np.stack([k1["x"] + rng.normal(0, 1.2, n1) * (1.2 ** k1["octave"]),
                     k1["y"] + rng.normal(0, 1.2, n1) * (1.2 ** k1["octave"])], axis=1).astype(np.float32)
    flips = np.packbits(rng.uniform(size=(n1, 256)) < 0.06, axis=1, bitorder="little")
    q_desc = d1 ^ flips
    q_level = np.clip(k1["octave"] + rng.randint(0, 2, n1), 0, 7).astype(np.int32)
    q_radius = (3.0 * (np.float32(1.2) ** q_level)).astype(np.float32)
    q_ok = (rng.uniform(size=n1) > 0.15).astype(np.uint8)
    # exact-boundary cases for the strict window test: a projection exactly `radius` away from its key point
    q_xy[0] = (k1["x"][0] + q_radius[0], k1["y"][0])
    q_xy[1] = (k1["x"][1] + np.float32(0.5) * q_radius[1], k1["y"][1])
    q_ok[:2] = 1
    m = ORBMatcher()
    bi, bd, n = m.SearchFuse(q_desc, q_xy, q_radius, q_level, q_ok, k1, d1, w, h, sigma2)
    r_bi, r_bd, r_n = oracle_mod.search_fuse(q_desc, q_xy, q_radius, q_level, q_ok, k1, d1, w, h, sigma2)
    assert n == r_n and np.array_equal(bi, r_bi) and np.array_equal(bd, r_bd)
    assert n > 500 and (bi[q_ok == 0] == -1).all() and (bd[bi >= 0] <= 50).all() and (bd[bi < 0] == 51).all()
    # without candidates / without key points
    e_bi, e_bd, e_n = m.SearchFuse(q_desc[:3], q_xy[:3], q_radius[:3], q_level[:3], np.zeros(3, np.uint8), k1, d1, w, h, sigma2)
    assert e_n == 0 and (e_bi == -1).all()
    # ---- the same search on a device-resident key-frame record (orbm_search_fuse_device): the KeyFrame's grid as orbf builds it
    import torch
    from monoorbslam3_amd.frame import FramePost
    dev = torch.device("cuda", 0)
    post = FramePost(w, h, 460.0, 460.0, w / 2.0, h / 2.0)
    _, k1u, start, items = post(k1)
    assert k1u.tobytes() == np.ascontiguousarray(k1).tobytes()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    nq = len(q_desc)
    d = dict(q_desc=up(q_desc), q_xy=up(q_xy), q_radius=up(q_radius), q_level=up(q_level), q_ok=up(q_ok),
             kps=torch.from_numpy(np.frombuffer(np.ascontiguousarray(k1).tobytes(), np.uint8).copy()).to(dev), desc=up(d1),
             cell_start=up(start.astype(np.int32)), cell_items=up(np.concatenate([items, np.zeros(1, items.dtype)]).astype(np.int32)),
             sigma2=up(sigma2), best_idx=torch.full((nq,), 9, dtype=torch.int32, device=dev),
             best_dist=torch.full((nq,), 9, dtype=torch.int32, device=dev), result=torch.zeros(8, dtype=torch.int32, device=dev))
    m.SearchFuseDevice(d, nq, post.cols, post.rows, list_cap=48)
    res = d["result"].cpu().numpy()
    assert res[1] == 0 and res[0] == r_n
    assert np.array_equal(d["best_idx"].cpu().numpy(), r_bi) and np.array_equal(d["best_dist"].cpu().numpy(), r_bd)
    m.SearchFuseDevice(d, nq, post.cols, post.rows, list_cap=1)   # windows with more than one hit: reported
    assert d["result"].cpu().numpy()[1] == 1


def test_window_lists_on_a_device_resident_frame_record(oracle_mod):
    """extract -> orbf frame record (undistorted key points + CSR grid) -> orbm_window_lists_device, all on
    device buffers; every query's list = the oracle's getFeaturesInArea (Frame and KeyFrame variants, level rules, the
    fuse's chi-square gate) in the same order, with the Hamming distance of every hit."""
    import ctypes as C
    import torch
    from monoorbslam3_amd import _lib
    from monoorbslam3_amd.extractor import ORBExtractor, KP_DTYPE
    from monoorbslam3_amd.frame import FramePost
    from monoorbslam3_amd.matcher import MatcherHandle, _mlib
    dev = torch.device("cuda", 0)
    w, h, nf = 752, 480, 1500
    img = torch.from_numpy(synth.make_frames(1, w, h, seed=77)).to(dev)
    ex = ORBExtractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    cap = ex.max_keypoints(w, h)
    d_kp = torch.zeros((1, cap, 28), dtype=torch.uint8, device=dev)
    d_un = torch.zeros((1, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((1,), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_batch_device(img.data_ptr(), 1, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(), st)
    fp = FramePost(w, h, 458.654, 457.296, 367.215, 248.375, dist=(-0.2834, 0.0739, 1.9e-4, 1.8e-5))
    d_start = torch.zeros((1, fp.n_cells + 1), dtype=torch.int32, device=dev)
    d_items = torch.zeros((1, cap), dtype=torch.int32, device=dev)
    fp.post_device(1, d_kp.data_ptr(), d_n.data_ptr(), cap, d_un.data_ptr(), d_start.data_ptr(), d_items.data_ptr(), st)
    torch.cuda.synchronize()
    n = int(d_n[0])
    kps = np.frombuffer(d_un[0, :n].cpu().numpy().tobytes(), KP_DTYPE)
    desc = d_desc[0, :n].cpu().numpy()
    cols, rows = fp.cols, fp.rows
    rng = np.random.RandomState(3)
    nq = 700
    pick = rng.randint(0, n, nq)
    q_xy = np.stack([kps["x"][pick] + rng.normal(0, 6, nq), kps["y"][pick] + rng.normal(0, 6, nq)], 1).astype(np.float32)
    q_xy[:8] = [(-30, 50), (w + 20, 100), (100, -70), (100, h + 55), (0, 0), (w - 1, h - 1), (2000, 2000), (-500, -500)]
    q_r = rng.choice([4.0, 9.5, 17.0, 40.0, 100.0], nq).astype(np.float32)
    q_r[8] = np.float32(abs(kps["x"][pick[8]] - q_xy[8, 0]))       # a key point exactly on the window edge
    q_min = rng.randint(-1, 7, nq).astype(np.int32)
    q_max = np.where(rng.uniform(size=nq) < 0.2, -1, q_min + rng.randint(0, 3, nq)).astype(np.int32)
    q_ok = (rng.uniform(size=nq) > 0.1).astype(np.uint8)
    q_ok[:9] = 1
    q_desc = (desc[pick] ^ np.packbits(rng.uniform(size=(nq, 256)) < 0.05, axis=1, bitorder="little")).astype(np.uint8)
    sigma2 = ((np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    dq, dxy, dr, dmin, dmax, dok, ds2 = t(q_desc), t(q_xy), t(q_r), t(q_min), t(q_max), t(q_ok), t(sigma2)
    mh = MatcherHandle(device=0)
    L = _mlib()
    LCAP = 2048
    for strict, gate in ((0, False), (1, False), (1, True)):
        d_cnt = torch.full((nq,), -9, dtype=torch.int32, device=dev)
        d_lst = torch.zeros((nq, LCAP), dtype=torch.int32, device=dev)
        _lib.check(L.orbm_window_lists_device(mh._h, d_un.data_ptr(), d_desc.data_ptr(), d_start.data_ptr(), d_items.data_ptr(),
                                              cols, rows, dq.data_ptr(), dxy.data_ptr(), dr.data_ptr(), dmin.data_ptr(),
                                              dmax.data_ptr(), dok.data_ptr(), nq, strict, ds2.data_ptr() if gate else None,
                                              LCAP, d_cnt.data_ptr(), d_lst.data_ptr(), st))
        torch.cuda.synchronize()
        cnt, lst = d_cnt.cpu().numpy(), d_lst.cpu().numpy().view(np.uint32)
        total = 0
        for i in range(nq):
            if not q_ok[i]:
                assert cnt[i] == -1
                continue
            ref = oracle_mod.features_in_area(kps, w, h, float(q_xy[i, 0]), float(q_xy[i, 1]), float(q_r[i]), int(q_min[i]),
                                              int(q_max[i]))
            if strict:  # KeyFrame.cpp:204
                ref = [j for j in ref if abs(kps["x"][j] - q_xy[i, 0]) < q_r[i] and abs(kps["y"][j] - q_xy[i, 1]) < q_r[i]]
            if gate:    # ORBMatcher.cpp:566-567
                e2 = [np.float32(np.float32(q_xy[i, 0] - kps["x"][j]) * np.float32(q_xy[i, 0] - kps["x"][j])) +
                      np.float32(np.float32(q_xy[i, 1] - kps["y"][j]) * np.float32(q_xy[i, 1] - kps["y"][j])) for j in ref]
                ref = [j for j, e in zip(ref, e2) if not float(np.float32(e)) > 5.991 * float(sigma2[kps["octave"][j]])]
            got = lst[i, :cnt[i]]
            assert cnt[i] == len(ref) and (got & 0x3FFFFF).tolist() == [int(j) for j in ref], (strict, gate, i)
            if len(ref):
                d = np.unpackbits(desc[np.asarray(ref, int)] ^ q_desc[i], axis=1).sum(1)
                assert (got >> 22).tolist() == d.tolist()
            total += len(ref)
        assert total > (100 if gate else 3000)
    assert cnt[6] == 0 and cnt[7] == 0


def test_window_searches_device_lists_equal_the_host_twin(oracle_mod):
    """The four window searches with their candidate lists built on the device (default) and on the host
    (ORBM_VAR_WINDOW = 1): identical results, both equal to the oracle's."""
    from monoorbslam3_amd.matcher import MatcherHandle, ORBMatcher
    w, h, k1, d1, k2, d2 = _two_views()
    rng = np.random.RandomState(21)
    n1, n2 = len(k1), len(k2)
    dev_h = MatcherHandle()
    host_h = MatcherHandle()
    host_h.set_variant("window", "host")
    q_xy = np.stack([k1["x"] - 6.0 + rng.normal(0, 1.5, n1), k1["y"] - 4.0 + rng.normal(0, 1.5, n1)], axis=1).astype(np.float32)
    q_ok = (rng.uniform(size=n1) > 0.3).astype(np.uint8)
    mp0 = np.where(rng.uniform(size=n2) > 0.95, 12345, -1).astype(np.int32)
    q_level = np.clip(k1["octave"] + rng.randint(0, 2, n1), 0, 7).astype(np.int32)
    sigma2 = ((np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2).astype(np.float32)
    pre = np.stack([k1["x"], k1["y"]], axis=1)
    res = []
    for hd in (dev_h, host_h):
        m = ORBMatcher(0.9, True, handle=hd)
        a = m.SearchByProjectionFrame(d1, q_xy, (14.0 * k1["size"]).astype(np.float32), k1["octave"], k1["angle"], q_ok, k2, d2, w, h, mp0)
        b = m.SearchByProjectionPoints(d1, q_xy, (12.0 * (1.2 ** q_level)).astype(np.float32), q_level, q_ok, k2, d2, w, h, mp0)
        f = m.SearchFuse(d1, q_xy, (3.0 * (np.float32(1.2) ** q_level)).astype(np.float32), q_level, q_ok, k2, d2, w, h, sigma2)
        i = m.SearchForInitialization(k1, d1, k2, d2, w, h, pre, 100)
        res.append((a, b, f, i))
    for x, y in zip(res[0], res[1]):
        for u, v in zip(x, y):
            assert np.array_equal(np.asarray(u), np.asarray(v))
    ref = oracle_mod.search_by_projection_frame(True, d1, q_xy, (14.0 * k1["size"]).astype(np.float32), k1["octave"], k1["angle"], q_ok, k2, d2, w, h, mp0)
    assert res[0][0][0] == ref[0] and np.array_equal(res[0][0][1], ref[1]) and ref[0] > 100
    # a window longer than the device list buffer falls back to the host lists and stays exact
    big = np.full(n1, 400.0, np.float32)
    m = ORBMatcher(0.9, False, handle=dev_h)
    got = m.SearchByProjectionFrame(d1[:50], q_xy[:50], big[:50], k1["octave"][:50], k1["angle"][:50], np.ones(50, np.uint8), k2, d2, w, h, mp0)
    ref = oracle_mod.search_by_projection_frame(False, d1[:50], q_xy[:50], big[:50], k1["octave"][:50], k1["angle"][:50], np.ones(50, np.uint8), k2, d2, w, h, mp0)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1])


def _device_record(w, h, nf, seed):
    """extract -> orbf frame record, everything left on the device; returns the tensors and host copies for the oracle"""
    import torch
    from monoorbslam3_amd.extractor import ORBExtractor, KP_DTYPE
    from monoorbslam3_amd.frame import FramePost
    dev = torch.device("cuda", 0)
    img = torch.from_numpy(synth.make_frames(1, w, h, seed=seed)).to(dev)
    ex = ORBExtractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    cap = ex.max_keypoints(w, h)
    d_kp = torch.zeros((1, cap, 28), dtype=torch.uint8, device=dev)
    d_un = torch.zeros((1, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((1,), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_batch_device(img.data_ptr(), 1, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(), st)
    fp = FramePost(w, h, 458.654, 457.296, 367.215, 248.375, dist=(-0.2834, 0.0739, 1.9e-4, 1.8e-5))
    d_start = torch.zeros((1, fp.n_cells + 1), dtype=torch.int32, device=dev)
    d_items = torch.zeros((1, cap), dtype=torch.int32, device=dev)
    fp.post_device(1, d_kp.data_ptr(), d_n.data_ptr(), cap, d_un.data_ptr(), d_start.data_ptr(), d_items.data_ptr(), st)
    torch.cuda.synchronize()
    n = int(d_n[0])
    kps = np.frombuffer(d_un[0, :n].cpu().numpy().tobytes(), KP_DTYPE)
    desc = d_desc[0, :n].cpu().numpy()
    return dict(dev=dev, n=n, kps=kps, desc=desc, d_un=d_un, d_desc=d_desc, d_start=d_start, d_items=d_items, cols=fp.cols,
                rows=fp.rows)


@pytest.mark.parametrize("mode", ["frame", "points"])
@pytest.mark.parametrize("scene", ["tracking", "contended", "crowded_window"])
def test_search_by_projection_greedy_pass_on_the_device(oracle_mod, mode, scene):
    """ORBMatcher.cpp:229-246 / :379-407 with the greedy claim order resolved on the device (fixed point over the window
    lists) against the oracle's sequential loops: identical frame_mp, match count and counters.
    tracking: every query near its own key point (few conflicts).  contended: 1500 queries drawn around 40 key points with
    radii that cover many of them -- most queries lose their first choice, some several times, chains of displacements.
    crowded_window: every query sees the same few candidates (all 600 queries at one spot), most end without a match."""
    import torch
    from monoorbslam3_amd.matcher import ORBMatcher
    w, h = 752, 480
    R = _device_record(w, h, 1500, seed=91)
    dev, n, kps, desc = R["dev"], R["n"], R["kps"], R["desc"]
    rng = np.random.RandomState({"tracking": 1, "contended": 2, "crowded_window": 3}[scene])
    if scene == "tracking":
        nq = 1200
        pick = rng.randint(0, n, nq)
        q_xy = np.stack([kps["x"][pick] + rng.normal(0, 3, nq), kps["y"][pick] + rng.normal(0, 3, nq)], 1)
        q_r = (7.0 * 1.2 ** kps["octave"][pick]).astype(np.float32)
        flip = 0.04
    elif scene == "contended":
        nq = 1500
        hubs = rng.randint(0, n, 40)
        pick = hubs[rng.randint(0, 40, nq)]
        q_xy = np.stack([kps["x"][pick] + rng.normal(0, 10, nq), kps["y"][pick] + rng.normal(0, 10, nq)], 1)
        q_r = rng.choice([15.0, 25.0, 40.0], nq).astype(np.float32)
        flip = 0.12
    else:
        nq = 600
        pick = np.full(nq, rng.randint(0, n))
        q_xy = np.tile([[kps["x"][pick[0]], kps["y"][pick[0]]]], (nq, 1)) + rng.normal(0, 1.5, (nq, 2))
        q_r = np.full(nq, 30.0, np.float32)
        flip = 0.2
    q_xy = q_xy.astype(np.float32)
    q_level = np.clip(kps["octave"][pick] + rng.randint(-1, 2, nq), 0, 7).astype(np.int32)
    q_angle = ((kps["angle"][pick] + rng.normal(0, 20, nq)) % 360).astype(np.float32)
    q_ok = (rng.uniform(size=nq) > 0.07).astype(np.uint8)
    q_desc = (desc[pick] ^ np.packbits(rng.uniform(size=(nq, 256)) < flip, axis=1, bitorder="little")).astype(np.uint8)
    mp0 = np.where(rng.uniform(size=n) < 0.15, 7, -1).astype(np.int32)      # some slots already hold a map point
    m = ORBMatcher(0.8, True)
    # the oracle's sequential loops (and the host entry point, whose greedy pass is the reference's loop on the same lists)
    if mode == "frame":
        want_n, want_mp = oracle_mod.search_by_projection_frame(True, q_desc, q_xy, q_r, q_level, q_angle, q_ok, kps, desc, w, h, mp0)
        host_n, host_mp = m.SearchByProjectionFrame(q_desc, q_xy, q_r, q_level, q_angle, q_ok, kps, desc, w, h, mp0)
        want_cnt = None
    else:
        want_n, want_mp, want_cnt = oracle_mod.search_by_projection_points(0.8, q_desc, q_xy, q_r, q_level, q_ok, kps, desc, w, h, mp0)
        host_n, host_mp, host_cnt = m.SearchByProjectionPoints(q_desc, q_xy, q_r, q_level, q_ok, kps, desc, w, h, mp0)
        assert tuple(host_cnt) == tuple(want_cnt)
    assert host_n == want_n and np.array_equal(host_mp, want_mp)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    mp_full = np.full(R["d_un"].shape[1], 5, np.int32)
    mp_full[:n] = mp0
    d = dict(q_desc=t(q_desc), q_xy=t(q_xy), q_radius=t(q_r), q_level=t(q_level), q_angle=t(q_angle), q_ok=t(q_ok),
             kps2=R["d_un"], desc2=R["d_desc"], cell_start=R["d_start"], cell_items=R["d_items"], frame_mp=t(mp_full),
             result=torch.full((8,), -7, dtype=torch.int32, device=dev))
    cap = 48
    for attempt in range(6):     # the documented protocol: a pool that is too small reports it and changes nothing
        d["frame_mp"] = t(mp_full)
        m.SearchByProjectionDevice(mode, d, nq, n, R["cols"], R["rows"], list_cap=cap)
        torch.cuda.synchronize()
        res = d["result"].cpu().numpy()
        if res[1] == 0:
            break
        assert np.array_equal(d["frame_mp"].cpu().numpy(), mp_full)
        cap *= 4
    assert res[1] == 0
    got_mp = d["frame_mp"].cpu().numpy()
    print("%s / %s: %d queries, %d list entries, %d matches, %d sweeps, list_cap %d" % (mode, scene, nq, res[3], res[0], res[2], cap))
    assert res[0] == want_n
    assert np.array_equal(got_mp[:n], want_mp) and np.all(got_mp[n:] == 5)
    if mode == "points":
        assert tuple(res[4:7]) == tuple(want_cnt)
    if scene != "tracking":
        assert res[2] >= 3        # the scene really needs the iteration


@pytest.mark.parametrize("mode", ["frame", "points"])
def test_search_by_projection_device_longest_displacement_chain(oracle_mod, mode):
    """The worst case for the fixed point: 96 candidates in one window, every query carries the same descriptor and so
    wants them in the same order -- query i ends with the (i+1)-th closest one after being displaced i times.  The record
    is hand-made (positions, descriptors with 1, 2, 3 ... bits flipped, the oracle's CSR grid), uploaded, and the device
    result must still be the sequential loop's, in about as many sweeps as the chain is long."""
    import torch
    from monoorbslam3_amd.extractor import KP_DTYPE
    from monoorbslam3_amd.matcher import ORBMatcher
    dev = torch.device("cuda", 0)
    w, h, n, nq = 752, 480, 96, 120
    rng = np.random.RandomState(11)
    kps = np.zeros(n, KP_DTYPE)
    kps["x"] = 300 + rng.uniform(-8, 8, n).astype(np.float32)
    kps["y"] = 200 + rng.uniform(-8, 8, n).astype(np.float32)
    kps["size"], kps["octave"], kps["class_id"] = 1.0, 2, -1
    kps["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    base = rng.randint(0, 256, 32).astype(np.uint8)
    desc = np.tile(base, (n, 1))
    order = rng.permutation(n)              # candidate order[k] is the k-th closest: k + 1 flipped bits
    for k, c in enumerate(order):
        bits = np.zeros(256, bool)
        bits[rng.choice(256, k + 1, replace=False)] = True
        desc[c] ^= np.packbits(bits, bitorder="little")
    _, un, start, items = oracle_mod.frame_post(w, h, 458.654, 457.296, 367.215, 248.375, (0.0, 0.0, 0.0, 0.0), kps, undistort=False)
    q_desc = np.tile(base, (nq, 1))
    q_xy = np.tile(np.float32([[300, 200]]), (nq, 1))
    q_r = np.full(nq, 25.0, np.float32)
    q_level = np.full(nq, 2, np.int32)
    q_angle = rng.uniform(0, 360, nq).astype(np.float32)
    q_ok = np.ones(nq, np.uint8)
    mp0 = np.full(n, -1, np.int32)
    if mode == "frame":
        want_n, want_mp = oracle_mod.search_by_projection_frame(False, q_desc, q_xy, q_r, q_level, q_angle, q_ok, un, desc, w, h, mp0)
        assert want_n == n and np.array_equal(want_mp[order], np.arange(n))    # query i holds the i-th closest candidate
    else:
        want_n, want_mp, want_cnt = oracle_mod.search_by_projection_points(0.8, q_desc, q_xy, q_r, q_level, q_ok, un, desc, w, h, mp0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    cols, rows = (w + 39) // 40, (h + 39) // 40
    assert len(start) == cols * rows + 1
    d = dict(q_desc=t(q_desc), q_xy=t(q_xy), q_radius=t(q_r), q_level=t(q_level), q_angle=t(q_angle), q_ok=t(q_ok),
             kps2=t(np.frombuffer(un.tobytes(), np.uint8).reshape(n, 28).copy()), desc2=t(desc), cell_start=t(start.astype(np.int32)),
             cell_items=t(items.astype(np.int32)), frame_mp=t(mp0), result=torch.zeros(8, dtype=torch.int32, device=dev))
    m = ORBMatcher(0.8, False)
    m.SearchByProjectionDevice(mode, d, nq, n, cols, rows, list_cap=128)
    torch.cuda.synchronize()
    res = d["result"].cpu().numpy()
    print("%s: %d matches in %d sweeps over %d list entries" % (mode, res[0], res[2], res[3]))
    assert res[1] == 0 and res[3] == nq * n
    assert res[0] == want_n and np.array_equal(d["frame_mp"].cpu().numpy(), want_mp)
    if mode == "frame":
        assert res[2] >= n // 2
    else:
        assert tuple(res[4:7]) == tuple(want_cnt)


@pytest.mark.parametrize("mode", ["frame", "points"])
def test_search_by_projection_device_edge_cases(oracle_mod, mode):
    """No queries; every query switched off; every slot of the frame already taken; a pool too small for the lists (reported,
    nothing changed); a query set larger than the LDS budget (refused with ORBX_E_UNSUPPORTED, no launch)."""
    import torch
    from monoorbslam3_amd._lib import OrbxError
    from monoorbslam3_amd.matcher import ORBMatcher
    w, h = 752, 480
    R = _device_record(w, h, 1000, seed=5)
    dev, n, kps, desc = R["dev"], R["n"], R["kps"], R["desc"]
    capk = R["d_un"].shape[1]
    rng = np.random.RandomState(4)
    nq = 300
    pick = rng.randint(0, n, nq)
    q_xy = np.stack([kps["x"][pick], kps["y"][pick]], 1).astype(np.float32)
    q_r = np.full(nq, 12.0, np.float32)
    q_level = kps["octave"][pick].astype(np.int32)
    q_angle = kps["angle"][pick].astype(np.float32)
    q_desc = desc[pick].copy()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    m = ORBMatcher(0.8, True)

    def run(nq_, q_ok, mp0, cap=48):
        d = dict(q_desc=t(q_desc), q_xy=t(q_xy), q_radius=t(q_r), q_level=t(q_level), q_angle=t(q_angle), q_ok=t(q_ok),
                 kps2=R["d_un"], desc2=R["d_desc"], cell_start=R["d_start"], cell_items=R["d_items"], frame_mp=t(mp0),
                 result=torch.full((8,), -3, dtype=torch.int32, device=dev))
        m.SearchByProjectionDevice(mode, d, nq_, capk, R["cols"], R["rows"], list_cap=cap)
        torch.cuda.synchronize()
        return d["result"].cpu().numpy(), d["frame_mp"].cpu().numpy()
    free = np.full(capk, -1, np.int32)
    res, mp = run(0, np.ones(nq, np.uint8), free)                     # no queries
    assert res[0] == 0 and res[1] == 0 and np.array_equal(mp, free)
    res, mp = run(nq, np.zeros(nq, np.uint8), free)                    # all switched off
    assert res[0] == 0 and res[3] == 0 and np.array_equal(mp, free) and (mode == "frame" or res[4] == nq)
    taken = np.full(capk, 9, np.int32)
    res, mp = run(nq, np.ones(nq, np.uint8), taken)                    # nothing free
    assert res[0] == 0 and res[3] > 0 and np.array_equal(mp, taken) and (mode == "frame" or res[6] > 0)
    big = q_r.copy()
    q_r[:] = 300.0                                                     # windows of hundreds of key points: 300 * 1 entries is too few
    res, mp = run(nq, np.ones(nq, np.uint8), free, cap=1)
    assert res[1] == 1 and res[0] == 0 and np.array_equal(mp, free)
    q_r[:] = big
    with pytest.raises(OrbxError):                                     # nq + n2 above what one workgroup's LDS holds
        d = dict(q_desc=t(q_desc), q_xy=t(q_xy), q_radius=t(q_r), q_level=t(q_level), q_angle=t(q_angle), q_ok=t(np.ones(nq, np.uint8)),
                 kps2=R["d_un"], desc2=R["d_desc"], cell_start=R["d_start"], cell_items=R["d_items"], frame_mp=t(free),
                 result=torch.zeros(8, dtype=torch.int32, device=dev))
        m.SearchByProjectionDevice(mode, d, nq, 60000, R["cols"], R["rows"])
