"""ctypes loader for the CPU oracle (oracle/liborb_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (monoorbslam3_amd/) never
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborb_ref.so")
MAX_LEVELS = 16


class Cfg(C.Structure):
    _fields_ = [
        ("n_features", C.c_int), ("n_levels", C.c_int), ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int),
        ("scale_factor", C.c_float), ("log_scale_factor", C.c_float),
        ("scale_factors", C.c_float * MAX_LEVELS), ("inv_scale_factors", C.c_float * MAX_LEVELS),
        ("square_sigmas", C.c_float * MAX_LEVELS), ("inv_square_sigmas", C.c_float * MAX_LEVELS),
        ("n_features_per_level", C.c_int * MAX_LEVELS), ("u_max", C.c_int * 16), ("blur_taps", C.c_int * 7),
    ]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4")])


class Camera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("n_dist", C.c_int32), ("dist", C.c_float * 12), ("undistort", C.c_int32),
                ("size_scale", C.c_void_p)]


class Fv(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("node_ids", C.c_void_p), ("offsets", C.c_void_p), ("indices", C.c_void_p)]


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
            for f in ("orb_ref.c", "orb_ref.h", "orb_pattern.inc")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


def build_native():
    """Rebuild the oracle -O3 -march=native for THIS machine and make lib() use it (bench.py's cpu_baseline leg only: the
    tests keep the portable -O2 build).  Returns a description of the build that is loaded."""
    global _LIB, _lib
    native = os.path.join(_HERE, "liborb_ref_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])
    except Exception:
        return "gcc -O2 (the -march=native build failed on this machine)"
    _LIB, _lib = native, None
    return "gcc -O3 -march=native -ffp-contract=off"


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orbref_fast_atan2.restype = C.c_float
        L.orbref_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orbref_ic_angle.restype = C.c_float
        L.orbref_round_f.argtypes = [C.c_float]
        L.orbref_floor_f.argtypes = [C.c_float]
        L.orbref_ceil_f.argtypes = [C.c_float]
        L.orbref_round_d.argtypes = [C.c_double]
        L.orbref_pattern.restype = C.POINTER(C.c_int8)
        L.orbref_cfg_init.argtypes = [C.POINTER(Cfg), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orbref_sincos_deg.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orbref_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orbref_sincosf_check_libm.restype = C.c_long
        L.orbref_sincos_deg_n.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orbref_sincosf_check_libm.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orbref_ic_angle.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orbref_brief.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orbref_grid_build.restype = C.c_void_p
        L.orbref_voc_build.restype = C.c_void_p
        L.orbref_voc_build.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        L.orbref_voc_free.argtypes = [C.c_void_p]
        L.orbref_voc_n_words.argtypes = [C.c_void_p]
        L.orbref_voc_transform_feature.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbref_voc_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
        L.orbref_frame_post.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_int, C.c_void_p]
        L.orbref_grid_free.argtypes = [C.c_void_p]
        L.orbref_features_in_area.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int,
                                              C.c_int, C.c_void_p, C.c_int]
        L.orbref_search_by_bow.argtypes = [C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.POINTER(Fv), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Fv)]
        L.orbref_search_for_triangulation.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                      C.POINTER(Fv), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                      C.POINTER(Fv), C.c_void_p]
        L.orbref_search_for_initialization.argtypes = [C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                       C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                       C.c_void_p, C.c_void_p, C.c_int]
        L.orbref_search_by_projection_frame.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                        C.c_int, C.c_void_p]
        L.orbref_search_by_projection_points.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                         C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                         C.c_int, C.c_void_p, C.c_void_p]
        L.orbref_search_fuse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Thin object wrapper: one orbref_cfg + stage functions on numpy arrays."""

    def __init__(self, n_features=1000, scale_factor=1.2, n_levels=8, ini_th_fast=20, min_th_fast=10,
                 blur_variant=0):
        self.L = lib()
        self.cfg = Cfg()
        self.L.orbref_cfg_init(C.byref(self.cfg), n_features, scale_factor, n_levels, ini_th_fast, min_th_fast)
        self.L.orbref_set_blur_taps(C.byref(self.cfg), blur_variant)

    # -- tables -----------------------------------------------------------
    @property
    def n_levels(self):
        return self.cfg.n_levels

    def quotas(self):
        return list(self.cfg.n_features_per_level[: self.cfg.n_levels])

    def requota(self, n_features):
        self.L.orbref_cfg_requota(C.byref(self.cfg), n_features)

    def u_max(self):
        return list(self.cfg.u_max)

    def scale_factors(self):
        return np.array(self.cfg.scale_factors[: self.cfg.n_levels], dtype=np.float32)

    def level_size(self, w, h, level):
        lw, lh = C.c_int(), C.c_int()
        self.L.orbref_level_size(C.byref(self.cfg), w, h, level, C.byref(lw), C.byref(lh))
        return lw.value, lh.value

    # -- stages -----------------------------------------------------------
    def resize(self, src, dw, dh):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.empty((dh, dw), dtype=np.uint8)
        self.L.orbref_resize_linear(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw)
        return dst

    def pyramid(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        levels = [np.empty(self.level_size(w, h, l)[::-1], dtype=np.uint8) for l in range(self.n_levels)]
        ptrs = (C.c_void_p * self.n_levels)(*[l.ctypes.data for l in levels])
        self.L.orbref_pyramid(C.byref(self.cfg), _p(img), w, h, w, ptrs)
        return levels

    def fast_box(self, img, x0, y0, x1, y1, threshold):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        cap = max((x1 - x0) * (y1 - y0), 1)
        out = np.zeros(cap, dtype=CAND_DTYPE)
        n = self.L.orbref_fast_box(_p(img), img.shape[1], x0, y0, x1, y1, threshold, _p(out), cap)
        return out[:n]

    def fast_strength(self, img, x, y):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        return self.L.orbref_fast_strength(_p(img), img.shape[1], x, y)

    def level_candidates(self, level_img):
        img = np.ascontiguousarray(level_img, dtype=np.uint8)
        h, w = img.shape
        cap = max((w - 38) * (h - 38), 1)
        out = np.zeros(cap, dtype=CAND_DTYPE)
        n = self.L.orbref_level_candidates(C.byref(self.cfg), _p(img), w, h, w, _p(out), cap)
        return out[:n]

    def distribute(self, cands, min_x, max_x, min_y, max_y, n_features):
        cands = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        out = np.zeros(max(len(cands), 1), dtype=CAND_DTYPE)
        n = self.L.orbref_distribute_octree(_p(cands), len(cands), min_x, max_x, min_y, max_y, n_features,
                                            _p(out), len(out))
        return out[:n]

    def ic_angle(self, img, x, y):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        return float(self.L.orbref_ic_angle(C.byref(self.cfg), _p(img), img.shape[1], x, y))

    def blur(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        out = np.empty_like(img)
        self.L.orbref_gaussian_blur7(C.byref(self.cfg), _p(img), img.shape[1], img.shape[0], img.shape[1],
                                     _p(out), img.shape[1])
        return out

    def brief(self, blur, x, y, angle):
        blur = np.ascontiguousarray(blur, dtype=np.uint8)
        d = np.zeros(32, dtype=np.uint8)
        self.L.orbref_brief(_p(blur), blur.shape[1], x, y, angle, _p(d))
        return d

    def sincos_deg(self, angle):
        c, s = C.c_float(), C.c_float()
        self.L.orbref_sincos_deg(angle, C.byref(c), C.byref(s))
        return c.value, s.value

    def extract(self, img, cap=None):
        """operator(): returns (keypoints[KP_DTYPE], descriptors[n,32], per-level counts)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        if cap is None:
            cap = 4 * self.cfg.n_features + 64
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        counts = (C.c_int * MAX_LEVELS)()
        n = self.L.orbref_extract(C.byref(self.cfg), _p(img), w, h, w, _p(kps), _p(desc), cap, counts)
        if n < 0:
            raise RuntimeError("oracle: capacity too small")
        return kps[:n].copy(), desc[:n].copy(), list(counts[: self.n_levels])


# -- matcher helpers ---------------------------------------------------------
def hamming(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return lib().orbref_hamming(_p(a), _p(b))


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, dtype=np.int32)
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    lib().orbref_three_maxima(_p(sizes), len(sizes), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


def _fv(csr):
    node_ids, offsets, indices = csr
    node_ids = np.ascontiguousarray(node_ids, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    f = Fv(len(node_ids), node_ids.ctypes.data, offsets.ctypes.data, indices.ctypes.data)
    f._keep = (node_ids, offsets, indices)
    return f


def search_by_bow(nn_ratio, check_orientation, desc1, angle1, kf_mp_ok, fv1, desc2, angle2, frame_mp, fv2):
    desc1 = np.ascontiguousarray(desc1, dtype=np.uint8)
    desc2 = np.ascontiguousarray(desc2, dtype=np.uint8)
    angle1 = np.ascontiguousarray(angle1, dtype=np.float32)
    angle2 = np.ascontiguousarray(angle2, dtype=np.float32)
    kf_mp_ok = np.ascontiguousarray(kf_mp_ok, dtype=np.uint8)
    frame_mp = np.ascontiguousarray(frame_mp, dtype=np.int32).copy()
    f1, f2 = _fv(fv1), _fv(fv2)
    n = lib().orbref_search_by_bow(nn_ratio, int(check_orientation), _p(desc1), _p(angle1), _p(kf_mp_ok),
                                   len(desc1), C.byref(f1), _p(desc2), _p(angle2), _p(frame_mp), len(desc2),
                                   C.byref(f2))
    return n, frame_mp


def search_for_triangulation(check_orientation, desc1, angle1, has_mp1, fv1, desc2, angle2, has_mp2, fv2):
    desc1 = np.ascontiguousarray(desc1, dtype=np.uint8)
    desc2 = np.ascontiguousarray(desc2, dtype=np.uint8)
    angle1 = np.ascontiguousarray(angle1, dtype=np.float32)
    angle2 = np.ascontiguousarray(angle2, dtype=np.float32)
    has_mp1 = np.ascontiguousarray(has_mp1, dtype=np.uint8)
    has_mp2 = np.ascontiguousarray(has_mp2, dtype=np.uint8)
    m12 = np.full(len(desc1), -1, dtype=np.int32)
    f1, f2 = _fv(fv1), _fv(fv2)
    n = lib().orbref_search_for_triangulation(int(check_orientation), _p(desc1), _p(angle1), _p(has_mp1),
                                              len(desc1), C.byref(f1), _p(desc2), _p(angle2), _p(has_mp2),
                                              len(desc2), C.byref(f2), _p(m12))
    return n, m12


def search_for_initialization(nn_ratio, check_orientation, kps1, desc1, kps2, desc2, img_w, img_h, prematched,
                              window_size=100):
    kps1 = np.ascontiguousarray(kps1, dtype=KP_DTYPE)
    kps2 = np.ascontiguousarray(kps2, dtype=KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, dtype=np.uint8)
    desc2 = np.ascontiguousarray(desc2, dtype=np.uint8)
    pre = np.ascontiguousarray(prematched, dtype=np.float32).copy()
    m12 = np.full(len(kps1), -1, dtype=np.int32)
    n = lib().orbref_search_for_initialization(nn_ratio, int(check_orientation), _p(kps1), _p(desc1), len(kps1),
                                               _p(kps2), _p(desc2), len(kps2), img_w, img_h, _p(pre), _p(m12),
                                               window_size)
    return n, m12, pre


def features_in_area(kps, img_w, img_h, x, y, r, min_level, max_level):
    kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    g = lib().orbref_grid_build(_p(kps), len(kps), img_w, img_h)
    out = np.zeros(max(len(kps), 1), dtype=np.int32)
    n = lib().orbref_features_in_area(g, _p(kps), x, y, r, min_level, max_level, _p(out), len(out))
    lib().orbref_grid_free(g)
    return out[:n]


def frame_post(width, height, fx, fy, cx, cy, dist, kps, undistort=True, size_scale=None):
    """Frame.cpp:24-51: returns (raw with scaled size, undistorted, cell_start, cell_items)."""
    cam = Camera(width=width, height=height, fx=fx, fy=fy, cx=cx, cy=cy, n_dist=len(dist), undistort=int(undistort))
    for i, v in enumerate(dist):
        cam.dist[i] = v
    if size_scale is not None:
        size_scale = np.ascontiguousarray(size_scale, dtype=np.float32)
        assert size_scale.shape == (height, width)
        cam.size_scale = size_scale.ctypes.data
    raw = np.ascontiguousarray(kps, dtype=KP_DTYPE).copy()
    un = np.zeros_like(raw)
    L = lib()
    L.orbref_frame_post(C.byref(cam), _p(raw), len(raw), _p(un))

    class _Grid(C.Structure):
        _fields_ = [("cols", C.c_int), ("rows", C.c_int), ("img_w", C.c_int), ("img_h", C.c_int),
                    ("cell_start", C.POINTER(C.c_int32)), ("cell_items", C.POINTER(C.c_int32))]
    g = L.orbref_grid_build(_p(un), len(un), width, height)
    gs = C.cast(g, C.POINTER(_Grid)).contents
    nc = gs.cols * gs.rows
    start = np.ctypeslib.as_array(gs.cell_start, (nc + 1,)).copy()
    items = np.ctypeslib.as_array(gs.cell_items, (max(len(un), 1),))[: start[-1]].copy()
    L.orbref_grid_free(g)
    return raw, un, start, items


def search_by_projection_frame(check_orientation, q_desc, q_xy, q_radius, q_octave, q_angle, q_ok, kps2, desc2, img_w,
                               img_h, frame_mp):
    qd = np.ascontiguousarray(q_desc, dtype=np.uint8)
    qx = np.ascontiguousarray(q_xy, dtype=np.float32)
    qr = np.ascontiguousarray(q_radius, dtype=np.float32)
    qo = np.ascontiguousarray(q_octave, dtype=np.int32)
    qa = np.ascontiguousarray(q_angle, dtype=np.float32)
    qk = np.ascontiguousarray(q_ok, dtype=np.uint8)
    k2 = np.ascontiguousarray(kps2, dtype=KP_DTYPE)
    d2 = np.ascontiguousarray(desc2, dtype=np.uint8)
    mp = np.ascontiguousarray(frame_mp, dtype=np.int32).copy()
    n = lib().orbref_search_by_projection_frame(int(check_orientation), _p(qd), _p(qx), _p(qr), _p(qo), _p(qa), _p(qk),
                                                len(qd), _p(k2), _p(d2), len(k2), img_w, img_h, _p(mp))
    return n, mp


def search_by_projection_points(nn_ratio, q_desc, q_xy, q_radius, q_level, q_ok, kps2, desc2, img_w, img_h, frame_mp):
    qd = np.ascontiguousarray(q_desc, dtype=np.uint8)
    qx = np.ascontiguousarray(q_xy, dtype=np.float32)
    qr = np.ascontiguousarray(q_radius, dtype=np.float32)
    ql = np.ascontiguousarray(q_level, dtype=np.int32)
    qk = np.ascontiguousarray(q_ok, dtype=np.uint8)
    k2 = np.ascontiguousarray(kps2, dtype=KP_DTYPE)
    d2 = np.ascontiguousarray(desc2, dtype=np.uint8)
    mp = np.ascontiguousarray(frame_mp, dtype=np.int32).copy()
    cnt = np.zeros(3, np.int32)
    n = lib().orbref_search_by_projection_points(nn_ratio, _p(qd), _p(qx), _p(qr), _p(ql), _p(qk), len(qd), _p(k2),
                                                 _p(d2), len(k2), img_w, img_h, _p(mp), _p(cnt))
    return n, mp, tuple(cnt.tolist())


def search_fuse(q_desc, q_xy, q_radius, q_level, q_ok, kps, desc, img_w, img_h, sigma2):
    """Per-point core of the static fuse SearchByProjection (ORBMatcher.cpp:524-592): (best_idx, best_dist, n_found)."""
    qd = np.ascontiguousarray(q_desc, dtype=np.uint8)
    qx = np.ascontiguousarray(q_xy, dtype=np.float32)
    qr = np.ascontiguousarray(q_radius, dtype=np.float32)
    ql = np.ascontiguousarray(q_level, dtype=np.int32)
    qk = np.ascontiguousarray(q_ok, dtype=np.uint8)
    k = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    d = np.ascontiguousarray(desc, dtype=np.uint8)
    s2 = np.ascontiguousarray(sigma2, dtype=np.float32)
    bi = np.full(len(qd), -1, np.int32)
    bd = np.zeros(len(qd), np.int32)
    n = lib().orbref_search_fuse(_p(qd), _p(qx), _p(qr), _p(ql), _p(qk), len(qd), _p(k), _p(d), len(k), img_w, img_h,
                                 _p(s2), _p(bi), _p(bd))
    return bi, bd, n


def parse_vocabulary_text(path):
    """TemplatedVocabulary::loadFromTextFile (:1338-1420) in Python; the trailing empty line does not become a node
    (the reference turns it into a phantom root child with an uninitialised descriptor -- see include/orbv.h)."""
    with open(path) as f:
        k, L, scoring, weighting = (int(t) for t in f.readline().split())
        parent, leaf, desc, weight = [0], [0], [[0] * 32], [0.0]
        for line in f:
            t = line.split()
            if not t:
                continue
            parent.append(int(t[0])); leaf.append(1 if int(t[1]) > 0 else 0)
            desc.append([int(v) & 255 for v in t[2:34]]); weight.append(float(t[34]))
    return dict(k=k, L=L, scoring=scoring, weighting=weighting, parent=np.array(parent, np.int32),
                is_leaf=np.array(leaf, np.uint8), desc=np.array(desc, np.uint8), weight=np.array(weight, np.float64))


class Vocabulary:
    """DBoW2 vocabulary restatement (TemplatedVocabulary.h:1127-1259)."""

    def __init__(self, voc):
        self.voc = voc
        self._keep = [np.ascontiguousarray(voc["parent"], dtype=np.int32), np.ascontiguousarray(voc["is_leaf"], dtype=np.uint8),
                      np.ascontiguousarray(voc["desc"], dtype=np.uint8), np.ascontiguousarray(voc["weight"], dtype=np.float64)]
        self.h = lib().orbref_voc_build(voc["k"], voc["L"], voc["scoring"], voc["weighting"], len(self._keep[0]),
                                        *[_p(a) for a in self._keep])
        self.n_words = lib().orbref_voc_n_words(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().orbref_voc_free(self.h)
            except TypeError:  # interpreter shutdown
                pass
            self.h = None

    def transform_features(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, dtype=np.uint8)
        n = len(desc)
        word, node, w = np.zeros(n, np.uint32), np.full(n, 0xFFFFFFFF, np.uint32), np.zeros(n, np.float64)
        for i in range(n):
            lib().orbref_voc_transform_feature(self.h, desc[i].ctypes.data, levelsup, word[i:].ctypes.data,
                                               w[i:].ctypes.data, node[i:].ctypes.data)
        return word, node, w

    def transform(self, desc, levelsup=4):
        """returns (bow_ids, bow_vals, (fv_nodes, fv_off, fv_idx))"""
        desc = np.ascontiguousarray(desc, dtype=np.uint8)
        n = len(desc)
        m = max(n, 1)
        bi, bv = np.zeros(m, np.uint32), np.zeros(m, np.float64)
        fn, fo, fi = np.zeros(m, np.uint32), np.zeros(m + 1, np.int32), np.zeros(m, np.uint32)
        nw, nf = C.c_int(), C.c_int()
        lib().orbref_voc_transform(self.h, _p(desc), n, levelsup, _p(bi), _p(bv), C.addressof(nw), _p(fn), _p(fo), _p(fi),
                                   C.addressof(nf))
        return bi[: nw.value], bv[: nw.value], (fn[: nf.value], fo[: nf.value + 1], fi[: fo[nf.value]])


def distinctive_descriptor(desc):
    """MapPoint::computeDescriptor (MapPoint.cpp:103-152) for one map point."""
    desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
    return lib().orbref_distinctive_descriptor(_p(desc), len(desc))


def best2(a, b):
    """dense best / second-best of every row of a among the rows of b (ORBMatcher.cpp:148-162)"""
    a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 32)
    bi, bd, sd = np.zeros(len(a), np.int32), np.zeros(len(a), np.uint16), np.zeros(len(a), np.uint16)
    lib().orbref_best2(_p(a), len(a), _p(b), len(b), _p(bi), _p(bd), _p(sd))
    return bi, bd, sd
