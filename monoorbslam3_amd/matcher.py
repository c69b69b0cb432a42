"""Python mirror of the reference's ORBMatcher over the C ABI (include/orbm.h).

Same constructor as modules/ORB/ORBMatcher.h:14 (nnRatio, checkOrientation) and the
Search* entry points of the north-star path, expressed over plain arrays instead of
Frame / KeyFrame objects: a "frame" here is (descriptors[n,32], angles[n] or
keypoints, FeatureVector CSR, map-point mask).  Distances are computed by the HIP
kernels; the library's host code performs the reference's greedy resolution.
"""
import ctypes as C

import numpy as np

from . import _lib
from .extractor import KP_DTYPE

TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30


class _Fv(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_ids", C.c_void_p), ("offsets", C.c_void_p), ("indices", C.c_void_p)]


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


_sigs_done = False


def _mlib():
    global _sigs_done
    L = _lib.lib()
    if not _sigs_done:
        vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        sigs = {
            "orbm_create": (i32, [i32, C.POINTER(vp)]),
            "orbm_destroy": (None, [vp]),
            "orbm_set_variant": (i32, [vp, i32, i32]),
            "orbm_hamming_matrix": (i32, [vp, vp, i32, vp, i32, vp]),
            "orbm_hamming_matrix_device": (i32, [vp, vp, i32, vp, i32, vp, vp]),
            "orbm_best2_device": (i32, [vp, i32, vp, sz, vp, i32, vp, sz, vp, i32, vp, vp, vp, vp, vp, vp]),
            "orbm_best2": (i32, [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]),
            "orbm_hamming_csr": (i32, [vp, vp, i32, vp, i32, vp, vp, i32, vp, vp]),
            "orbm_search_by_bow": (i32, [vp, f32, i32, vp, vp, vp, i32, C.POINTER(_Fv), vp, vp, vp, i32,
                                         C.POINTER(_Fv), C.POINTER(i32)]),
            "orbm_search_for_triangulation": (i32, [vp, i32, vp, vp, vp, i32, C.POINTER(_Fv), vp, vp, vp, i32,
                                                    C.POINTER(_Fv), vp, C.POINTER(i32)]),
            "orbm_search_for_initialization": (i32, [vp, f32, i32, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, i32,
                                                     C.POINTER(i32)]),
            "orbm_search_by_projection_frame": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp,
                                                      C.POINTER(i32)]),
            "orbm_search_by_projection_points": (i32, [vp, f32, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp,
                                                       C.POINTER(i32), vp]),
            "orbm_search_fuse": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp, i32, vp, vp, C.POINTER(i32)]),
            "orbm_search_by_projection_frame_device": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32,
                                                             vp, vp, vp]),
            "orbm_search_by_projection_points_device": (i32, [vp, f32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32,
                                                              vp, vp, vp]),
            "orbm_search_by_bow_device": (i32, [vp, f32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
            "orbm_search_for_initialization_device": (i32, [vp, f32, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, i32, i32,
                                                            vp, vp, vp]),
            "orbm_search_fuse_device": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, i32, vp, vp, vp, vp]),
            "orbm_search_for_triangulation_device": (i32, [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp,
                                                           vp, vp]),
            "orbm_window_lists_device": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, i32, vp, vp, vp]),
            "orbm_distinctive_descriptors": (i32, [vp, vp, vp, i32, vp]),
            "orbm_distinctive_descriptors_device": (i32, [vp, vp, vp, i32, vp, vp]),
            "orbm_three_maxima": (None, [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _sigs_done = True
    return L


def _fv(csr):
    node_ids, offsets, indices = csr
    node_ids = np.ascontiguousarray(node_ids, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    f = _Fv(len(node_ids), node_ids.ctypes.data, offsets.ctypes.data, indices.ctypes.data)
    f._keep = (node_ids, offsets, indices)
    return f


_default = None


def _handle():
    global _default
    if _default is None:
        _default = MatcherHandle()
    return _default


class MatcherHandle:
    """One orbm_t: a HIP stream plus scratch.  Use one per host thread."""

    def __init__(self, device=-1):
        self._L = _mlib()
        self._h = C.c_void_p()
        _lib.check(self._L.orbm_create(device, C.byref(self._h)))

    def set_variant(self, name, value):
        """orbm_set_variant: "best2" = "fp4" | "i8" | "valu" (dense best / second-best kernel), "window" = "device" | "host",
        "best2_resident" = 0 | 1 | 2 (k_best2_fp4 as that many workgroups per CU walking the query blocks)."""
        which = {"best2": 0, "window": 1, "best2_resident": 2, "init_lanes": 3}[name]
        val = {"fp4": 0, "i8": 1, "valu": 2, "device": 0, "host": 1}.get(value, value)
        _lib.check(self._L.orbm_set_variant(self._h, which, int(val)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.orbm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ORBMatcher:
    """ORBMatcher(nnRatio=0.6, checkOrientation=True) (reference modules/ORB/ORBMatcher.h:14)."""

    def __init__(self, nnRatio=0.6, checkOrientation=True, handle=None):
        self.nn_ratio = float(nnRatio)
        self.be_check_orientation = bool(checkOrientation)
        self._hd = handle or _handle()
        self._L = self._hd._L

    # -- DescriptorDistance (ORBMatcher.cpp:17-31) --------------------------------
    @staticmethod
    def DescriptorDistance(a, b):
        return int(ORBMatcher.hamming_matrix(np.asarray(a, np.uint8).reshape(1, 32),
                                             np.asarray(b, np.uint8).reshape(1, 32))[0, 0])

    @staticmethod
    def hamming_matrix(a, b, handle=None):
        hd = handle or _handle()
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 32)
        out = np.zeros((len(a), len(b)), np.uint16)
        _lib.check(hd._L.orbm_hamming_matrix(hd._h, _vp(a), len(a), _vp(b), len(b), _vp(out)))
        return out

    @staticmethod
    def best2(a, b, row_ok=None, col_ok=None, handle=None):
        """(best index, best distance, second distance) per row of `a` among rows of `b`."""
        hd = handle or _handle()
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 32)
        bi = np.zeros(len(a), np.int32)
        bd = np.zeros(len(a), np.uint16)
        sd = np.zeros(len(a), np.uint16)
        ro = None if row_ok is None else np.ascontiguousarray(row_ok, dtype=np.uint8)
        co = None if col_ok is None else np.ascontiguousarray(col_ok, dtype=np.uint8)
        _lib.check(hd._L.orbm_best2(hd._h, _vp(a), len(a), _vp(b), len(b), None if ro is None else _vp(ro),
                                    None if co is None else _vp(co), _vp(bi), _vp(bd), _vp(sd)))
        return bi, bd, sd

    @staticmethod
    def hamming_csr(a, b, q_idx, off, c_idx, handle=None):
        hd = handle or _handle()
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 32)
        q_idx = np.ascontiguousarray(q_idx, dtype=np.int32)
        off = np.ascontiguousarray(off, dtype=np.int32)
        c_idx = np.ascontiguousarray(c_idx, dtype=np.int32)
        out = np.zeros(int(off[-1]) if len(off) else 0, np.uint16)
        _lib.check(hd._L.orbm_hamming_csr(hd._h, _vp(a), len(a), _vp(b), len(b), _vp(q_idx), _vp(off), len(q_idx),
                                          _vp(c_idx), _vp(out)))
        return out

    @staticmethod
    def ComputeDistinctiveDescriptors(desc, off, handle=None):
        """MapPoint::computeDescriptor (MapPoint.cpp:103-152) for many map points: desc[off[g]:off[g+1]] are the
        observations of point g; returns the index (inside its group) of each point's new descriptor, -1 if none."""
        hd = handle or _handle()
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
        off = np.ascontiguousarray(off, dtype=np.int32)
        out = np.zeros(max(len(off) - 1, 0), np.int32)
        _lib.check(hd._L.orbm_distinctive_descriptors(hd._h, _vp(desc), _vp(off), len(off) - 1, _vp(out)))
        return out

    @staticmethod
    def ComputeThreeMaxima(hist_sizes):
        L = _mlib()
        s = np.ascontiguousarray(hist_sizes, dtype=np.int32)
        i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        L.orbm_three_maxima(_vp(s), len(s), C.byref(i1), C.byref(i2), C.byref(i3))
        return i1.value, i2.value, i3.value

    # -- SearchByBow (ORBMatcher.cpp:118-201) --------------------------------------
    def SearchByBow(self, kf_desc, kf_angles, kf_mp_ok, kf_fv, fr_desc, fr_angles, frame_mp, fr_fv):
        """Returns (numMatch, frame_mp'): frame_mp'[j] = key-frame feature whose MapPoint is assigned, -1 = null."""
        d1 = np.ascontiguousarray(kf_desc, dtype=np.uint8)
        d2 = np.ascontiguousarray(fr_desc, dtype=np.uint8)
        a1 = np.ascontiguousarray(kf_angles, dtype=np.float32)
        a2 = np.ascontiguousarray(fr_angles, dtype=np.float32)
        ok = np.ascontiguousarray(kf_mp_ok, dtype=np.uint8)
        mp = np.ascontiguousarray(frame_mp, dtype=np.int32).copy()
        f1, f2 = _fv(kf_fv), _fv(fr_fv)
        n = C.c_int()
        _lib.check(self._L.orbm_search_by_bow(self._hd._h, self.nn_ratio, int(self.be_check_orientation), _vp(d1),
                                              _vp(a1), _vp(ok), len(d1), C.byref(f1), _vp(d2), _vp(a2), _vp(mp),
                                              len(d2), C.byref(f2), C.byref(n)))
        return n.value, mp

    # -- SearchForTriangulation (ORBMatcher.cpp:417-522) -----------------------------
    def SearchForTriangulation(self, desc1, angles1, has_mp1, fv1, desc2, angles2, has_mp2, fv2):
        d1 = np.ascontiguousarray(desc1, dtype=np.uint8)
        d2 = np.ascontiguousarray(desc2, dtype=np.uint8)
        a1 = np.ascontiguousarray(angles1, dtype=np.float32)
        a2 = np.ascontiguousarray(angles2, dtype=np.float32)
        h1 = np.ascontiguousarray(has_mp1, dtype=np.uint8)
        h2 = np.ascontiguousarray(has_mp2, dtype=np.uint8)
        m12 = np.full(len(d1), -1, np.int32)
        f1, f2 = _fv(fv1), _fv(fv2)
        n = C.c_int()
        _lib.check(self._L.orbm_search_for_triangulation(self._hd._h, int(self.be_check_orientation), _vp(d1), _vp(a1),
                                                         _vp(h1), len(d1), C.byref(f1), _vp(d2), _vp(a2), _vp(h2),
                                                         len(d2), C.byref(f2), _vp(m12), C.byref(n)))
        return n.value, m12

    # -- SearchForInitialization (ORBMatcher.cpp:33-116) -----------------------------
    def SearchForInitialization(self, kps1, desc1, kps2, desc2, img_w, img_h, vecPreMatched, windowSize=100):
        k1 = np.ascontiguousarray(kps1, dtype=KP_DTYPE)
        k2 = np.ascontiguousarray(kps2, dtype=KP_DTYPE)
        d1 = np.ascontiguousarray(desc1, dtype=np.uint8)
        d2 = np.ascontiguousarray(desc2, dtype=np.uint8)
        pre = np.ascontiguousarray(vecPreMatched, dtype=np.float32).copy()
        m12 = np.full(len(k1), -1, np.int32)
        n = C.c_int()
        _lib.check(self._L.orbm_search_for_initialization(self._hd._h, self.nn_ratio, int(self.be_check_orientation),
                                                          _vp(k1), _vp(d1), len(k1), _vp(k2), _vp(d2), len(k2), img_w,
                                                          img_h, _vp(pre), _vp(m12), windowSize, C.byref(n)))
        return n.value, m12, pre

    # -- SearchByProjection(lastFrame | lastKF, curFrame, th) (ORBMatcher.cpp:203-348) ----------
    def SearchByProjectionFrame(self, q_desc, q_xy, q_radius, q_octave, q_angle, q_ok, kps2, desc2, img_w, img_h,
                                frame_mp):
        """Queries = features of the last frame with their MapPoint descriptor and projected position (the camera
        maths stays with the caller).  Returns (numMatch, frame_mp')."""
        qd = np.ascontiguousarray(q_desc, dtype=np.uint8)
        qx = np.ascontiguousarray(q_xy, dtype=np.float32)
        qr = np.ascontiguousarray(q_radius, dtype=np.float32)
        qo = np.ascontiguousarray(q_octave, dtype=np.int32)
        qa = np.ascontiguousarray(q_angle, dtype=np.float32)
        qk = np.ascontiguousarray(q_ok, dtype=np.uint8)
        k2 = np.ascontiguousarray(kps2, dtype=KP_DTYPE)
        d2 = np.ascontiguousarray(desc2, dtype=np.uint8)
        mp = np.ascontiguousarray(frame_mp, dtype=np.int32).copy()
        n = C.c_int()
        _lib.check(self._L.orbm_search_by_projection_frame(self._hd._h, int(self.be_check_orientation), _vp(qd), _vp(qx),
                                                           _vp(qr), _vp(qo), _vp(qa), _vp(qk), len(qd), _vp(k2), _vp(d2),
                                                           len(k2), img_w, img_h, _vp(mp), C.byref(n)))
        return n.value, mp

    # -- SearchByProjection(frame, mapPoints, th) (ORBMatcher.cpp:350-415) -----------------------
    def SearchByProjectionPoints(self, q_desc, q_xy, q_radius, q_level, q_ok, kps2, desc2, img_w, img_h, frame_mp):
        """Returns (numMatch, frame_mp', (numOutViewAndBad, fail1, fail2))."""
        qd = np.ascontiguousarray(q_desc, dtype=np.uint8)
        qx = np.ascontiguousarray(q_xy, dtype=np.float32)
        qr = np.ascontiguousarray(q_radius, dtype=np.float32)
        ql = np.ascontiguousarray(q_level, dtype=np.int32)
        qk = np.ascontiguousarray(q_ok, dtype=np.uint8)
        k2 = np.ascontiguousarray(kps2, dtype=KP_DTYPE)
        d2 = np.ascontiguousarray(desc2, dtype=np.uint8)
        mp = np.ascontiguousarray(frame_mp, dtype=np.int32).copy()
        cnt = np.zeros(3, np.int32)
        n = C.c_int()
        _lib.check(self._L.orbm_search_by_projection_points(self._hd._h, self.nn_ratio, _vp(qd), _vp(qx), _vp(qr), _vp(ql),
                                                            _vp(qk), len(qd), _vp(k2), _vp(d2), len(k2), img_w, img_h,
                                                            _vp(mp), C.byref(n), _vp(cnt)))
        return n.value, mp, tuple(cnt.tolist())

    # -- the same two searches on a device-resident frame record, greedy pass on the device -----------------------
    def SearchByProjectionDevice(self, mode, d, nq, n2, grid_cols, grid_rows, list_cap=48, stream=None):
        """mode "frame" | "points".  d: dict of torch device tensors -- q_desc [nq,32] u8, q_xy [nq,2] f32, q_radius f32,
        q_level i32 (octave / predicted level), q_angle f32 (frame only), q_ok u8, kps2 (undistorted records, u8 [*,28]),
        desc2 u8 [*,32], cell_start i32, cell_items i32, frame_mp i32 [n2] (in/out), result i32 [8] (out).  Enqueues on
        `stream`; nothing is copied or synchronised (orbm_search_by_projection_{frame,points}_device)."""
        st = _lib.stream_arg(stream)
        p = lambda k: d[k].data_ptr()  # noqa: E731
        if mode == "frame":
            _lib.check(self._L.orbm_search_by_projection_frame_device(
                self._hd._h, int(self.be_check_orientation), p("q_desc"), p("q_xy"), p("q_radius"), p("q_level"), p("q_angle"),
                p("q_ok"), nq, p("kps2"), p("desc2"), p("cell_start"), p("cell_items"), grid_cols, grid_rows, n2, list_cap,
                p("frame_mp"), p("result"), st))
        else:
            _lib.check(self._L.orbm_search_by_projection_points_device(
                self._hd._h, self.nn_ratio, p("q_desc"), p("q_xy"), p("q_radius"), p("q_level"), p("q_ok"), nq, p("kps2"),
                p("desc2"), p("cell_start"), p("cell_items"), grid_cols, grid_rows, n2, list_cap, p("frame_mp"), p("result"), st))

    def SearchForInitializationDevice(self, d, n1, n2, grid_cols, grid_rows, window=100, list_cap=768, stream=None):
        """orbm_search_for_initialization_device on torch device tensors: d = dict(kps1, desc1, kps2 (frame 2's record as
        orbf_frame_post_device leaves it), desc2, cell_start, cell_items, pre (float32 [n1, 2], in / out), matches12 (int32 [n1], out),
        result (int32 x 8, out))."""
        p = lambda k: d[k].data_ptr()  # noqa: E731
        _lib.check(self._L.orbm_search_for_initialization_device(
            self._hd._h, self.nn_ratio, int(self.be_check_orientation), p("kps1"), p("desc1"), n1, p("kps2"), p("desc2"),
            p("cell_start"), p("cell_items"), grid_cols, grid_rows, n2, p("pre"), window, list_cap, p("matches12"), p("result"),
            _lib.stream_arg(stream)))

    # -- static SearchByProjection(keyFrame, mapPoints, Map*, th): the fuse (ORBMatcher.cpp:524-592) ------------
    def SearchFuseDevice(self, d, nq, grid_cols, grid_rows, list_cap=48, stream=None):
        """orbm_search_fuse_device on torch device tensors: d = dict(q_desc, q_xy, q_radius, q_level, q_ok, kps, desc, cell_start,
        cell_items, sigma2, best_idx (int32 [nq], out), best_dist (int32 [nq], out), result (int32 x 8, out))."""
        p = lambda k: d[k].data_ptr()  # noqa: E731
        _lib.check(self._L.orbm_search_fuse_device(
            self._hd._h, p("q_desc"), p("q_xy"), p("q_radius"), p("q_level"), p("q_ok"), nq, p("kps"), p("desc"), p("cell_start"),
            p("cell_items"), grid_cols, grid_rows, p("sigma2"), list_cap, p("best_idx"), p("best_dist"), p("result"), _lib.stream_arg(stream)))

    def SearchByBowDevice(self, d, n1, n2, stream=None):
        """orbm_search_by_bow_device on torch device tensors: d = dict(desc1, kps1 (orbx_kp records), kf_mp_ok, fv1=(nodes, off, idx, n), desc2, kps2,
        frame_mp (in / out), fv2=(nodes, off, idx, n), result (int32 x 8))."""
        p = lambda t: t.data_ptr()  # noqa: E731
        _lib.check(self._hd._L.orbm_search_by_bow_device(
            self._hd._h, self.nn_ratio, int(self.be_check_orientation), p(d["desc1"]), p(d["kps1"]), p(d["kf_mp_ok"]), n1,
            p(d["fv1"][0]), p(d["fv1"][1]), p(d["fv1"][2]), p(d["fv1"][3]), p(d["desc2"]), p(d["kps2"]), p(d["frame_mp"]), n2,
            p(d["fv2"][0]), p(d["fv2"][1]), p(d["fv2"][2]), p(d["fv2"][3]), p(d["result"]), _lib.stream_arg(stream)))

    def SearchForTriangulationDevice(self, d, n1, n2, stream=None):
        """orbm_search_for_triangulation_device: d = dict(desc1, kps1, has_mp1, fv1, desc2, kps2, has_mp2, fv2, matches12, result)."""
        p = lambda t: t.data_ptr()  # noqa: E731
        _lib.check(self._hd._L.orbm_search_for_triangulation_device(
            self._hd._h, int(self.be_check_orientation), p(d["desc1"]), p(d["kps1"]), p(d["has_mp1"]), n1, p(d["fv1"][0]), p(d["fv1"][1]),
            p(d["fv1"][2]), p(d["fv1"][3]), p(d["desc2"]), p(d["kps2"]), p(d["has_mp2"]), n2, p(d["fv2"][0]), p(d["fv2"][1]),
            p(d["fv2"][2]), p(d["fv2"][3]), p(d["matches12"]), p(d["result"]), _lib.stream_arg(stream)))

    def SearchFuse(self, q_desc, q_xy, q_radius, q_level, q_ok, kps, desc, img_w, img_h, sigma2):
        """Per-point core of the fuse: (best_idx, best_dist, n_found); the observation rewiring stays with the caller."""
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
        n = C.c_int()
        _lib.check(self._L.orbm_search_fuse(self._hd._h, _vp(qd), _vp(qx), _vp(qr), _vp(ql), _vp(qk), len(qd), _vp(k),
                                            _vp(d), len(k), img_w, img_h, _vp(s2), len(s2), _vp(bi), _vp(bd), C.byref(n)))
        return bi, bd, n.value
