"""Python mirror of the reference's ORBExtractor over the C ABI (include/orbx.h).

Interface follows modules/ORB/ORBExtractor.h:27-122 of the reference: same
constructor arguments, ``__call__(image) -> (keypoints, descriptors)`` standing
for ``operator()(image, keyPoints, descriptors)``, and the static scale-table
getters.  All pixel work happens in the HIP library; this file only marshals
numpy arrays / device pointers.
"""
import ctypes as C

import numpy as np

from . import _lib

# cv::KeyPoint layout (28 bytes)
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
STAGES = ("resize", "fast", "blur", "octree", "orient", "desc")
# kernel-choice switches of include/orbx.h (ORBX_VAR_*): name -> (index, named values)
VARIANTS = {
    "fast": (0, {"auto": 0, "cells": 1, "strips": 2}),
    "blur": (1, {"valu": 0, "auto": 1, "mfma": 2}),
    "resize_lds": (2, {"never": 0, "auto": 1, "always": 2}),
    "resize2": (3, {"never": 0, "auto": 1, "always": 2}),
    "side_blur": (4, {}),
    "early_fast": (5, {"auto": -1}),
    "split_level0": (6, {}),
    "streams": (7, {}),
    "zero_copy": (8, {}),
    "desc": (9, {"auto": 0, "separate": 1, "fused": 2}),
    "fast_cell_group": (10, {}),
}


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class ORBExtractor:
    """ORBExtractor(nFeatures=1000, scaleFactor=1.2, nLevels=8, iniThFast=20, minThFast=10)
    (reference modules/ORB/ORBExtractor.h:29-30)."""

    def __init__(self, nFeatures=1000, scaleFactor=1.2, nLevels=8, iniThFast=20, minThFast=10, *, max_width=0,
                 max_height=0, max_batch=1, blur_variant=0, device=-1, variants=None, _requota_of=None):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        if _requota_of is not None:
            _lib.check(self._L.orbx_create_requota(_requota_of._h, nFeatures, C.byref(self._h)))
        else:
            cfg = _lib.OrbxCfg(nFeatures, scaleFactor, nLevels, iniThFast, minThFast, max_width, max_height,
                               max_batch, blur_variant, device)
            _lib.check(self._L.orbx_create(C.byref(cfg), C.byref(self._h)))
        n = C.c_int()
        self._sf = np.zeros(_lib.MAX_LEVELS, np.float32)
        self._isf = np.zeros(_lib.MAX_LEVELS, np.float32)
        self._ss = np.zeros(_lib.MAX_LEVELS, np.float32)
        self._iss = np.zeros(_lib.MAX_LEVELS, np.float32)
        self._quotas = np.zeros(_lib.MAX_LEVELS, np.int32)
        self._umax = np.zeros(16, np.int32)
        lsf = C.c_float()
        _lib.check(self._L.orbx_tables(self._h, C.byref(n), _vp(self._sf), _vp(self._isf), _vp(self._ss),
                                       _vp(self._iss), C.byref(lsf), _vp(self._quotas), _vp(self._umax)))
        self.n_levels = n.value
        self.n_features = nFeatures
        self._lsf = lsf.value
        for k, v in (variants or {}).items():
            self.set_variant(k, v)

    def set_variant(self, name, value):
        """orbx_set_variant: pick a kernel / stream layout per handle (see VARIANTS; a value is an int or one of the names)."""
        idx, names = VARIANTS[name]
        _lib.check(self._L.orbx_set_variant(self._h, idx, names.get(value, value) if isinstance(value, str) else int(value)))

    def get_variant(self, name):
        v = C.c_int()
        _lib.check(self._L.orbx_get_variant(self._h, VARIANTS[name][0], C.byref(v)))
        return v.value

    @classmethod
    def requota(cls, nFeatures, other):
        """ORBExtractor(int nFeatures, const ORBExtractor&) (reference ORBExtractor.cpp:477-493)."""
        return cls(nFeatures, _requota_of=other)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.orbx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- static getters of the reference (ORBExtractor.h:44-86) -----------------
    def getScaleFactor(self, level=0):
        return float(self._sf[level])

    def getLogScaleFactor(self):
        return self._lsf

    def getMaxScaleFactor(self):
        return float(self._sf[self.n_levels - 1])

    def getScaleFactors(self):
        return self._sf[: self.n_levels].copy()

    def getInvScaleFactor(self, level):
        return float(self._isf[level])

    def getInvScaleFactors(self):
        return self._isf[: self.n_levels].copy()

    def getNumLevels(self):
        return self.n_levels

    def getSquareSigmas(self):
        return self._ss[: self.n_levels].copy()

    def getSquareSigma(self, level):
        return float(self._ss[level])

    def getInvSquareSigma(self, level):
        return float(self._iss[level])

    def features_per_level(self):
        return self._quotas[: self.n_levels].copy()

    def u_max(self):
        return self._umax.copy()

    def level_size(self, w, h, level):
        lw, lh = C.c_int(), C.c_int()
        _lib.check(self._L.orbx_level_size(self._h, w, h, level, C.byref(lw), C.byref(lh)))
        return lw.value, lh.value

    def max_keypoints(self, w, h):
        n = self._L.orbx_max_keypoints(self._h, w, h)
        if n < 0:
            _lib.check(n)
        return n

    # -- operator() ----------------------------------------------------------------
    def __call__(self, image):
        """operator()(image, keyPoints, descriptors): returns (keypoints[KP_DTYPE], descriptors[n,32] u8).
        An empty image or zero keypoints returns empty arrays (the reference leaves its outputs untouched)."""
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise TypeError("image must be 8UC1 (2-D uint8)")  # reference asserts CV_8UC1 (:499)
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        _lib.check(self._L.orbx_extract(self._h, _vp(image), w, h, image.strides[0], _vp(kps), _vp(desc), cap,
                                        C.byref(n)))
        return kps[: n.value].copy(), desc[: n.value].copy()

    def extract_batch(self, images):
        """Batch of equally sized frames (n, h, w) u8 on the host -> list of (keypoints, descriptors)."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        nf, h, w = images.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros((nf, cap), KP_DTYPE)
        desc = np.zeros((nf, cap, 32), np.uint8)
        counts = np.zeros(nf, np.int32)
        _lib.check(self._L.orbx_extract_batch(self._h, _vp(images), nf, w, h, w, h * w, _vp(kps), _vp(desc), cap,
                                              _vp(counts)))
        return [(kps[f, : counts[f]].copy(), desc[f, : counts[f]].copy()) for f in range(nf)]

    def extract_batch_device(self, d_imgs_ptr, n_frames, w, h, stride, frame_stride, d_kp_ptr, d_desc_ptr, cap,
                             d_n_ptr, stream=None):
        """Everything already in HBM (raw device pointers as ints); enqueues without synchronising."""
        _lib.check(self._L.orbx_extract_batch_device(self._h, d_imgs_ptr, n_frames, w, h, stride, frame_stride,
                                                     d_kp_ptr, d_desc_ptr, cap, d_n_ptr, _lib.stream_arg(stream)))

    def stream_wait_fast(self, stream):
        """`stream` (raw hipStream_t as int) waits for the FAST stage of the last enqueued batch (orbx_stream_wait_fast)."""
        _lib.check(self._L.orbx_stream_wait_fast(self._h, stream))

    def synchronize(self):
        _lib.check(self._L.orbx_synchronize(self._h))

    def host_register(self, array):
        """Page-lock a long-lived frame buffer (orbx_host_register): later calls copy frames out of it without blocking."""
        _lib.check(self._L.orbx_host_register(_vp(array), array.nbytes))

    def host_unregister(self, array):
        _lib.check(self._L.orbx_host_unregister(_vp(array)))

    # -- stage taps (parity tests) -----------------------------------------------
    def tap_level(self, frame, level, w, h, blurred=False):
        lw, lh = self.level_size(w, h, level)
        out = np.zeros((lh, lw), np.uint8)
        _lib.check(self._L.orbx_tap_level(self._h, frame, level, int(blurred), _vp(out), out.size))
        return out

    def tap_candidates(self, frame, level, cap=1 << 20):
        xs = np.zeros(cap, np.uint16)
        ys = np.zeros(cap, np.uint16)
        rs = np.zeros(cap, np.uint8)
        n = C.c_int()
        _lib.check(self._L.orbx_tap_candidates(self._h, frame, level, _vp(xs), _vp(ys), _vp(rs), cap, C.byref(n)))
        return xs[: n.value].copy(), ys[: n.value].copy(), rs[: n.value].copy()

    def tap_level_counts(self, frame):
        c = np.zeros(_lib.MAX_LEVELS, np.int32)
        _lib.check(self._L.orbx_tap_level_counts(self._h, frame, _vp(c)))
        return c[: self.n_levels].copy()

    def tap_sincos(self, angles_deg):
        """(cos, sin) pairs of the descriptor rotation as the device evaluates them (orbx_tap_sincos)."""
        a = np.ascontiguousarray(angles_deg, np.float32)
        out = np.zeros((a.size, 2), np.float32)
        _lib.check(self._L.orbx_tap_sincos(self._h, _vp(a), a.size, _vp(out)))
        return out

    def set_stage_timing(self, enable=True):
        _lib.check(self._L.orbx_set_stage_timing(self._h, int(enable)))

    def fast_time_in_step_ms(self):
        """Timing mode 2 (set_stage_timing(2)): FAST's launches of the last call, timed on their own streams."""
        ms, n = C.c_float(), C.c_int()
        _lib.check(self._L.orbx_fast_times_in_step_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stage_times_in_step_ms(self):
        """Timing mode 2: every stage of the last call, timed by events on the streams its kernels were launched on."""
        ms = np.zeros(len(STAGES), np.float32)
        _lib.check(self._L.orbx_stage_times_in_step_ms(self._h, _vp(ms)))
        return dict(zip(STAGES, ms.tolist()))

    def stage_times_ms(self):
        ms = np.zeros(len(STAGES), np.float32)
        _lib.check(self._L.orbx_stage_times_ms(self._h, _vp(ms)))
        return dict(zip(STAGES, ms.tolist()))
