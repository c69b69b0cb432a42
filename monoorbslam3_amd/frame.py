"""Host mirror of the reference's Frame-constructor post-processing (modules/BasicObject/Frame.cpp:24-51) over the
C ABI of include/orbf.h: key-point size scaling, undistortion and the 40-px grid index, on the GPU."""
import ctypes as C

import numpy as np

from . import _lib
from .extractor import KP_DTYPE

GRID_SIZE = 40  # Frame.h:18


class OrbfCamera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("n_dist", C.c_int32), ("dist", C.c_float * 12), ("undistort", C.c_int32),
                ("size_scale", C.c_void_p)]


_bound = False


def _L():
    global _bound
    L = _lib.lib()
    if not _bound:
        vp, i32 = C.c_void_p, C.c_int
        L.orbf_create.restype = i32
        L.orbf_create.argtypes = [C.POINTER(OrbfCamera), i32, C.POINTER(vp)]
        L.orbf_destroy.restype = None
        L.orbf_destroy.argtypes = [vp]
        L.orbf_grid_dims.restype = i32
        L.orbf_grid_dims.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        L.orbf_frame_post_device.restype = i32
        L.orbf_frame_post_device.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp]
        L.orbf_frame_post.restype = i32
        L.orbf_frame_post.argtypes = [vp, vp, i32, vp, vp, vp]
        _bound = True
    return L


class FramePost:
    """Camera(w, h, K, distCoeffs) (Camera.cpp:17-22) + the three loops of Frame::Frame that consume it."""

    def __init__(self, width, height, fx, fy, cx, cy, dist=(), undistort=True, size_scale=None, device=0):
        if len(dist) > 12:
            raise ValueError("at most 12 distortion coefficients (tilt terms are not supported)")
        cam = OrbfCamera(width=width, height=height, fx=fx, fy=fy, cx=cx, cy=cy, n_dist=len(dist),
                         undistort=int(undistort))
        for i, v in enumerate(dist):
            cam.dist[i] = v
        if size_scale is not None:
            size_scale = np.ascontiguousarray(size_scale, dtype=np.float32)
            if size_scale.shape != (height, width):
                raise ValueError("size_scale must be height x width")
            cam.size_scale = size_scale.ctypes.data
        self._h = C.c_void_p()
        _lib.check(_L().orbf_create(C.byref(cam), device, C.byref(self._h)))
        cols, rows = C.c_int(), C.c_int()
        _lib.check(_L().orbf_grid_dims(self._h, C.byref(cols), C.byref(rows)))
        self.cols, self.rows = cols.value, rows.value
        self.n_cells = self.cols * self.rows

    def close(self):
        if getattr(self, "_h", None):
            try:
                _L().orbf_destroy(self._h)
            except TypeError:  # interpreter shutdown: module globals are already gone
                pass
            self._h = None

    __del__ = close

    def __call__(self, kps):
        """One frame, host arrays: returns (raw with scaled size, undistorted, cell_start, cell_items)."""
        raw = np.ascontiguousarray(kps, dtype=KP_DTYPE).copy()
        n = len(raw)
        un = np.zeros(n, dtype=KP_DTYPE)
        start = np.zeros(self.n_cells + 1, dtype=np.int32)
        items = np.zeros(max(n, 1), dtype=np.int32)
        _lib.check(_L().orbf_frame_post(self._h, raw.ctypes.data, n, un.ctypes.data, start.ctypes.data,
                                        items.ctypes.data))
        return raw, un, start, items[: start[-1]]

    def post_device(self, n_frames, d_kp_raw, d_n, cap, d_kp_un, d_cell_start, d_cell_items, stream=None):
        """Batch on device pointers (ints); see include/orbf.h."""
        _lib.check(_L().orbf_frame_post_device(self._h, n_frames, d_kp_raw, d_n, cap, d_kp_un, d_cell_start,
                                               d_cell_items, _lib.stream_arg(stream)))
