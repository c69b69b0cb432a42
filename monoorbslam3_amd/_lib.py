"""Loader for the in-tree HIP library (monoorbslam3_amd/lib/liborbx.so).

There is no Python or CPU fallback: if the shared library is missing or a symbol
is absent the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liborbx.so")

MAX_LEVELS = 16


class OrbxCfg(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("max_batch", C.c_int32), ("blur_variant", C.c_int32),
                ("device", C.c_int32)]


class OrbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbx error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64; if liborbx.so (linked against /opt/rocm) is loaded first, the process
    # ends up with two HIP runtimes and the second one sees no device.  Importing torch first makes both share one.
    try:
        import torch  # noqa: F401
    except Exception:  # a pure C/C++ consumer without torch is fine
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    sigs = {
        "orbx_create": (i32, [C.POINTER(OrbxCfg), C.POINTER(vp)]),
        "orbx_create_requota": (i32, [vp, i32, C.POINTER(vp)]),
        "orbx_destroy": (None, [vp]),
        "orbx_tables": (i32, [vp, C.POINTER(i32), vp, vp, vp, vp, C.POINTER(f32), vp, vp]),
        "orbx_level_size": (i32, [vp, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]),
        "orbx_max_keypoints": (i32, [vp, i32, i32]),
        "orbx_extract": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]),
        "orbx_extract_batch": (i32, [vp, vp, i32, i32, i32, i32, sz, vp, vp, i32, vp]),
        "orbx_extract_batch_device": (i32, [vp, vp, i32, i32, i32, i32, sz, vp, vp, i32, vp, vp]),
        "orbx_synchronize": (i32, [vp]),
        "orbx_host_register": (i32, [vp, sz]),
        "orbx_host_unregister": (i32, [vp]),
        "orbx_stream_wait_fast": (i32, [vp, vp]),
        "orbx_tap_level": (i32, [vp, i32, i32, i32, vp, sz]),
        "orbx_tap_candidates": (i32, [vp, i32, i32, vp, vp, vp, i32, C.POINTER(i32)]),
        "orbx_tap_level_counts": (i32, [vp, i32, vp]),
        "orbx_tap_sincos": (i32, [vp, vp, i32, vp]),
        "orbx_set_variant": (i32, [vp, i32, i32]),
        "orbx_get_variant": (i32, [vp, i32, C.POINTER(i32)]),
        "orbx_set_stage_timing": (i32, [vp, i32]),
        "orbx_fast_times_in_step_ms": (i32, [vp, C.POINTER(f32), C.POINTER(i32)]),
        "orbx_stage_times_ms": (i32, [vp, vp]),
        "orbx_stage_times_in_step_ms": (i32, [vp, vp]),
        "orbx_last_error": (C.c_char_p, []),
        "orbx_version": (C.c_char_p, []),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise OrbxError(rc, lib().orbx_last_error().decode("utf-8", "replace"))


def stream_arg(stream=None):
    """The `stream` argument of a *_device entry point for a wrapper call.  An explicit stream (raw hipStream_t as int) is
    passed through.  None follows torch's CURRENT stream when torch is loaded and the GPU is initialised -- inside
    `with torch.cuda.stream(s):` the call is enqueued on s, like the tensor operations around it -- and is NULL otherwise,
    which the C ABI runs on stream 0 (include/orbx.h, "Streams"); torch's default stream IS stream 0."""
    if stream is not None:
        return stream
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        return torch.cuda.current_stream().cuda_stream or None
    return None


def kernels_sha16():
    """First 16 hex digits of the sha256 over the kernel sources (csrc/*.hip, *.h): ties a profile to the code it measured."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
