"""Python mirror of the optional local-BA linearisation kernels (include/orbba.h).

One call = what g2o does for the monocular projection edges in one LM iteration of
Optimize::localBundleAdjustment (reference modules/Backend/Optimize.cpp:892-893): residuals, analytic
Jacobians (G2oTypes.cpp:36-47), Huber weights and the block J^T W J / -J^T W e accumulation, in double.
"""
import ctypes as C

import numpy as np

from . import _lib


def set_variant(name, value):
    """orbba_set_variant (per process): "chol" = "lds" | "global", "pose_lds" = edges of a frame staged in LDS (0 .. 3000)."""
    L = _lib.lib()
    L.orbba_set_variant.restype = C.c_int
    L.orbba_set_variant.argtypes = [C.c_int, C.c_int]
    which = {"chol": 0, "pose_lds": 1}[name]
    _lib.check(L.orbba_set_variant(which, int({"lds": 0, "global": 1}.get(value, value))))


class _Problem(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("huber_delta", C.c_double), ("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("pose_R", C.c_void_p), ("pose_t", C.c_void_p), ("pose_fixed", C.c_void_p), ("points", C.c_void_p),
                ("edge_pose", C.c_void_p), ("edge_point", C.c_void_p), ("edge_z", C.c_void_p),
                ("edge_inv_sigma2", C.c_void_p), ("camera_model", C.c_int32), ("fisheye_k", C.c_double * 4)]


def _cam_tail(cam):
    """cam = (fx, fy, cx, cy) -> Pinhole; (fx, fy, cx, cy, k1, k2, k3, k4) -> Fisheye (Kannala-Brandt, Fisheye.cpp)"""
    if len(cam) == 8:
        return 1, (C.c_double * 4)(*[float(np.float32(v)) for v in cam[4:]])  # the reference keeps them as float
    assert len(cam) == 4
    return 0, (C.c_double * 4)(0, 0, 0, 0)


class _Result(C.Structure):
    _fields_ = [("chi2", C.c_void_p), ("error", C.c_void_p), ("H_pp", C.c_void_p), ("b_p", C.c_void_p),
                ("H_ll", C.c_void_p), ("b_l", C.c_void_p), ("H_lp", C.c_void_p), ("kernel_ms", C.c_float)]


HUBER_MONO = float(np.sqrt(np.float32(5.991)))  # thHuberMono = sqrtf(5.991) (Optimize.cpp:857)


def linearize(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2,
              huber_delta=HUBER_MONO, device=-1):
    """Returns dict(chi2, error, H_pp, b_p, H_ll, b_l, H_lp, kernel_ms)."""
    L = _lib.lib()
    fn = L.orbba_linearize
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(_Problem), C.POINTER(_Result), C.c_int]
    f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    R, t, P = f8(pose_R).reshape(-1, 9), f8(pose_t).reshape(-1, 3), f8(points).reshape(-1, 3)
    fix = np.ascontiguousarray(pose_fixed, dtype=np.uint8)
    ep = np.ascontiguousarray(edge_pose, dtype=np.int32)
    el = np.ascontiguousarray(edge_point, dtype=np.int32)
    z, w = f8(edge_z).reshape(-1, 2), f8(edge_inv_sigma2)
    npz, nl, ne = len(R), len(P), len(ep)
    out = {"chi2": np.zeros(ne), "error": np.zeros((ne, 2)), "H_pp": np.zeros((npz, 6, 6)), "b_p": np.zeros((npz, 6)),
           "H_ll": np.zeros((nl, 3, 3)), "b_l": np.zeros((nl, 3)), "H_lp": np.zeros((ne, 3, 6))}
    vp = lambda a: a.ctypes.data  # noqa: E731
    prob = _Problem(cam[0], cam[1], cam[2], cam[3], huber_delta, npz, nl, ne, vp(R), vp(t), vp(fix), vp(P), vp(ep), vp(el),
                    vp(z), vp(w), *_cam_tail(cam))
    res = _Result(vp(out["chi2"]), vp(out["error"]), vp(out["H_pp"]), vp(out["b_p"]), vp(out["H_ll"]), vp(out["b_l"]),
                  vp(out["H_lp"]), 0.0)
    _lib.check(fn(C.byref(prob), C.byref(res), device))
    out["kernel_ms"] = float(res.kernel_ms)
    return out


class _LmOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("max_trials", C.c_int32), ("tau", C.c_double),
                ("good_step_lower", C.c_double), ("good_step_upper", C.c_double), ("user_lambda_init", C.c_double),
                ("edge_active", C.c_void_p)]


class _LmResult(C.Structure):
    _fields_ = [("pose_R", C.c_void_p), ("pose_t", C.c_void_p), ("points", C.c_void_p), ("chi2", C.c_void_p),
                ("iterations", C.c_int32), ("trials", C.c_int32), ("lam", C.c_double), ("chi2_initial", C.c_double),
                ("chi2_final", C.c_double), ("device_ms", C.c_float)]


def _problem(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2, huber_delta):
    f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    keep = [f8(pose_R).reshape(-1, 9), f8(pose_t).reshape(-1, 3), np.ascontiguousarray(pose_fixed, dtype=np.uint8),
            f8(points).reshape(-1, 3), np.ascontiguousarray(edge_pose, dtype=np.int32),
            np.ascontiguousarray(edge_point, dtype=np.int32), f8(edge_z).reshape(-1, 2), f8(edge_inv_sigma2)]
    prob = _Problem(cam[0], cam[1], cam[2], cam[3], huber_delta, len(keep[0]), len(keep[3]), len(keep[4]),
                    *[a.ctypes.data for a in keep], *_cam_tail(cam))
    return prob, keep


def _lm_out(prob):
    out = {"pose_R": np.zeros((prob.n_poses, 3, 3)), "pose_t": np.zeros((prob.n_poses, 3)),
           "points": np.zeros((prob.n_points, 3)), "chi2": np.zeros(prob.n_edges)}
    res = _LmResult(out["pose_R"].ctypes.data, out["pose_t"].ctypes.data, out["points"].ctypes.data,
                    out["chi2"].ctypes.data)
    return out, res


def _lm_finish(out, res):
    out.update(iterations=res.iterations, trials=res.trials, lam=res.lam, chi2_initial=res.chi2_initial,
               chi2_final=res.chi2_final, device_ms=float(res.device_ms))
    return out


def optimize(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2,
             huber_delta=HUBER_MONO, iterations=5, edge_active=None, device=-1):
    """optimizer.optimize(iterations) of Optimize::localBundleAdjustment's graph (Optimize.cpp:811-893) on the GPU."""
    L = _lib.lib()
    fn = L.orbba_optimize
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(_Problem), C.POINTER(_LmOptions), C.POINTER(_LmResult), C.c_int]
    prob, keep = _problem(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2, huber_delta)
    opt = _LmOptions(max_iterations=iterations)
    if edge_active is not None:
        act = np.ascontiguousarray(edge_active, dtype=np.uint8)
        opt.edge_active = act.ctypes.data
    out, res = _lm_out(prob)
    _lib.check(fn(C.byref(prob), C.byref(opt), C.byref(res), device))
    return _lm_finish(out, res)


def local_bundle_adjustment(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2,
                            huber_delta=HUBER_MONO, device=-1):
    """Optimize.cpp:892-922: optimize(5) with Huber, drop the chi2 > 5.991 edges and the kernel, optimize(10); the
    returned dict also holds `outlier` (the observations the reference erases at :924-935)."""
    L = _lib.lib()
    fn = L.orbba_local_bundle_adjustment
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(_Problem), C.POINTER(_LmResult), C.c_void_p, C.c_int]
    prob, keep = _problem(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2, huber_delta)
    out, res = _lm_out(prob)
    outlier = np.zeros(prob.n_edges, np.uint8)
    _lib.check(fn(C.byref(prob), C.byref(res), outlier.ctypes.data, device))
    out["outlier"] = outlier.astype(bool)
    return _lm_finish(out, res)


class _PoseProblem(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("huber_delta", C.c_double),
                ("n_frames", C.c_int32), ("rounds", C.c_int32), ("iterations", C.c_int32), ("edge_off", C.c_void_p),
                ("pose_R", C.c_void_p), ("pose_t", C.c_void_p), ("points", C.c_void_p), ("edge_z", C.c_void_p),
                ("edge_inv_sigma2", C.c_void_p), ("camera_model", C.c_int32), ("fisheye_k", C.c_double * 4)]


class _PoseResult(C.Structure):
    _fields_ = [("pose_R", C.c_void_p), ("pose_t", C.c_void_p), ("inlier", C.c_void_p), ("n_inliers", C.c_void_p),
                ("chi2", C.c_void_p), ("kernel_ms", C.c_float)]


def pose_optimize_batch(cam, pose_R, pose_t, edge_off, points, edge_z, edge_inv_sigma2, huber_delta=HUBER_MONO, device=-1):
    """Optimize::poseOptimize (Optimize.cpp:444-545) for many frames at once; frame f owns edges
    edge_off[f]:edge_off[f+1].  Returns dict(pose_R, pose_t, inlier, n_inliers, chi2, kernel_ms)."""
    L = _lib.lib()
    fn = L.orbba_pose_optimize_batch
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(_PoseProblem), C.POINTER(_PoseResult), C.c_int]
    f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    R, t = f8(pose_R).reshape(-1, 9), f8(pose_t).reshape(-1, 3)
    off = np.ascontiguousarray(edge_off, dtype=np.int32)
    P, z, w = f8(points).reshape(-1, 3), f8(edge_z).reshape(-1, 2), f8(edge_inv_sigma2).reshape(-1)
    B, ne = len(R), int(off[-1])
    out = {"pose_R": np.zeros((B, 3, 3)), "pose_t": np.zeros((B, 3)), "inlier": np.zeros(max(ne, 1), np.uint8),
           "n_inliers": np.zeros(B, np.int32), "chi2": np.zeros(max(ne, 1))}
    vp = lambda a: a.ctypes.data  # noqa: E731
    prob = _PoseProblem(cam[0], cam[1], cam[2], cam[3], huber_delta, B, 0, 0, vp(off), vp(R), vp(t), vp(P), vp(z), vp(w), *_cam_tail(cam))
    res = _PoseResult(vp(out["pose_R"]), vp(out["pose_t"]), vp(out["inlier"]), vp(out["n_inliers"]), vp(out["chi2"]), 0.0)
    _lib.check(fn(C.byref(prob), C.byref(res), device))
    out["inlier"] = out["inlier"][:ne].astype(bool)
    out["chi2"] = out["chi2"][:ne]
    out["kernel_ms"] = float(res.kernel_ms)
    return out


def pose_edges_device(n2, nq, d_frame_mp, d_kps, d_q_points, d_edge_off, d_points, d_z, d_w, d_edge_kp=None, stream=None):
    """orbba_pose_edges_device on torch device tensors: poseOptimize's edges from a device-resident frame's frame_mp."""
    import torch
    L = _lib.lib()
    fn = L.orbba_pose_edges_device
    fn.restype, fn.argtypes = C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 9
    st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    _lib.check(fn(n2, nq, d_frame_mp.data_ptr(), d_kps.data_ptr(), d_q_points.data_ptr(), d_edge_off.data_ptr(), d_points.data_ptr(),
                  d_z.data_ptr(), d_w.data_ptr(), d_edge_kp.data_ptr() if d_edge_kp is not None else None, st))


def pose_optimize_batch_device(cam, d_R, d_t, d_edge_off, d_points, d_z, d_w, d_R_out, d_t_out, d_inlier, d_n_inliers, d_chi2,
                               n_frames=1, huber_delta=HUBER_MONO, stream=None):
    """orbba_pose_optimize_batch_device: every array a torch device tensor (float64 / int32 / uint8), nothing copied."""
    import torch
    L = _lib.lib()
    fn = L.orbba_pose_optimize_batch_device
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(_PoseProblem), C.POINTER(_PoseResult), C.c_void_p]
    st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    prob = _PoseProblem(cam[0], cam[1], cam[2], cam[3], huber_delta, n_frames, 0, 0, d_edge_off.data_ptr(), d_R.data_ptr(),
                        d_t.data_ptr(), d_points.data_ptr(), d_z.data_ptr(), d_w.data_ptr(), *_cam_tail(cam))
    res = _PoseResult(d_R_out.data_ptr(), d_t_out.data_ptr(), d_inlier.data_ptr(), d_n_inliers.data_ptr(), d_chi2.data_ptr(), 0.0)
    _lib.check(fn(C.byref(prob), C.byref(res), st))
