"""numpy (float64) restatement of the local-BA projection-edge linearisation -- TEST INFRASTRUCTURE ONLY.

Follows, per edge: EdgeSE3Project3D::computeError / linearizeOplus (reference modules/Backend/G2oTypes.h:247-251,
G2oTypes.cpp:36-47), Pinhole::project / getProjJacobian (modules/Sensor/Pinhole.cpp:28-32, :49-53), the Huber
kernel set up at modules/Backend/Optimize.cpp:857,:880-882, and g2o's BaseBinaryEdge::constructQuadraticForm
(g2o 20201223, not vendored: H_ii += J_i^T W J_i, H_ij += J_i^T W J_j, b_i -= J_i^T W e with W = rho'(chi2) Omega).
PARITY UNPINNED: neither g2o nor Eigen is available here; pinned by the numeric-difference test of J against e
and hand-checkable cases in tests/test_ba.py.
"""
import numpy as np


def hat(v):
    """lie::Hat (modules/Utils/LieAlgeBra.h): [0 -z y; z 0 -x; -y x 0] for each row of v."""
    z = np.zeros(len(v))
    return np.stack([np.stack([z, -v[:, 2], v[:, 1]], 1), np.stack([v[:, 2], z, -v[:, 0]], 1),
                     np.stack([-v[:, 1], v[:, 0], z], 1)], 1)


def project(cam, Pc):
    """camera->project(Pc): cam = (fx, fy, cx, cy) is Pinhole::project (modules/Sensor/Pinhole.cpp:28-32);
    (fx, fy, cx, cy, k1, k2, k3, k4) is Fisheye::project (modules/Sensor/Fisheye.cpp:35-49, Kannala-Brandt; the reference
    keeps dist_coeffs as float, so k is rounded to float first)."""
    fx, fy, cx, cy = cam[:4]
    X, Y, Z = Pc[:, 0], Pc[:, 1], Pc[:, 2]
    if len(cam) == 4:
        return np.stack([fx * (X / Z) + cx, fy * (Y / Z) + cy], 1)
    k = [float(np.float32(v)) for v in cam[4:8]]
    a, b = X / Z, Y / Z
    r = np.sqrt(a * a + b * b)
    theta = np.arctan(r)
    theta2 = theta * theta
    theta3 = theta * theta2
    theta5 = theta2 * theta3
    theta7 = theta2 * theta5
    theta9 = theta2 * theta7
    theta_d = theta + k[0] * theta3 + k[1] * theta5 + k[2] * theta7 + k[3] * theta9
    return np.stack([fx * theta_d * a / r + cx, fy * theta_d * b / r + cy], 1)


def proj_jacobian(cam, Pc):
    """camera->getProjJacobian(Pc) (G2oTypes.cpp:42), n x 2 x 3: Pinhole.cpp:49-53 or Fisheye.cpp:83-108."""
    fx, fy, cx, cy = cam[:4]
    X, Y, Z = Pc[:, 0], Pc[:, 1], Pc[:, 2]
    zero = np.zeros_like(X)
    if len(cam) == 4:
        return np.stack([np.stack([fx / Z, zero, -fx * X / (Z * Z)], 1), np.stack([zero, fy / Z, -fy * Y / (Z * Z)], 1)], 1)
    k = [float(np.float32(v)) for v in cam[4:8]]
    x2, y2, z2 = X * X, Y * Y, Z * Z
    r2 = x2 + y2
    r = np.sqrt(r2)
    r3 = r2 * r
    theta = np.arctan2(r, Z)
    theta2 = theta * theta
    theta3 = theta2 * theta
    theta4 = theta2 * theta2
    theta5 = theta4 * theta
    theta6 = theta2 * theta4
    theta7 = theta6 * theta
    theta8 = theta4 * theta4
    theta9 = theta8 * theta
    f = theta + theta3 * k[0] + theta5 * k[1] + theta7 * k[2] + theta9 * k[3]
    fd = 1 + 3 * k[0] * theta2 + 5 * k[1] * theta4 + 7 * k[2] * theta6 + 9 * k[3] * theta8
    j00 = fx * (fd * Z * x2 / (r2 * (r2 + z2)) + f * y2 / r3)
    j10 = fy * (fd * Z * Y * X / (r2 * (r2 + z2)) - f * Y * X / r3)
    j01 = fx * (fd * Z * Y * X / (r2 * (r2 + z2)) - f * Y * X / r3)
    j11 = fy * (fd * Z * y2 / (r2 * (r2 + z2)) + f * x2 / r3)
    j02 = -fx * fd * X / (r2 + z2)
    j12 = -fy * fd * Y / (r2 + z2)
    return np.stack([np.stack([j00, j01, j02], 1), np.stack([j10, j11, j12], 1)], 1)


def residual(cam, R, t, P, z):
    """e = z - project(R P + t) for matched rows."""
    Pc = np.einsum("eij,ej->ei", R, P) + t
    return z - project(cam, Pc), Pc


def linearize(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2, huber_delta):
    R = np.asarray(pose_R, np.float64).reshape(-1, 3, 3)[edge_pose]
    t = np.asarray(pose_t, np.float64).reshape(-1, 3)[edge_pose]
    P = np.asarray(points, np.float64).reshape(-1, 3)[edge_point]
    z = np.asarray(edge_z, np.float64).reshape(-1, 2)
    om = np.asarray(edge_inv_sigma2, np.float64)
    e, Pc = residual(cam, R, t, P, z)
    Jp = proj_jacobian(cam, Pc)
    Jl = -Jp @ R                                              # G2oTypes.cpp:44
    Jq = np.concatenate([Jp @ hat(Pc), -Jp], axis=2)          # G2oTypes.cpp:45-46
    chi2 = om * (e * e).sum(1)
    rw = np.ones_like(chi2)
    if huber_delta > 0:
        out = chi2 > huber_delta * huber_delta
        rw[out] = huber_delta / np.sqrt(chi2[out])
    W = rw * om
    fixed = np.asarray(pose_fixed, bool)[edge_pose]
    n_p, n_l = len(np.asarray(pose_fixed)), len(np.asarray(points).reshape(-1, 3))
    Hpp_e = W[:, None, None] * np.einsum("eki,ekj->eij", Jq, Jq)
    bp_e = -W[:, None] * np.einsum("eki,ek->ei", Jq, e)
    Hll_e = W[:, None, None] * np.einsum("eki,ekj->eij", Jl, Jl)
    bl_e = -W[:, None] * np.einsum("eki,ek->ei", Jl, e)
    Hlp = W[:, None, None] * np.einsum("eki,ekj->eij", Jl, Jq)
    Hpp_e[fixed] = 0
    bp_e[fixed] = 0
    Hlp[fixed] = 0
    H_pp = np.zeros((n_p, 6, 6)); b_p = np.zeros((n_p, 6)); H_ll = np.zeros((n_l, 3, 3)); b_l = np.zeros((n_l, 3))
    np.add.at(H_pp, edge_pose, Hpp_e)
    np.add.at(b_p, edge_pose, bp_e)
    np.add.at(H_ll, edge_point, Hll_e)
    np.add.at(b_l, edge_point, bl_e)
    return {"chi2": chi2, "error": e, "H_pp": H_pp, "b_p": b_p, "H_ll": H_ll, "b_l": b_l, "H_lp": Hlp, "J_point": Jl,
            "J_pose": Jq}


def se3_exp(upd):
    """g2o SE3Quat::exp (g2o 20201223 types/slam3d/se3quat.h, not vendored) for update = (omega, upsilon), as used by
    VertexSE3::oplusImpl (reference modules/Backend/G2oTypes.h:112-115): returns (R, t) of the increment."""
    w, u = np.asarray(upd[:3], np.float64), np.asarray(upd[3:], np.float64)
    th = np.sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2])
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    K2 = K @ K
    if th < 0.00001:
        Rm = np.eye(3) + K + 0.5 * K2
        V = np.eye(3) + 0.5 * K + K2 / 6.0
    else:
        Rm = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / (th * th) * K2
        V = np.eye(3) + (1 - np.cos(th)) / (th * th) * K + (th - np.sin(th)) / (th ** 3) * K2
    return Rm, V @ u


def robust_chi2(chi2, huber_delta, active=None):
    """SparseOptimizer::activeRobustChi2 with g2o's RobustKernelHuber: rho(e2) = e2 inside delta^2, 2 delta sqrt(e2) -
    delta^2 outside; plain chi2 without a kernel (Optimize.cpp:904 removes it for the second round)."""
    r = np.array(chi2, np.float64)
    if huber_delta > 0:
        out = r > huber_delta * huber_delta
        r[out] = 2 * huber_delta * np.sqrt(r[out]) - huber_delta * huber_delta
    if active is not None:
        r = r[np.asarray(active, bool)]
    return float(r.sum())


def lm_optimize(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2, huber_delta,
                iterations, edge_active=None, tau=1e-5, max_trials=10, lower=1.0 / 3.0, upper=2.0 / 3.0):
    """g2o's OptimizationAlgorithmLevenberg + BlockSolver_6_3 with marginalised points, restated densely
    (g2o 20201223: optimization_algorithm_levenberg.cpp solve(), block_solver.hpp buildSystem/solve, sparse_optimizer.cpp
    optimize()) for the graph Optimize::localBundleAdjustment builds (reference modules/Backend/Optimize.cpp:811-911).
    Inactive edges (setLevel(1), Optimize.cpp:900-902) take no part.  Returns dict(pose_R, pose_t, points, chi2,
    iterations, lam, chi2_initial, chi2_final, trials)."""
    R = np.array(pose_R, np.float64).reshape(-1, 3, 3)
    t = np.array(pose_t, np.float64).reshape(-1, 3)
    P = np.array(points, np.float64).reshape(-1, 3)
    fixed = np.asarray(pose_fixed, bool)
    ep, el = np.asarray(edge_pose), np.asarray(edge_point)
    act = np.ones(len(ep), bool) if edge_active is None else np.asarray(edge_active, bool)
    free = np.flatnonzero(~fixed)
    slot = -np.ones(len(fixed), int)
    slot[free] = np.arange(len(free))
    nf, nl = len(free), len(P)

    def system(Rc, tc, Pc):
        lin = linearize(cam, Rc, tc, fixed, Pc, ep[act], el[act], np.asarray(edge_z).reshape(-1, 2)[act],
                        np.asarray(edge_inv_sigma2)[act], huber_delta)
        Hpl = np.zeros((nf, 6, nl, 3))
        es = slot[ep[act]]
        ok = es >= 0
        np.add.at(Hpl, (es[ok], slice(None), el[act][ok], slice(None)), np.transpose(lin["H_lp"][ok], (0, 2, 1)))
        return lin, Hpl

    def chi(Rc, tc, Pc):
        e, _ = residual(cam, Rc[ep], tc[ep], Pc[el], np.asarray(edge_z, np.float64).reshape(-1, 2))
        return np.asarray(edge_inv_sigma2, np.float64) * (e * e).sum(1)

    lam, ni, its, trials_total = 0.0, 2.0, 0, 0
    chi2_initial = robust_chi2(chi(R, t, P), huber_delta, act)
    for it in range(iterations):
        current = robust_chi2(chi(R, t, P), huber_delta, act)
        lin, Hpl = system(R, t, P)
        Hpp, bp, Hll, bl = lin["H_pp"][free], lin["b_p"][free], lin["H_ll"], lin["b_l"]
        if it == 0:  # computeLambdaInit: tau * max |diagonal| over the free vertices
            diag = np.concatenate([np.abs(np.einsum("nii->ni", Hpp)).ravel(), np.abs(np.einsum("nii->ni", Hll)).ravel()])
            lam, ni = tau * float(diag.max()), 2.0
        rho, qmax = 0.0, 0
        while True:
            Rb, tb, Pb = R.copy(), t.copy(), P.copy()  # push()
            inv = np.linalg.inv(Hll + lam * np.eye(3))
            T = np.einsum("panb,nbc->panc", Hpl, inv)  # Hpl * Hll^-1, block by block
            S = -(T.reshape(nf * 6, nl * 3) @ Hpl.reshape(nf * 6, nl * 3).T)
            for i in range(nf):
                S[6 * i:6 * i + 6, 6 * i:6 * i + 6] += Hpp[i] + lam * np.eye(6)
            rhs = bp.reshape(-1) - T.reshape(nf * 6, nl * 3) @ bl.reshape(-1)
            ok2 = True
            try:
                Lc = np.linalg.cholesky(S)
                xp = np.linalg.solve(Lc.T, np.linalg.solve(Lc, rhs))
            except np.linalg.LinAlgError:
                ok2, xp = False, np.zeros(nf * 6)
            xl = np.einsum("nab,nb->na", inv, bl - (Hpl.reshape(nf * 6, nl, 3) * xp[:, None, None]).sum(0))
            for i, ip in enumerate(free):  # VertexSE3::oplusImpl: exp(update) * estimate
                dR, dt = se3_exp(xp[6 * i:6 * i + 6])
                R[ip], t[ip] = dR @ R[ip], dR @ t[ip] + dt
            P = P + xl  # Vertex3D::oplusImpl
            temp = robust_chi2(chi(R, t, P), huber_delta, act) if ok2 else np.finfo(np.float64).max
            scale = float((xp * (lam * xp + bp.reshape(-1))).sum() + (xl * (lam * xl + bl)).sum()) + 1e-3
            rho = (current - temp) / scale
            trials_total += 1
            if rho > 0 and np.isfinite(temp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, upper)
                lam *= max(lower, alpha)
                ni = 2.0
                current = temp
            else:
                lam *= ni
                ni *= 2
                R, t, P = Rb, tb, Pb  # pop()
                if not np.isfinite(lam):
                    break
            qmax += 1
            if not (rho < 0 and qmax < max_trials):
                break
        its += 1
        if qmax == max_trials or rho == 0 or not np.isfinite(lam):
            break  # Terminate: optimize() leaves its loop
    final = chi(R, t, P)
    return {"pose_R": R, "pose_t": t, "points": P, "chi2": final, "iterations": its, "lam": lam,
            "chi2_initial": chi2_initial, "chi2_final": robust_chi2(final, huber_delta, act), "trials": trials_total}


def local_bundle_adjustment(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2,
                            huber_delta):
    """The optimisation part of Optimize::localBundleAdjustment (Optimize.cpp:892-922): optimize(5) with the Huber
    kernel; edges with chi2 > 5.991 leave (setLevel(1)) and the kernel is removed; optimize(10); final outlier flags."""
    a = lm_optimize(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2, huber_delta, 5)
    active = a["chi2"] <= 5.991
    b = lm_optimize(cam, a["pose_R"], a["pose_t"], pose_fixed, a["points"], edge_pose, edge_point, edge_z, edge_inv_sigma2,
                    0.0, 10, edge_active=active)
    # A level-1 edge is outside optimize(10)'s active set, so g2o does not recompute its error: e->chi2() at :919 still
    # returns the first round's value (> 5.991) and the observation is erased even if its residual at the final estimate
    # would pass (contrast Optimize.cpp:510, where poseOptimize calls computeError() explicitly before chi2()).
    b["chi2_final_estimate"] = b["chi2"].copy()
    b["chi2"] = np.where(active, b["chi2"], a["chi2"])
    b["outlier"] = ~active | (b["chi2"] > 5.991)
    b["first_round"] = a
    return b


def pose_optimize(cam, R0, t0, Pw, z, inv_sigma2, huber_delta, rounds=4, iterations=10, tau=1e-5, max_trials=10,
                  lower=1.0 / 3.0, upper=2.0 / 3.0):
    """Optimize::poseOptimize (reference modules/Backend/Optimize.cpp:444-545) for one frame: a single VertexSE3 and one
    EdgeSE3Project3DOnlyPose (G2oTypes.h:209-236, G2oTypes.cpp:27-34) per matched map point, Huber kernel on every edge;
    four rounds of optimize(10), each restarted from the initial pose (:497) on the edges classified as inliers by the
    previous round (chi2 <= 5.991, :506-516; the `iter == 2` test at :518 never fires, so the kernel stays).  The LM loop
    is g2o 20201223's OptimizationAlgorithmLevenberg on the 6x6 system.  Returns dict(R, t, inlier, n_inliers, chi2)."""
    R0 = np.array(R0, np.float64).reshape(3, 3)
    t0 = np.array(t0, np.float64).reshape(3)
    Pw = np.asarray(Pw, np.float64).reshape(-1, 3)
    z = np.asarray(z, np.float64).reshape(-1, 2)
    om = np.asarray(inv_sigma2, np.float64)
    n = len(Pw)
    inlier = np.ones(n, bool)
    def err(R, t):
        Pc = Pw @ R.T + t
        e = z - project(cam, Pc)
        return e, Pc

    def chi_of(R, t):
        e, _ = err(R, t)
        return om * (e * e).sum(1)

    if n < 3:  # :491
        return {"R": R0, "t": t0, "inlier": inlier, "n_inliers": 0, "chi2": chi_of(R0, t0)}
    R, t = R0, t0
    for _ in range(rounds):
        R, t = R0.copy(), t0.copy()  # vPose->setEstimate(Tcw) (:497)
        act = inlier.copy()          # initializeOptimization(0): level-0 edges
        lam, ni = 0.0, 2.0
        if act.any():
            for it in range(iterations):
                e, Pc = err(R, t)
                chi = om * (e * e).sum(1)
                current = robust_chi2(chi, huber_delta, act)
                Jp = proj_jacobian(cam, Pc)
                Jq = np.concatenate([Jp @ hat(Pc), -Jp], axis=2)
                rw = np.ones(n)
                out = chi > huber_delta * huber_delta
                rw[out] = huber_delta / np.sqrt(chi[out])
                W = np.where(act, rw * om, 0.0)
                H = np.einsum("e,eki,ekj->ij", W, Jq, Jq)
                b = -np.einsum("e,eki,ek->i", W, Jq, e)
                if it == 0:
                    lam, ni = tau * float(np.abs(np.diag(H)).max()), 2.0
                rho, qmax = 0.0, 0
                while True:
                    Rb, tb = R, t
                    try:
                        Lc = np.linalg.cholesky(H + lam * np.eye(6))
                        x = np.linalg.solve(Lc.T, np.linalg.solve(Lc, b))
                        ok2 = True
                    except np.linalg.LinAlgError:
                        x, ok2 = np.zeros(6), False
                    dR, dt = se3_exp(x)
                    R, t = dR @ R, dR @ t + dt
                    temp = robust_chi2(chi_of(R, t), huber_delta, act) if ok2 else np.finfo(np.float64).max
                    scale = float((x * (lam * x + b)).sum()) + 1e-3
                    rho = (current - temp) / scale
                    if rho > 0 and np.isfinite(temp):
                        lam *= max(lower, min(1.0 - (2 * rho - 1) ** 3, upper))
                        ni = 2.0
                        current = temp
                    else:
                        lam *= ni
                        ni *= 2
                        R, t = Rb, tb
                        if not np.isfinite(lam):
                            break
                    qmax += 1
                    if not (rho < 0 and qmax < max_trials):
                        break
                if qmax == max_trials or rho == 0 or not np.isfinite(lam):
                    break
        inlier = ~(chi_of(R, t) > 5.991)  # :503-516 (outliers get computeError() first, so every edge is re-evaluated)
    return {"R": R, "t": t, "inlier": inlier, "n_inliers": int(inlier.sum()), "chi2": chi_of(R, t)}
