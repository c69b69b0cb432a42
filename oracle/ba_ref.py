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


def residual(cam, R, t, P, z):
    """e = z - project(R P + t) for matched rows."""
    fx, fy, cx, cy = cam
    Pc = np.einsum("eij,ej->ei", R, P) + t
    u = fx * (Pc[:, 0] / Pc[:, 2]) + cx
    v = fy * (Pc[:, 1] / Pc[:, 2]) + cy
    return z - np.stack([u, v], 1), Pc


def linearize(cam, pose_R, pose_t, pose_fixed, points, edge_pose, edge_point, edge_z, edge_inv_sigma2, huber_delta):
    fx, fy, cx, cy = cam
    R = np.asarray(pose_R, np.float64).reshape(-1, 3, 3)[edge_pose]
    t = np.asarray(pose_t, np.float64).reshape(-1, 3)[edge_pose]
    P = np.asarray(points, np.float64).reshape(-1, 3)[edge_point]
    z = np.asarray(edge_z, np.float64).reshape(-1, 2)
    om = np.asarray(edge_inv_sigma2, np.float64)
    e, Pc = residual(cam, R, t, P, z)
    X, Y, Z = Pc[:, 0], Pc[:, 1], Pc[:, 2]
    zero = np.zeros_like(X)
    Jp = np.stack([np.stack([fx / Z, zero, -fx * X / (Z * Z)], 1), np.stack([zero, fy / Z, -fy * Y / (Z * Z)], 1)], 1)
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
    """g2o SE3Quat::exp for update = (omega, upsilon) (VertexSE3::oplusImpl, G2oTypes.h:112-115)."""
    w, u = upd[:3], upd[3:]
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        Rm = np.eye(3) + K
        V = np.eye(3) + 0.5 * K
    else:
        Rm = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
    return Rm, V @ u
