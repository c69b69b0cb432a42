"""Optional second kernel set (local-BA linearisation).  CPU: the numpy oracle's analytic Jacobians against
numeric differences of the residual under the reference's update rules.  GPU: HIP kernels vs the oracle,
relative tolerance 1e-9 (f64; only the summation order differs)."""
import numpy as np
import pytest

from monoorbslam3_amd import synth

TOL = 1e-9


def _close(a, b, tol=TOL):
    scale = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() <= tol * scale


def test_jacobians_match_numeric_differences():
    """KAT (10) of SURVEY 8c: J_point = de/dP, J_pose = de/d(delta) with T <- exp(delta) * T (rotation first)"""
    from oracle import ba_ref
    pr = synth.make_ba_problem(4, 40, seed=3)
    lin = ba_ref.linearize(pr["cam"], pr["pose_R"], pr["pose_t"], pr["pose_fixed"], pr["points"], pr["edge_pose"],
                           pr["edge_point"], pr["edge_z"], pr["edge_inv_sigma2"], 0.0)
    R = pr["pose_R"].reshape(-1, 3, 3)
    h = 1e-6
    for e in range(0, len(pr["edge_pose"]), 7):
        ip, il = pr["edge_pose"][e], pr["edge_point"][e]
        z = pr["edge_z"][e:e + 1]

        def res(Rm, tv, P):
            return ba_ref.residual(pr["cam"], Rm[None], tv[None], P[None], z)[0][0]
        e0 = res(R[ip], pr["pose_t"][ip], pr["points"][il])
        assert np.allclose(e0, lin["error"][e])
        for k in range(3):
            d = np.zeros(3); d[k] = h
            num = (res(R[ip], pr["pose_t"][ip], pr["points"][il] + d) - res(R[ip], pr["pose_t"][ip], pr["points"][il] - d)) / (2 * h)
            assert np.allclose(num, lin["J_point"][e][:, k], rtol=1e-5, atol=1e-6)
        for k in range(6):
            d = np.zeros(6); d[k] = h
            Rp, tp = ba_ref.se3_exp(d)
            Rm, tm = ba_ref.se3_exp(-d)
            num = (res(Rp @ R[ip], Rp @ pr["pose_t"][ip] + tp, pr["points"][il]) -
                   res(Rm @ R[ip], Rm @ pr["pose_t"][ip] + tm, pr["points"][il])) / (2 * h)
            assert np.allclose(num, lin["J_pose"][e][:, k], rtol=1e-5, atol=1e-5)


def test_huber_and_fixed_blocks():
    from oracle import ba_ref
    from monoorbslam3_amd.ba import HUBER_MONO
    pr = synth.make_ba_problem(5, 60, seed=5, n_fixed=2)
    a = ba_ref.linearize(pr["cam"], pr["pose_R"], pr["pose_t"], pr["pose_fixed"], pr["points"], pr["edge_pose"],
                         pr["edge_point"], pr["edge_z"], pr["edge_inv_sigma2"], HUBER_MONO)
    assert abs(HUBER_MONO ** 2 - 5.991) < 1e-5
    assert (a["chi2"] > 5.991).any() and (a["chi2"] < 5.991).any()      # both Huber branches exercised
    assert np.all(a["H_pp"][:2] == 0) and np.all(a["b_p"][:2] == 0)      # fixed key frames carry no pose blocks
    assert np.abs(a["H_pp"][2:]).max() > 0 and np.abs(a["H_ll"]).max() > 0
    assert np.allclose(a["H_pp"], a["H_pp"].transpose(0, 2, 1)) and np.allclose(a["H_ll"], a["H_ll"].transpose(0, 2, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("n_poses,n_points", [(20, 3000), (3, 17)])
def test_gpu_matches_oracle(n_poses, n_points):
    from oracle import ba_ref
    from monoorbslam3_amd import ba
    pr = synth.make_ba_problem(n_poses, n_points, seed=11, n_fixed=min(4, n_poses - 1))
    ref = ba_ref.linearize(pr["cam"], pr["pose_R"], pr["pose_t"], pr["pose_fixed"], pr["points"], pr["edge_pose"],
                           pr["edge_point"], pr["edge_z"], pr["edge_inv_sigma2"], ba.HUBER_MONO)
    got = ba.linearize(pr["cam"], pr["pose_R"], pr["pose_t"], pr["pose_fixed"], pr["points"], pr["edge_pose"],
                       pr["edge_point"], pr["edge_z"], pr["edge_inv_sigma2"])
    for k in ("chi2", "error", "H_pp", "b_p", "H_ll", "b_l", "H_lp"):
        assert _close(got[k], ref[k]), k
    if n_points == 3000:
        assert len(pr["edge_pose"]) > 20000
        print("BA linearise: %d edges, %.3f ms on device" % (len(pr["edge_pose"]), got["kernel_ms"]))
