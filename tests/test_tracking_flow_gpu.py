"""The hot path and its neighbours chained the way the reference's Tracking thread uses them for one frame
(Tracking.cpp: Frame::Frame -> SearchByProjection(frame, local map points) -> Optimize::poseOptimize), checked against
ground truth that does not come from the oracle: a fronto-parallel textured plane at depth Z seen by a camera that
translates, so the true pose of the second view is known exactly from the crop offsets."""
import numpy as np
import pytest

from monoorbslam3_amd import synth

pytestmark = pytest.mark.gpu


def test_track_one_frame_against_ground_truth():
    from monoorbslam3_amd import ba
    from monoorbslam3_amd.extractor import ORBExtractor
    from monoorbslam3_amd.frame import FramePost
    from monoorbslam3_amd.matcher import ORBMatcher
    w, h, Z = 752, 480, 10.0
    fx = fy = 460.0
    cx, cy = 376.0, 240.0
    canvas = synth.make_canvas(w + 80, h + 60, seed=2024)
    dx, dy = 11, -7  # the second view's crop is shifted by (dx, dy) pixels: every pixel moves by (-dx, -dy)
    f1 = np.ascontiguousarray(canvas[30:30 + h, 40:40 + w])
    f2 = np.ascontiguousarray(canvas[30 + dy:30 + dy + h, 40 + dx:40 + dx + w])
    ex = ORBExtractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h)
    post = FramePost(w, h, fx, fy, cx, cy)  # no distortion: undistortKeyPoints copies (Pinhole.cpp:62)
    k1, d1 = ex(f1)
    k2, d2 = ex(f2)
    _, k1u, _, _ = post(k1)
    _, k2u, start2, items2 = post(k2)
    assert start2[-1] == len(k2u)
    # "local map": every key point of view 1 back-projected onto the plane (camera 1 = world)
    Pw = np.stack([(k1u["x"] - cx) * Z / fx, (k1u["y"] - cy) * Z / fy, np.full(len(k1u), Z)], 1).astype(np.float64)
    # view 2: P_c2 = P_w + t with t = (-dx, -dy) * Z / f
    t_true = np.array([-dx * Z / fx, -dy * Z / fy, 0.0])
    # Tracking predicts the pose with a motion model; here: identity, so the search window has to absorb the motion
    q_xy = np.stack([k1u["x"], k1u["y"]], 1).astype(np.float32)
    q_level = k1u["octave"].astype(np.int32)
    q_radius = (16.0 * 1.2 ** q_level).astype(np.float32)
    n_match, mp, counters = ORBMatcher(0.8, True).SearchByProjectionPoints(
        d1, q_xy, q_radius, q_level, np.ones(len(k1u), np.uint8), k2u, d2, w, h, np.full(len(k2u), -1, np.int32))
    assert n_match > 400 and counters[0] == 0
    idx2 = np.flatnonzero(mp >= 0)
    idx1 = mp[idx2]
    # most matches are the true correspondences (same canvas point): displaced by exactly (-dx, -dy)
    disp = np.stack([k2u["x"][idx2] - k1u["x"][idx1], k2u["y"][idx2] - k1u["y"][idx1]], 1)
    tol = 1.5 * 1.2 ** k2u["octave"][idx2]  # a level-l key point is localised to about one level-l pixel
    good = (np.abs(disp[:, 0] + dx) < tol) & (np.abs(disp[:, 1] + dy) < tol)
    print("matches %d, true correspondences %.1f %%" % (n_match, 100 * good.mean()))
    assert good.mean() > 0.7
    # Optimize::poseOptimize from the identity prediction
    z = np.stack([k2u["x"][idx2], k2u["y"][idx2]], 1).astype(np.float64)
    inv_sigma2 = 1.0 / k2u["size"][idx2].astype(np.float64) ** 2  # Optimize.cpp:478: kp.size is the level's scale factor
    out = ba.pose_optimize_batch((fx, fy, cx, cy), np.eye(3)[None], np.zeros((1, 3)), np.array([0, len(idx2)], np.int32),
                                 Pw[idx1], z, inv_sigma2)
    R, t = out["pose_R"][0], out["pose_t"][0]
    assert np.abs(R - np.eye(3)).max() < 2e-3
    assert np.abs(t - t_true).max() < 0.02  # 0.02 m at Z = 10 m and f = 460 px is about one pixel
    # the outliers poseOptimize drops are the wrong matches
    print("pose: t = %s (true %s), inliers %d of %d, of which true %.1f %%" % (t, t_true, out["n_inliers"][0], len(idx2),
                                                                               100 * good[out["inlier"]].mean()))
    assert out["n_inliers"][0] > 0.6 * len(idx2)
    assert good[out["inlier"]].mean() > 0.95
