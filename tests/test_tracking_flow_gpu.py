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


def _chain_stream(torch, dev, kind):
    """"explicit": a torch stream of the test's own (non-blocking), made current.  "null": None -- the wrappers pass NULL."""
    if kind == "null":
        return None
    chain = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(chain)
    assert chain.cuda_stream != 0
    return chain.cuda_stream


@pytest.mark.parametrize("stream_kind", ["explicit", "null"])
def test_tracking_chain_stays_on_the_device(stream_kind):
    """Tracking.cpp:289-336 with no host hop: orbx_extract_batch_device -> orbf_frame_post_device ->
    orbm_search_by_projection_points_device -> orbba_pose_edges_device -> orbba_pose_optimize_batch_device, one stream,
    nothing copied or waited for in between; the pose, the inlier flags and the matches must equal what the same steps
    give through the host entry points (same kernels underneath, same edge order)."""
    import torch
    from monoorbslam3_amd import ba
    from monoorbslam3_amd.extractor import ORBExtractor, KP_DTYPE
    from monoorbslam3_amd.frame import FramePost
    from monoorbslam3_amd.matcher import ORBMatcher
    dev = torch.device("cuda", 0)
    w, h, Z = 752, 480, 10.0
    fx = fy = 460.0
    cx, cy = 376.0, 240.0
    cam = (fx, fy, cx, cy)
    canvas = synth.make_canvas(w + 80, h + 60, seed=2024)
    dx, dy = 9, 6
    f1 = np.ascontiguousarray(canvas[30:30 + h, 40:40 + w])
    f2 = np.ascontiguousarray(canvas[30 + dy:30 + dy + h, 40 + dx:40 + dx + w])
    ex = ORBExtractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h)
    post = FramePost(w, h, fx, fy, cx, cy)
    # view 1 (the "local map") through the host path
    k1, d1 = ex(f1)
    _, k1u, _, _ = post(k1)
    nq = len(k1u)
    Pw = np.stack([(k1u["x"] - cx) * Z / fx, (k1u["y"] - cy) * Z / fy, np.full(nq, Z)], 1).astype(np.float32)
    q_xy = np.stack([k1u["x"], k1u["y"]], 1).astype(np.float32)
    q_level = k1u["octave"].astype(np.int32)
    q_radius = (16.0 * 1.2 ** q_level).astype(np.float32)
    q_ok = np.ones(nq, np.uint8)
    # ---- view 2: the device chain
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    capk = ex.max_keypoints(w, h)
    img = t(f2[None])
    d_kp = torch.zeros((1, capk, 28), dtype=torch.uint8, device=dev)
    d_un = torch.zeros((1, capk, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((1, capk, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((1,), dtype=torch.int32, device=dev)
    d_start = torch.zeros((1, post.n_cells + 1), dtype=torch.int32, device=dev)
    d_items = torch.zeros((1, capk), dtype=torch.int32, device=dev)
    # ONE stream for the whole chain: a non-blocking stream of the caller's, or NULL -- which every entry point, with or
    # without a handle, runs on stream 0 (include/orbx.h, "Streams"), torch's default stream: the fills above, the four
    # handles' kernels and the reads below are then in order with no wait in between
    st = _chain_stream(torch, dev, stream_kind)
    ex.extract_batch_device(img.data_ptr(), 1, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), capk, d_n.data_ptr(), st)
    post.post_device(1, d_kp.data_ptr(), d_n.data_ptr(), capk, d_un.data_ptr(), d_start.data_ptr(), d_items.data_ptr(), st)
    # the key-point count is needed as a host integer by the next two calls' signatures: capk (slots past the count hold
    # no grid items, so the search never returns them) keeps the chain free of a read-back
    d = dict(q_desc=t(d1), q_xy=t(q_xy), q_radius=t(q_radius), q_level=t(q_level), q_ok=t(q_ok), kps2=d_un, desc2=d_desc,
             cell_start=d_start, cell_items=d_items, frame_mp=torch.full((capk,), -1, dtype=torch.int32, device=dev),
             result=torch.zeros(8, dtype=torch.int32, device=dev))
    m = ORBMatcher(0.8, True)
    m.SearchByProjectionDevice("points", d, nq, capk, post.cols, post.rows, list_cap=64, stream=st)
    e_off = torch.zeros(2, dtype=torch.int32, device=dev)
    e_P = torch.zeros((capk, 3), dtype=torch.float64, device=dev)
    e_z = torch.zeros((capk, 2), dtype=torch.float64, device=dev)
    e_w = torch.zeros(capk, dtype=torch.float64, device=dev)
    e_kp = torch.zeros(capk, dtype=torch.int32, device=dev)
    d_Pw = t(Pw)
    ba.pose_edges_device(capk, nq, d["frame_mp"], d_un, d_Pw, e_off, e_P, e_z, e_w, e_kp, stream=st)
    R0, t0 = t(np.eye(3)[None]), t(np.zeros((1, 3)))
    R_out = torch.zeros((1, 3, 3), dtype=torch.float64, device=dev)
    t_out = torch.zeros((1, 3), dtype=torch.float64, device=dev)
    inl = torch.zeros(capk, dtype=torch.uint8, device=dev)
    n_inl = torch.zeros(1, dtype=torch.int32, device=dev)
    chi2 = torch.zeros(capk, dtype=torch.float64, device=dev)
    ba.pose_optimize_batch_device(cam, R0, t0, e_off, e_P, e_z, e_w, R_out, t_out, inl, n_inl, chi2, stream=st)
    torch.cuda.synchronize()   # the first wait of the chain
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
    # ---- the same through the host entry points
    n2 = int(d_n[0])
    k2u = np.frombuffer(d_un[0, :n2].cpu().numpy().tobytes(), KP_DTYPE)
    d2 = d_desc[0, :n2].cpu().numpy()
    n_match, mp, _ = m.SearchByProjectionPoints(d1, q_xy, q_radius, q_level, q_ok, k2u, d2, w, h, np.full(n2, -1, np.int32))
    got_mp = d["frame_mp"].cpu().numpy()
    print("device chain: result", d["result"].cpu().numpy().tolist(), "host matches", n_match, "key points", n2, "queries", nq)
    assert int(d["result"][1]) == 0 and int(d["result"][0]) == n_match and n_match > 300
    assert np.array_equal(got_mp[:n2], mp) and np.all(got_mp[n2:] == -1)
    idx2 = np.flatnonzero(mp >= 0)
    idx1 = mp[idx2]
    ne = int(e_off[1])
    assert ne == len(idx2) and np.array_equal(e_kp[:ne].cpu().numpy(), idx2)
    zz = np.stack([k2u["x"][idx2], k2u["y"][idx2]], 1).astype(np.float64)
    ww = (np.float32(1.0) / k2u["size"][idx2] / k2u["size"][idx2]).astype(np.float64)
    assert np.array_equal(e_z[:ne].cpu().numpy(), zz) and np.array_equal(e_w[:ne].cpu().numpy(), ww)
    ref = ba.pose_optimize_batch(cam, np.eye(3)[None], np.zeros((1, 3)), np.array([0, ne], np.int32), Pw[idx1].astype(np.float64), zz, ww)
    assert np.array_equal(R_out.cpu().numpy(), ref["pose_R"]) and np.array_equal(t_out.cpu().numpy(), ref["pose_t"])
    assert int(n_inl[0]) == ref["n_inliers"][0] and np.array_equal(inl[:ne].cpu().numpy().astype(bool), ref["inlier"])
    t_true = np.array([-dx * Z / fx, -dy * Z / fy, 0.0])
    assert np.abs(t_out.cpu().numpy()[0] - t_true).max() < 0.02


@pytest.mark.parametrize("stream_kind", ["explicit", "null"])
def test_bow_branch_stays_on_the_device(stream_kind):
    """Tracking.cpp:255-273 (the branch without a motion model) with no host hop: orbx_extract_batch_device of the key frame
    and the frame in one batch -> orbv_transform_device (computeBow, levelsup 4) -> orbm_search_by_bow_device, one stream, the
    first wait at the very end.  frame_mp and the match count equal the host entry point's on the records and FeatureVectors
    read back afterwards."""
    import torch
    from monoorbslam3_amd.extractor import ORBExtractor, KP_DTYPE
    from monoorbslam3_amd.matcher import ORBMatcher
    from monoorbslam3_amd.vocabulary import ORBVocabulary
    dev = torch.device("cuda", 0)
    w, h = 752, 480
    canvas = synth.make_canvas(w + 80, h + 60, seed=77)
    f = np.stack([canvas[30:30 + h, 40:40 + w], canvas[34:34 + h, 47:47 + w]])  # the frame sees the key frame's scene shifted by (7, 4)
    ex = ORBExtractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    voc = ORBVocabulary.from_arrays(synth.make_vocabulary(10, 5, seed=3), device=0)
    cap = ex.max_keypoints(w, h)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
    img = torch.from_numpy(np.ascontiguousarray(f)).to(dev)
    d_kp, d_desc, d_n = z((2, cap, 28), torch.uint8), z((2, cap, 32), torch.uint8), z((2,), torch.int32)
    bow_ids, bow_vals, n_words = z((2, cap), torch.int32), z((2, cap), torch.float64), z((2,), torch.int32)
    fv_nodes, fv_off, fv_idx, n_fv = z((2, cap), torch.int32), z((2, cap + 1), torch.int32), z((2, cap), torch.int32), z((2,), torch.int32)
    kf_ok = torch.ones(cap, dtype=torch.uint8, device=dev)          # every key-frame feature has a live map point
    frame_mp = torch.full((cap,), -1, dtype=torch.int32, device=dev)
    result = z((8,), torch.int32)
    st = _chain_stream(torch, dev, stream_kind)
    ex.extract_batch_device(img.data_ptr(), 2, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(), st)
    voc.transform_device(2, d_desc.data_ptr(), d_n.data_ptr(), cap, 4, bow_ids.data_ptr(), bow_vals.data_ptr(), n_words.data_ptr(),
                         fv_nodes.data_ptr(), fv_off.data_ptr(), fv_idx.data_ptr(), n_fv.data_ptr(), st)
    m = ORBMatcher(0.7, True)
    # the counts are only needed as upper bounds by the call's signature: the capacity serves (slots past a frame's count
    # appear in no FeatureVector node, so they are neither queries nor candidates)
    d = dict(desc1=d_desc[0], kps1=d_kp[0], kf_mp_ok=kf_ok, fv1=(fv_nodes[0], fv_off[0], fv_idx[0], n_fv[0:1]), desc2=d_desc[1], kps2=d_kp[1],
             frame_mp=frame_mp, fv2=(fv_nodes[1], fv_off[1], fv_idx[1], n_fv[1:2]), result=result)
    m.SearchByBowDevice(d, cap, cap, stream=st)
    torch.cuda.synchronize()   # the first wait of the chain
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
    # ---- the same through the host entry point
    n = d_n.cpu().numpy()
    kps = [np.frombuffer(d_kp[i, :n[i]].cpu().numpy().tobytes(), KP_DTYPE) for i in range(2)]
    desc = [d_desc[i, :n[i]].cpu().numpy() for i in range(2)]
    nf = n_fv.cpu().numpy()
    fvs = [(fv_nodes[i, :nf[i]].cpu().numpy().view(np.uint32), fv_off[i, :nf[i] + 1].cpu().numpy(),
            fv_idx[i, :fv_off[i, nf[i]]].cpu().numpy().view(np.uint32)) for i in range(2)]
    n_host, mp_host = m.SearchByBow(desc[0], kps[0]["angle"], np.ones(n[0], np.uint8), fvs[0], desc[1], kps[1]["angle"],
                                    np.full(n[1], -1, np.int32), fvs[1])
    res = result.cpu().numpy()
    got = frame_mp.cpu().numpy()
    print("BoW branch on the device: result", res.tolist(), "host matches", n_host, "key points", n.tolist(), "nodes", nf.tolist())
    assert res[1] == 0 and res[0] == n_host and n_host > 100
    assert np.array_equal(got[:n[1]], mp_host) and np.all(got[n[1]:] == -1)


def test_two_searches_share_one_index_space_before_the_pose_edges():
    """include/orbba.h (orbba_pose_edges_device): Tracking.cpp:289-336 runs frame -> frame SearchByProjection and then map points ->
    frame on the SAME frame_mp.  On the device both searches get one shared index space -- concatenated query arrays, q_ok masks
    selecting each search's part -- so that frame_mp holds indices into ONE d_q_points array when the edges are built.
    Device chain: orbx_extract_batch_device -> orbf_frame_post_device -> orbm_search_by_projection_frame_device (first half of the
    queries) -> orbm_search_by_projection_points_device (second half) -> orbba_pose_edges_device, NULL stream throughout; against
    the same two searches through the host entry points and the edges listed by hand from their frame_mp."""
    import torch
    from monoorbslam3_amd import ba
    from monoorbslam3_amd.extractor import ORBExtractor, KP_DTYPE
    from monoorbslam3_amd.frame import FramePost
    from monoorbslam3_amd.matcher import ORBMatcher
    dev = torch.device("cuda", 0)
    w, h, Z = 752, 480, 10.0
    fx = fy = 460.0
    cx, cy = 376.0, 240.0
    canvas = synth.make_canvas(w + 80, h + 60, seed=515)
    dx, dy = 7, 5
    f1 = np.ascontiguousarray(canvas[30:30 + h, 40:40 + w])
    f2 = np.ascontiguousarray(canvas[30 + dy:30 + dy + h, 40 + dx:40 + dx + w])
    ex = ORBExtractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h)
    post = FramePost(w, h, fx, fy, cx, cy)
    k1, d1 = ex(f1)
    _, k1u, _, _ = post(k1)
    nq = len(k1u)
    Pw = np.stack([(k1u["x"] - cx) * Z / fx, (k1u["y"] - cy) * Z / fy, np.full(nq, Z)], 1).astype(np.float32)
    q_xy = np.stack([k1u["x"] - dx, k1u["y"] - dy], 1).astype(np.float32)   # where view 2 sees them
    q_level = k1u["octave"].astype(np.int32)
    q_angle = k1u["angle"].astype(np.float32)
    q_radius = (7.0 * 1.2 ** q_level).astype(np.float32)
    # one index space, two searches: even queries are "last frame" features (frame -> frame), odd ones local map points
    ok_frame = (np.arange(nq) % 2 == 0).astype(np.uint8)
    ok_points = (np.arange(nq) % 2 == 1).astype(np.uint8)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    capk = ex.max_keypoints(w, h)
    img = t(f2[None])
    d_kp = torch.zeros((1, capk, 28), dtype=torch.uint8, device=dev)
    d_un = torch.zeros((1, capk, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((1, capk, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((1,), dtype=torch.int32, device=dev)
    d_start = torch.zeros((1, post.n_cells + 1), dtype=torch.int32, device=dev)
    d_items = torch.zeros((1, capk), dtype=torch.int32, device=dev)
    ex.extract_batch_device(img.data_ptr(), 1, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), capk, d_n.data_ptr())
    post.post_device(1, d_kp.data_ptr(), d_n.data_ptr(), capk, d_un.data_ptr(), d_start.data_ptr(), d_items.data_ptr())
    frame_mp = torch.full((capk,), -1, dtype=torch.int32, device=dev)
    common = dict(q_desc=t(d1), q_xy=t(q_xy), q_radius=t(q_radius), q_level=t(q_level), q_angle=t(q_angle), kps2=d_un, desc2=d_desc,
                  cell_start=d_start, cell_items=d_items, frame_mp=frame_mp)
    m = ORBMatcher(0.8, True)
    res_f, res_p = torch.zeros(8, dtype=torch.int32, device=dev), torch.zeros(8, dtype=torch.int32, device=dev)
    m.SearchByProjectionDevice("frame", dict(common, q_ok=t(ok_frame), result=res_f), nq, capk, post.cols, post.rows, list_cap=64)
    m.SearchByProjectionDevice("points", dict(common, q_ok=t(ok_points), result=res_p), nq, capk, post.cols, post.rows, list_cap=64)
    e_off = torch.zeros(2, dtype=torch.int32, device=dev)
    e_P = torch.zeros((capk, 3), dtype=torch.float64, device=dev)
    e_z = torch.zeros((capk, 2), dtype=torch.float64, device=dev)
    e_w = torch.zeros(capk, dtype=torch.float64, device=dev)
    e_kp = torch.zeros(capk, dtype=torch.int32, device=dev)
    ba.pose_edges_device(capk, nq, frame_mp, d_un, t(Pw), e_off, e_P, e_z, e_w, e_kp)
    # ---- the host chain on the records read back
    n2 = int(d_n[0])
    k2u = np.frombuffer(d_un[0, :n2].cpu().numpy().tobytes(), KP_DTYPE)
    d2 = d_desc[0, :n2].cpu().numpy()
    n_f, mp = m.SearchByProjectionFrame(d1, q_xy, q_radius, q_level, q_angle, ok_frame, k2u, d2, w, h, np.full(n2, -1, np.int32))
    n_p, mp, _ = m.SearchByProjectionPoints(d1, q_xy, q_radius, q_level, ok_points, k2u, d2, w, h, mp)
    got = frame_mp.cpu().numpy()
    print("frame search %d + points search %d matches of %d queries, %d key points" % (n_f, n_p, nq, n2))
    assert int(res_f[1]) == 0 and int(res_p[1]) == 0 and int(res_f[0]) == n_f and int(res_p[0]) == n_p
    assert n_f > 100 and n_p > 100
    assert np.array_equal(got[:n2], mp) and np.all(got[n2:] == -1)
    idx2 = np.flatnonzero(mp >= 0)
    q = mp[idx2]
    assert (q % 2 == 0).any() and (q % 2 == 1).any()       # the edges mix both searches' queries
    ne = int(e_off[1])
    assert ne == len(idx2) and np.array_equal(e_kp[:ne].cpu().numpy(), idx2)
    assert np.array_equal(e_P[:ne].cpu().numpy(), Pw[q].astype(np.float64))
    zz = np.stack([k2u["x"][idx2], k2u["y"][idx2]], 1).astype(np.float64)
    ww = (np.float32(1.0) / k2u["size"][idx2] / k2u["size"][idx2]).astype(np.float64)
    assert np.array_equal(e_z[:ne].cpu().numpy(), zz) and np.array_equal(e_w[:ne].cpu().numpy(), ww)
