"""Frame post-processing (Frame.cpp:24-51): oracle KATs on the CPU, HIP-vs-oracle parity on the GPU."""
import numpy as np
import pytest

from monoorbslam3_amd import synth

KITTI = dict(width=1242, height=375, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)
EUROC = dict(width=752, height=480, fx=458.654, fy=457.296, cx=367.215, cy=248.375)
EUROC_DIST = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)


def _kps(oracle_mod, n, w, h, seed):
    rng = np.random.RandomState(seed)
    k = np.zeros(n, dtype=oracle_mod.KP_DTYPE)
    k["octave"] = rng.randint(0, 8, n)
    s = (1.2 ** k["octave"]).astype(np.float32)
    # level coordinates times the level scale, as operator() emits them (ORBExtractor.cpp:537-542)
    k["x"] = (rng.randint(19, w, n) / s).astype(np.int32).astype(np.float32) * s
    k["y"] = (rng.randint(19, h, n) / s).astype(np.int32).astype(np.float32) * s
    k["size"] = 31 * s
    k["angle"] = rng.uniform(0, 360, n)
    k["response"] = rng.randint(7, 200, n)
    k["class_id"] = -1
    return k


def _distort(cam, dist, x, y):
    """forward RAD_TAN model in float64 (what undistortion must invert)"""
    k = list(dist) + [0.0] * (12 - len(dist))
    xn, yn = (x - cam["cx"]) / cam["fx"], (y - cam["cy"]) / cam["fy"]
    r2 = xn * xn + yn * yn
    cd = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2)
    xd = xn * cd + 2 * k[2] * xn * yn + k[3] * (r2 + 2 * xn * xn)
    yd = yn * cd + k[2] * (r2 + 2 * yn * yn) + 2 * k[3] * xn * yn
    return xd * cam["fx"] + cam["cx"], yd * cam["fy"] + cam["cy"]


def test_oracle_undistort_is_copy_without_k1(oracle_mod):
    k = _kps(oracle_mod, 500, 1242, 375, 1)
    for dist in ((), (0.0, 0.1, 0.01, 0.01)):  # Pinhole.cpp:62 only looks at dist[0]
        raw, un, start, items = oracle_mod.frame_post(**KITTI, dist=dist, kps=k)
        assert raw.tobytes() == k.tobytes() and un.tobytes() == k.tobytes()
        assert start[-1] == len(k) and sorted(items) == list(range(len(k)))


def test_oracle_undistort_inverts_the_forward_model(oracle_mod):
    rng = np.random.RandomState(2)
    xu, yu = rng.uniform(60, 690, 400), rng.uniform(40, 440, 400)
    xd, yd = _distort(EUROC, EUROC_DIST, xu, yu)
    k = _kps(oracle_mod, 400, 752, 480, 3)
    k["x"], k["y"] = xd, yd
    raw, un, _, _ = oracle_mod.frame_post(**EUROC, dist=EUROC_DIST, kps=k)
    # only five fixed-point iterations (cv::undistortPoints' default): tight near the centre, ~0.07 px in the
    # corners of this strongly distorted camera (the forward shift there is 47 px)
    err = np.hypot(un["x"] - xu, un["y"] - yu)
    assert err.max() < 0.1
    centre = np.hypot(xu - EUROC["cx"], yu - EUROC["cy"]) < 150
    assert err[centre].max() < 2e-3
    assert np.array_equal(raw["x"], k["x"]) and np.array_equal(un["angle"], k["angle"])


def test_oracle_grid_is_cell_major_and_ascending(oracle_mod):
    k = _kps(oracle_mod, 2000, 752, 480, 4)
    _, un, start, items = oracle_mod.frame_post(**EUROC, dist=EUROC_DIST, kps=k)
    rows = 480 // 40
    assert len(start) == (752 // 40 + 1) * rows + 1  # Frame.cpp:32-40: 752 is not a multiple of 40
    for c in range(len(start) - 1):
        cell = items[start[c]:start[c + 1]]
        assert np.all(np.diff(cell) > 0)
        assert np.all((np.floor(un["x"][cell]).astype(int) // 40) * rows + np.floor(un["y"][cell]).astype(int) // 40 == c)
    inside = (np.floor(un["x"]) >= 0) & (np.floor(un["x"]) < 752) & (np.floor(un["y"]) >= 0) & (np.floor(un["y"]) < 480)
    assert start[-1] == inside.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["kitti_nodist", "euroc", "euroc_k3", "fisheye"])
def test_frame_post_parity(oracle_mod, case):
    from monoorbslam3_amd.frame import FramePost
    if case == "kitti_nodist":
        cam, dist, kw = KITTI, (), {}
    elif case == "euroc":
        cam, dist, kw = EUROC, EUROC_DIST, {}
    elif case == "euroc_k3":
        cam, dist, kw = EUROC, EUROC_DIST + (0.01, 0.002, -0.001, 0.0005, 1e-4, -2e-4, 3e-4, 1e-5), {}
    else:
        rng = np.random.RandomState(8)
        cam, dist = EUROC, EUROC_DIST
        kw = dict(undistort=False, size_scale=rng.uniform(0.8, 2.5, (480, 752)).astype(np.float32))
    k = _kps(oracle_mod, 2500, cam["width"], cam["height"], 11)
    k["x"][:40] = np.linspace(-3, cam["width"] + 3, 40)  # some key points leave the image after undistortion
    want = oracle_mod.frame_post(**cam, dist=dist, kps=k, **kw)
    fp = FramePost(**cam, dist=dist, **kw)
    got = fp(k)
    for g, w, name in zip(got, want, ("raw", "undistorted", "cell_start", "cell_items")):
        assert g.tobytes() == w.tobytes(), name
    assert len(fp(k[:0])[3]) == 0  # empty frame


@pytest.mark.gpu
def test_frame_post_batch_on_extractor_output(oracle_mod):
    """extract_batch_device -> frame_post_device without leaving the GPU; every frame equals the oracle's record."""
    import torch
    from monoorbslam3_amd.extractor import ORBExtractor
    from monoorbslam3_amd.frame import FramePost
    B, W, H = 6, 752, 480
    frames = synth.make_frames(B, W, H, seed=77)
    ex = ORBExtractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    cap = ex.max_keypoints(W, H)
    d_img = torch.from_numpy(frames).cuda()
    kp = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(B, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    ex.extract_batch_device(d_img.data_ptr(), B, W, H, W, W * H, kp.data_ptr(), desc.data_ptr(), cap, n.data_ptr(),
                            s.cuda_stream)
    fp = FramePost(**EUROC, dist=EUROC_DIST)
    un = torch.zeros_like(kp)
    start = torch.zeros((B, fp.n_cells + 1), dtype=torch.int32, device="cuda")
    items = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    fp.post_device(B, kp.data_ptr(), n.data_ptr(), cap, un.data_ptr(), start.data_ptr(), items.data_ptr(), s.cuda_stream)
    s.synchronize()
    n_h = n.cpu().numpy()
    for f in range(B):
        raw_h = kp[f].cpu().numpy().view(oracle_mod.KP_DTYPE).reshape(-1)[: n_h[f]]
        w_raw, w_un, w_start, w_items = oracle_mod.frame_post(**EUROC, dist=EUROC_DIST, kps=raw_h)
        assert un[f].cpu().numpy().view(oracle_mod.KP_DTYPE).reshape(-1)[: n_h[f]].tobytes() == w_un.tobytes()
        assert np.array_equal(start[f].cpu().numpy(), w_start)
        assert np.array_equal(items[f].cpu().numpy()[: w_start[-1]], w_items)
