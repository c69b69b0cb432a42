"""GPU parity: every stage of the HIP extractor against the CPU oracle, through the C ABI.

Bar: bit-exact pyramid / blurred levels, identical FAST candidate sets, identical
keypoint (u, v, octave, response) lists in the reference's order, identical 32-byte
descriptors; orientation bit-identical (shared arithmetic), checked with tolerance
1e-4 degrees as the stated fallback bound.
"""
import numpy as np
import pytest

from monoorbslam3_amd import synth

pytestmark = pytest.mark.gpu


def _mk(oracle_mod, n_features, w, h, ini=20, mn=7, n_levels=8, sf=1.2, batch=1, variants=None):
    from monoorbslam3_amd.extractor import ORBExtractor
    ex = ORBExtractor(n_features, sf, n_levels, ini, mn, max_width=w, max_height=h, max_batch=batch, variants=variants)
    orc = oracle_mod.Oracle(n_features, sf, n_levels, ini, mn)
    return ex, orc


def _check_frame(ex, orc, img, kps, desc, frame=0, stages=True):
    h, w = img.shape
    okps, odesc, ocounts = orc.extract(img)
    if stages:
        pyr = orc.pyramid(img)
        for l in range(orc.n_levels):
            got = ex.tap_level(frame, l, w, h, blurred=False)
            assert np.array_equal(got, pyr[l]), "pyramid level %d differs" % l
            gotb = ex.tap_level(frame, l, w, h, blurred=True)
            assert np.array_equal(gotb, orc.blur(pyr[l])), "blurred level %d differs" % l
            xs, ys, rs = ex.tap_candidates(frame, l)
            oc = orc.level_candidates(pyr[l])
            got_set = sorted(zip(ys.tolist(), xs.tolist(), rs.tolist()))
            ref_set = sorted(zip(oc["y"].astype(int).tolist(), oc["x"].astype(int).tolist(),
                                 oc["response"].astype(int).tolist()))
            assert got_set == ref_set, "FAST candidates differ at level %d" % l
        assert ex.tap_level_counts(frame).tolist() == ocounts
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kps[f], okps[f]), f
    assert np.max(np.abs(kps["angle"] - okps["angle"]), initial=0) <= 1e-4
    assert np.array_equal(kps["angle"], okps["angle"])
    assert np.array_equal(desc, odesc)


# 6000 features: per-level quotas too large for the LDS-resident quadtree -> global-scratch kernel
@pytest.mark.parametrize("w,h,nf", [(1242, 375, 2000), (752, 480, 1000), (640, 200, 500), (1242, 375, 6000),
                                    (1242, 375, 4000)])
@pytest.mark.parametrize("variant", ["strips", "cells", "cells/1", "cells/4", "cells/16"])
def test_single_frame_all_stages(oracle_mod, w, h, nf, variant):
    # both FAST kernels: per strip (batches) and per cell (few frames; cells/N: N cells per workgroup, which reserve their
    # place in the candidate list together -- 8 by default, 1 = every cell for itself)
    name, _, group = variant.partition("/")
    variants = {"fast": name}
    if group:
        variants["fast_cell_group"] = int(group)
    ex, orc = _mk(oracle_mod, nf, w, h, variants=variants)
    img = synth.make_frames(1, w, h, seed=synth.DEFAULT_SEED + w)[0]
    kps, desc = ex(img)
    assert len(kps) > nf // 2
    _check_frame(ex, orc, img, kps, desc)


def test_descriptor_rotation_is_glibc_sincosf(oracle_mod):
    """ORBExtractor.cpp:53-54: the device's (cos, sin) against the oracle's restatement of glibc sincosf -- which
    tests/test_oracle_kat.py pins against the host libm on every float in [0, 2*pi] -- on every 257th float bit
    pattern of [0, 360] degrees (4.4 M angles), plus the neighbourhood of every multiple of 45 degrees."""
    ex, _ = _mk(oracle_mod, 500, 640, 200)
    L = oracle_mod.lib()
    top = np.array([360.0], np.float32).view(np.uint32)[0]
    bits = [np.arange(0, int(top) + 1, 257, dtype=np.uint32)]
    for q in range(0, 9):
        c = int(np.array([45.0 * q], np.float32).view(np.uint32)[0])
        bits.append(np.arange(max(c - 2048, 0), min(c + 2048, int(top)) + 1, dtype=np.uint32))
    ang = np.ascontiguousarray(np.concatenate(bits).view(np.float32))
    got = ex.tap_sincos(ang)
    want = np.zeros_like(got)
    L.orbref_sincos_deg_n(ang.ctypes.data, ang.size, want.ctypes.data)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_batch_host(oracle_mod):
    w, h, nf = 1242, 375, 2000
    ex, orc = _mk(oracle_mod, nf, w, h, batch=4)
    imgs = synth.make_frames(4, w, h)
    res = ex.extract_batch(imgs)
    for f in range(4):
        _check_frame(ex, orc, imgs[f], res[f][0], res[f][1], frame=f, stages=(f in (0, 3)))
    # fewer frames than the handle was laid out for (the records then come back in three pieces instead of one), then one
    res2 = ex.extract_batch(imgs[1:3])
    for f in range(2):
        _check_frame(ex, orc, imgs[1 + f], res2[f][0], res2[f][1], frame=f, stages=False)
    kps, desc = ex(imgs[3])
    _check_frame(ex, orc, imgs[3], kps, desc, stages=True)


def test_requota_initial_extractor(oracle_mod):
    """the 2N 'initial' extractor (reference Tracking.cpp:24, ORBExtractor.cpp:477-493)"""
    from monoorbslam3_amd.extractor import ORBExtractor
    w, h = 752, 480
    ex = ORBExtractor(1000, 1.2, 8, 20, 7)
    ex2 = ORBExtractor.requota(2000, ex)
    orc = oracle_mod.Oracle(1000, 1.2, 8, 20, 7)
    orc.requota(2000)
    assert ex2.features_per_level().tolist() == orc.quotas()
    img = synth.make_frames(1, w, h, seed=5)[0]
    kps, desc = ex2(img)
    _check_frame(ex2, orc, img, kps, desc, stages=False)


def test_edge_cases(oracle_mod):
    from monoorbslam3_amd.extractor import ORBExtractor
    ex = ORBExtractor(500, 1.2, 8, 20, 7)
    orc = oracle_mod.Oracle(500, 1.2, 8, 20, 7)
    # empty image: silent no-op (reference ORBExtractor.cpp:497)
    k, d = ex(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    # flat image: zero keypoints (reference :512)
    k, d = ex(np.full((240, 320), 90, np.uint8))
    assert len(k) == 0
    # tiny quota levels / few corners: a handful of isolated squares
    img = np.full((240, 320), 60, np.uint8)
    img[100:120, 100:130] = 200
    img[40:60, 200:230] = 10
    k, d = ex(img)
    ok, od, _ = orc.extract(img)
    assert len(k) == len(ok) and len(k) > 0
    assert np.array_equal(k["x"], ok["x"]) and np.array_equal(k["y"], ok["y"]) and np.array_equal(d, od)
    # random noise: every cell saturated with candidates, quadtree final phase exercised hard
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, size=(300, 500)).astype(np.uint8)
    k, d = ex(img)
    _check_frame(ex, orc, img, k, d)
    # non-contiguous rows (stride > width)
    big = synth.make_frames(1, 700, 300, seed=11)[0]
    view = big[:, 20:660]
    k, d = ex(view)
    _check_frame(ex, orc, np.ascontiguousarray(view), k, d, stages=False)


def test_full_hd_config4(oracle_mod):
    """BASELINE config 4 frame size (1920x1080, 2000 features): single frame, final outputs + per-level counts"""
    w, h, nf = 1920, 1080, 2000
    ex, orc = _mk(oracle_mod, nf, w, h)
    img = synth.make_frames(1, w, h, seed=4321)[0]
    kps, desc = ex(img)
    _check_frame(ex, orc, img, kps, desc, stages=False)
    assert ex.tap_level_counts(0).tolist() == orc.extract(img)[2]


@pytest.mark.parametrize("nf", [2000, 10000])
def test_uhd_frame(oracle_mod, nf):
    """3840x2160 (8.3 Mpx, 9 200 FAST cells on level 0 alone): coordinates, candidate capacities and the quadtree's list
    bounds far above the benchmark sizes; 10 000 features takes the global-scratch quadtree."""
    w, h = 3840, 2160
    ex, orc = _mk(oracle_mod, nf, w, h)
    img = synth.make_frames(1, w, h, seed=99)[0]
    kps, desc = ex(img)
    _check_frame(ex, orc, img, kps, desc, stages=False)
    assert ex.tap_level_counts(0).tolist() == orc.extract(img)[2]
    assert kps["x"].max() > 3700 and kps["y"].max() > 2000


def test_frames_from_a_page_locked_buffer(oracle_mod):
    """orbx_host_register: the frame's copy becomes an asynchronous DMA, the launches are issued while it runs -- same records,
    also when the buffer's contents change between calls (every call waits for its own copy before it returns)."""
    w, h, nf = 752, 480, 1000
    ex, orc = _mk(oracle_mod, nf, w, h)
    buf = np.empty((h, w), np.uint8)
    ex.host_register(buf)
    try:
        for seed in (31, 32, 33):
            buf[:] = synth.make_frames(1, w, h, seed=seed)[0]
            kps, desc = ex(buf)
            _check_frame(ex, orc, buf.copy(), kps, desc, stages=False)
    finally:
        ex.host_unregister(buf)
    kps, desc = ex(buf)   # pageable again
    _check_frame(ex, orc, buf, kps, desc, stages=False)


def test_odd_sizes_and_strides(oracle_mod):
    """widths that are not multiples of 4 (blur / resize edge groups), tall images (nIni = 1), tiny top levels"""
    for (w, h, nf) in ((333, 251, 300), (405, 607, 500), (130, 97, 100)):
        ex, orc = _mk(oracle_mod, nf, w, h)
        img = synth.make_frames(1, w, h, seed=w * 3 + h)[0]
        kps, desc = ex(img)
        _check_frame(ex, orc, img, kps, desc, stages=True)


@pytest.mark.parametrize("resize2", ["never", "always"])
@pytest.mark.parametrize("w,h,nf,sf,levels,batch", [(1242, 375, 2000, 1.2, 8, 1), (333, 251, 300, 1.2, 8, 3), (641, 479, 700, 1.1, 9, 2),
                                                    (800, 600, 900, 1.5, 5, 1), (1920, 1080, 2000, 1.2, 8, 1), (405, 607, 500, 1.3, 6, 26)])
def test_pyramid_two_levels_per_launch(oracle_mod, resize2, w, h, nf, sf, levels, batch):
    """ORBExtractor.cpp:559-570 through k_resize2 (levels l+1 and l+2 from one launch, ORBX_VAR_RESIZE2 = 2: forced, also for a
    batch) and through k_resize alone (= 0): every level bit-exact against the oracle either way -- widths that are not
    multiples of 4, an odd number of levels (the last one on its own), scale factors whose patches are wider (1.5: the
    launch falls back to single levels where a patch would not fit)."""
    ex, orc = _mk(oracle_mod, nf, w, h, n_levels=levels, sf=sf, batch=batch, variants={"resize2": resize2})
    imgs = synth.make_frames(batch, w, h, seed=w + 7 * h)
    if batch == 1:
        kps, desc = ex(imgs[0])
        _check_frame(ex, orc, imgs[0], kps, desc, stages=True)
    else:
        out = ex.extract_batch(imgs)
        for f in (0, batch - 1):
            _check_frame(ex, orc, imgs[f], out[f][0], out[f][1], frame=f, stages=True)


@pytest.mark.parametrize("resize_lds", ["never", "always"])
@pytest.mark.parametrize("w,h,nf,sf,levels,batch", [(1242, 375, 2000, 1.2, 8, 1), (333, 251, 300, 1.2, 8, 3), (641, 479, 700, 1.1, 9, 2),
                                                    (800, 600, 900, 1.5, 5, 1), (1920, 1080, 2000, 1.2, 8, 1), (405, 607, 500, 1.3, 6, 26),
                                                    (1000, 163, 400, 1.9, 3, 2), (64, 48, 50, 1.2, 2, 1)])
def test_pyramid_source_tile_through_lds(oracle_mod, resize_lds, w, h, nf, sf, levels, batch):
    """ORBExtractor.cpp:559-570 through k_resize_lds (the source tile of a 256 x 32 output tile staged with 16-byte loads,
    ORBX_VAR_RESIZE_LDS = 2: forced, also for a single frame; what a resident batch runs by default) and through k_resize alone
    (= 0): every level bit-exact against the oracle either way.  Widths that are not multiples of 16 (the last row of the
    caller's image ends inside a 16-byte chunk), scale factors up to 1.9 (tiles too wide for the LDS array fall back),
    levels smaller than one tile.  ORBX_VAR_RESIZE2 = 0 so that the single-level kernels run for these small calls."""
    ex, orc = _mk(oracle_mod, nf, w, h, n_levels=levels, sf=sf, batch=batch, variants={"resize_lds": resize_lds, "resize2": "never"})
    imgs = synth.make_frames(batch, w, h, seed=3 * w + 11 * h)
    if batch == 1:
        kps, desc = ex(imgs[0])
        _check_frame(ex, orc, imgs[0], kps, desc, stages=True)
    else:
        out = ex.extract_batch(imgs)
        for f in (0, batch - 1):
            _check_frame(ex, orc, imgs[f], out[f][0], out[f][1], frame=f, stages=True)


@pytest.mark.parametrize("nf,sf,levels,ini,mn", [(800, 1.5, 4, 30, 10), (1200, 1.1, 12, 12, 5), (600, 1.2, 8, 7, 20),
                                                 (500, 1.2, 1, 20, 7)])
def test_other_constructor_arguments(oracle_mod, nf, sf, levels, ini, mn):
    """pyramid depth / scale / thresholds other than the yaml defaults (incl. min threshold above ini, one level)"""
    w, h = 640, 360
    ex, orc = _mk(oracle_mod, nf, w, h, ini=ini, mn=mn, n_levels=levels, sf=sf)
    assert ex.features_per_level().tolist() == orc.quotas()
    assert np.array_equal(ex.getScaleFactors(), orc.scale_factors())
    img = synth.make_frames(1, w, h, seed=levels * 17 + ini)[0]
    kps, desc = ex(img)
    _check_frame(ex, orc, img, kps, desc, stages=True)


@pytest.mark.parametrize("nf,sf,levels,ini,mn,w,h", [(800, 1.5, 4, 30, 10, 640, 360), (1200, 1.1, 12, 12, 5, 640, 360),
                                                     (2000, 1.2, 8, 20, 7, 1242, 375), (1000, 1.3, 6, 15, 15, 1000, 163)])
def test_batch_path_with_other_arguments(oracle_mod, nf, sf, levels, ini, mn, w, h):
    """The throughput path (a host batch of 26 frames: FAST per strip, Gaussian on the matrix pipe, two-launch orientation)
    with pyramid depths / scales / thresholds / widths other than the benchmark's; every stage of three of the frames."""
    B = 26
    ex, orc = _mk(oracle_mod, nf, w, h, ini=ini, mn=mn, n_levels=levels, sf=sf, batch=B)
    imgs = synth.make_frames(B, w, h, seed=levels * 101 + w)
    res = ex.extract_batch(imgs)
    for f in (0, 11, B - 1):
        _check_frame(ex, orc, imgs[f], res[f][0], res[f][1], frame=f, stages=True)


@pytest.mark.parametrize("ini,mn,kind", [(5, 2, "noise"), (2, 1, "noise"), (2, 1, "scene"), (3, 3, "noise"), (40, 1, "scene"),
                                         (254, 200, "noise")])
@pytest.mark.parametrize("variant", ["strips", "cells", "cells/16"])  # one wave per strip of cells (batches) / one wave per cell (few frames)
def test_fast_dense_corners_and_tiny_thresholds(oracle_mod, ini, mn, kind, variant):
    """The FAST kernel's rare paths: i.i.d. noise at low thresholds makes nearly every pixel a corner (more corners per
    strip than its LDS list holds -> the NMS sweeps the score map), thresholds 0..2 make the 6-bit arc test pass pixels
    in both polarities, thresholds near 255 pass nothing.  Candidates, key points and descriptors stay the oracle's."""
    w, h, nf = 500, 300, 1500
    name, _, group = variant.partition("/")
    ex, orc = _mk(oracle_mod, nf, w, h, ini=ini, mn=mn, variants=dict({"fast": name}, **({"fast_cell_group": int(group)} if group else {})))
    if kind == "noise":
        img = np.random.RandomState(ini * 31 + mn).randint(0, 256, (h, w)).astype(np.uint8)
    else:
        img = synth.make_frames(1, w, h, seed=4242 + ini)[0]
    kps, desc = ex(img)
    _check_frame(ex, orc, img, kps, desc, stages=True)
    assert (len(kps) == 0) == (ini == 254)


def test_blur_tap_variant_and_handle_reuse(oracle_mod):
    """the plain-rounded (sum 257) Gaussian tap set, and one handle used for several frame sizes in turn"""
    from monoorbslam3_amd.extractor import ORBExtractor
    ex = ORBExtractor(700, 1.2, 8, 20, 7, blur_variant=1)
    orc = oracle_mod.Oracle(700, 1.2, 8, 20, 7, blur_variant=1)
    for (w, h) in ((480, 270), (800, 300), (480, 270)):
        img = synth.make_frames(1, w, h, seed=w)[0]
        kps, desc = ex(img)
        _check_frame(ex, orc, img, kps, desc, stages=True)


@pytest.mark.parametrize("w,h", [(1242, 375), (752, 480), (1920, 1080), (331, 77), (161, 40), (640, 9 * 4)])
@pytest.mark.parametrize("blur_variant", [0, 1])
def test_gaussian_on_the_matrix_pipe(oracle_mod, w, h, blur_variant):
    """k_blur_mfma (the batch path's 7x7 Gaussian: banded i8 matrix products) forced for a single frame: every blurred
    level byte-identical to the oracle, for both tap sets (sum 256 and the plain-rounded sum 257 with its clamp), widths
    that are not multiples of the 128-column block or the 32-column tile, and levels too small for it (VALU kernels)."""
    from monoorbslam3_amd.extractor import ORBExtractor
    ex = ORBExtractor(800, 1.2, 8, 20, 7, blur_variant=blur_variant, variants={"blur": "mfma"})
    orc = oracle_mod.Oracle(800, 1.2, 8, 20, 7, blur_variant=blur_variant)
    img = synth.make_frames(1, w, h, seed=7 * w + h)[0]
    if blur_variant == 1:
        img[: h // 3, : w // 2] = 255  # saturated area: the sum-257 taps overshoot 255 here and must clamp
    kps, desc = ex(img)
    pyr = orc.pyramid(img)
    for l in range(orc.n_levels):
        if min(pyr[l].shape) < 1:
            continue
        assert np.array_equal(ex.tap_level(0, l, w, h, blurred=True), orc.blur(pyr[l])), "blurred level %d differs" % l
    okps, odesc, _ = orc.extract(img)
    assert len(kps) == len(okps) and np.array_equal(desc, odesc)


@pytest.mark.parametrize("w,h,nf,batch,blur_variant", [(1242, 375, 2000, 1, 0), (752, 480, 1000, 3, 0), (1920, 1080, 2000, 1, 0),
                                                       (3840, 2160, 4000, 1, 0), (2600, 60, 500, 2, 0), (200, 1400, 600, 1, 1),
                                                       (640, 360, 800, 9, 1), (331, 200, 400, 2, 0), (500, 163, 700, 1, 1),
                                                       (1242, 375, 6000, 2, 0)])
def test_blur_and_descriptors_in_one_pass(oracle_mod, w, h, nf, batch, blur_variant):
    """ORBExtractor.cpp:527-532 through k_blur_desc (ORBX_VAR_DESC = 2): a workgroup walks down a block of columns, keeps the
    blurred rows in an LDS ring and samples the descriptors of the block's key points from it -- no blurred level in memory.
    Key points and 32-byte descriptors equal the oracle's for both tap sets; widths that end inside a block, levels too
    narrow for the kernel (they keep the blur pass + k_orient_desc inside the same call), quotas where a bucket holds many
    key points, frames of a batch.  The blurred-level tap still answers (the blur pass is run for it on demand)."""
    from monoorbslam3_amd.extractor import ORBExtractor
    ex = ORBExtractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=batch, blur_variant=blur_variant,
                      variants={"desc": "fused"})
    orc = oracle_mod.Oracle(nf, 1.2, 8, 20, 7, blur_variant=blur_variant)
    imgs = synth.make_frames(batch, w, h, seed=5 * w + h + nf)
    if blur_variant == 1:
        imgs[:, : h // 3, : w // 2] = 255  # saturated area: the sum-257 taps must clamp
        imgs[:, h // 2:, ::7] ^= 0x5A        # ... and texture beside it
    out = [ex(imgs[0])] if batch == 1 else ex.extract_batch(imgs)
    for f in sorted({0, batch - 1}):
        _check_frame(ex, orc, imgs[f], out[f][0], out[f][1], frame=f, stages=(f == 0))
    # the two-kernel twin on the same handle gives the same bytes
    ex.set_variant("desc", "separate")
    out2 = [ex(imgs[0])] if batch == 1 else ex.extract_batch(imgs)
    for a, b2 in zip(out, out2):
        assert np.array_equal(a[0], b2[0]) and np.array_equal(a[1], b2[1])


def test_one_pass_descriptors_edge_cases(oracle_mod):
    """k_blur_desc on what a tracker can hand it: frames without a single corner (every count zero, nothing written), one
    handle used for two frame sizes in turn and back, a capacity smaller than the key-point count on the device entry point
    (the first `cap` records are written, the count reports the need), and corners crowded into a few patches (hundreds of
    key points in one (block, trip) bucket)."""
    import torch
    from monoorbslam3_amd.extractor import ORBExtractor, KP_DTYPE
    ex = ORBExtractor(900, 1.2, 8, 20, 7, variants={"desc": "fused"})
    orc = oracle_mod.Oracle(900, 1.2, 8, 20, 7)
    flat = np.full((3, 300, 500), 90, np.uint8)
    for kps, desc in ex.extract_batch(flat):
        assert len(kps) == 0 and desc.shape == (0, 32)
    for (w, h) in ((500, 300), (801, 333), (500, 300)):
        imgs = synth.make_frames(2, w, h, seed=w + h)
        out = ex.extract_batch(imgs)
        for f in range(2):
            _check_frame(ex, orc, imgs[f], out[f][0], out[f][1], frame=f, stages=False)
    # crowded: every corner inside a few noise patches
    rng = np.random.RandomState(11)
    img = np.full((2, 360, 640), 128, np.uint8)
    for _ in range(5):
        x0, y0 = rng.randint(20, 640 - 70), rng.randint(20, 360 - 70)
        img[:, y0:y0 + 48, x0:x0 + 48] = rng.randint(0, 256, (48, 48)).astype(np.uint8)
    ex5 = ORBExtractor(3000, 1.2, 8, 5, 2, variants={"desc": "fused"})
    orc5 = oracle_mod.Oracle(3000, 1.2, 8, 5, 2)
    out = ex5.extract_batch(img)
    _check_frame(ex5, orc5, img[1], out[1][0], out[1][1], frame=1, stages=False)
    assert len(out[1][0]) > 1000
    # device entry point with a capacity below the count
    w, h = 640, 360
    fr = synth.make_frames(1, w, h, seed=3)
    okps, odesc, _ = orc.extract(fr[0])
    cap = len(okps) - 37
    d_img = torch.from_numpy(fr).cuda()
    kp = torch.zeros((1, cap + 8, 28), dtype=torch.uint8, device="cuda")
    de = torch.full((1, cap + 8, 32), 0xEE, dtype=torch.uint8, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img.data_ptr(), 1, w, h, w, w * h, kp.data_ptr(), de.data_ptr(), cap, n.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(n[0]) == len(okps)
    got = np.frombuffer(kp[0, :cap].cpu().numpy().tobytes(), KP_DTYPE)
    assert np.array_equal(got["x"], okps["x"][:cap]) and np.array_equal(de[0, :cap].cpu().numpy(), odesc[:cap])
    assert (de[0, cap:].cpu().numpy() == 0xEE).all()  # nothing beyond the capacity is touched


@pytest.mark.parametrize("w,h,nf,n_patches,side", [(1242, 375, 2000, 12, 40), (752, 480, 1000, 6, 30), (640, 200, 3000, 5, 48)])
def test_quadtree_on_crowded_corners(oracle_mod, w, h, nf, n_patches, side):
    """Every corner of the frame sits in a few small noise patches: the quadtree has to divide seven to nine times before
    the quota is reached (a scene takes three or four), which is where the kernel extends its descent codes."""
    ex, orc = _mk(oracle_mod, nf, w, h, ini=5, mn=2)  # at these thresholds nearly every noise pixel is a corner
    rng = np.random.RandomState(w + nf)
    img = np.full((h, w), 128, np.uint8)
    for _ in range(n_patches):
        x0, y0 = rng.randint(20, w - 20 - side), rng.randint(20, h - 20 - side)
        img[y0:y0 + side, x0:x0 + side] = rng.randint(0, 256, (side, side))
    kps, desc = ex(img)
    assert len(kps) > nf * 0.9  # the quota is reached: the tree went as deep as the patches are dense
    _check_frame(ex, orc, img, kps, desc, stages=True)


@pytest.mark.parametrize("w,h,nf,n_patches,side,scene", [(1920, 1080, 2000, 10, 60, False), (1920, 1080, 3000, 3, 90, False),
                                                          (1920, 1080, 2000, 6, 50, True), (1500, 900, 1500, 4, 64, True),
                                                          (2600, 700, 2500, 8, 40, True)])
def test_quadtree_of_a_megapixel_level(oracle_mod, w, h, nf, n_patches, side, scene):
    """A single frame with levels of a megapixel and more: their quadtree runs its passes from a table of candidate counts per
    descent path (no candidate sweep per pass) and goes back to sweeps when a node below the table's depth still holds several
    candidates -- noise patches force that, on a flat background from the first passes on, on a scene after a few.  Same key
    points in the same order as the reference's list (ORBextractor.cc:640-830)."""
    ex, orc = _mk(oracle_mod, nf, w, h, ini=5 if not scene else 20, mn=2 if not scene else 7)
    rng = np.random.RandomState(w + nf + side)
    img = synth.make_frames(1, w, h, seed=777 + side)[0] if scene else np.full((h, w), 128, np.uint8)
    for _ in range(n_patches):
        x0, y0 = rng.randint(20, w - 20 - side), rng.randint(20, h - 20 - side)
        img[y0:y0 + side, x0:x0 + side] = rng.randint(0, 256, (side, side))
    kps, desc = ex(img)
    _check_frame(ex, orc, img, kps, desc, stages=False)
    assert ex.tap_level_counts(0).tolist() == orc.extract(img)[2]


def test_batch_device_pointers_and_determinism(oracle_mod):
    """HBM-resident batch API (the bench path): two runs give identical bytes, and they match the oracle"""
    import torch
    from monoorbslam3_amd.extractor import ORBExtractor, KP_DTYPE
    w, h, nf, B = 752, 480, 1000, 24
    imgs = synth.make_frames(B, w, h, seed=31)
    ex = ORBExtractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    cap = ex.max_keypoints(w, h)
    d_img = torch.from_numpy(imgs).cuda()
    outs = []
    for _ in range(2):
        d_kp = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
        d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
        ex.extract_batch_device(d_img.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
        ex.synchronize()
        outs.append((d_n.cpu().numpy(), d_kp.cpu().numpy(), d_desc.cpu().numpy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    orc = oracle_mod.Oracle(nf, 1.2, 8, 20, 7)
    n, kp, desc = outs[0]
    for f in (0, 7, B - 1):
        ok, od, _ = orc.extract(imgs[f])
        assert n[f] == len(ok)
        got = kp[f, :n[f]].copy().view(KP_DTYPE).reshape(-1)
        for fld in ("x", "y", "angle", "response", "octave"):
            assert np.array_equal(got[fld], ok[fld]), fld
        assert np.array_equal(desc[f, :n[f]], od)


def test_golden_fixtures_through_the_c_abi(oracle_mod):
    """the committed fixtures (tests/golden, oracle-generated) reproduced by the HIP path alone: extraction, bag of
    words, undistortion + grid, map-point descriptors"""
    import os
    from monoorbslam3_amd.extractor import ORBExtractor
    from monoorbslam3_amd.frame import FramePost
    from monoorbslam3_amd.matcher import ORBMatcher
    from monoorbslam3_amd.vocabulary import ORBVocabulary
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    e = np.load(os.path.join(gold, "extract_320x240_n300.npz"))
    z = np.load(os.path.join(gold, "records_320x240.npz"))
    ex = ORBExtractor(300, 1.2, 8, 20, 7)
    kps, desc = ex(e["image"])
    for f in ("x", "y", "angle", "response", "octave"):
        assert np.array_equal(kps[f], e[f]), f
    assert np.array_equal(desc, e["desc"])
    k, L, sc, wt = (int(v) for v in z["voc_hdr"])
    V = ORBVocabulary.from_arrays(dict(k=k, L=L, scoring=sc, weighting=wt, parent=z["voc_parent"], is_leaf=z["voc_leaf"],
                                       desc=z["voc_desc"], weight=z["voc_weight"]))
    bi, bv, (fn, fo, fi) = V.transform(desc, 1)
    assert np.array_equal(bi, z["bow_ids"]) and bv.tobytes() == z["bow_vals"].tobytes()
    assert np.array_equal(fn, z["fv_nodes"]) and np.array_equal(fo, z["fv_off"]) and np.array_equal(fi, z["fv_idx"])
    w, h, fx, fy, cx, cy = z["cam"]
    _, un, start, items = FramePost(int(w), int(h), float(fx), float(fy), float(cx), float(cy), dist=tuple(z["dist"]))(kps)
    assert np.array_equal(un["x"], z["un_x"]) and np.array_equal(un["y"], z["un_y"])
    assert np.array_equal(start, z["cell_start"]) and np.array_equal(items, z["cell_items"])
    assert ORBMatcher.ComputeDistinctiveDescriptors(z["g_desc"], z["g_off"]).tolist() == z["medoid"].tolist()


def test_stream_layout_variants_give_identical_results(tmp_path):
    """The internal stream layouts (ORBX_VAR_STREAMS / _SIDE_BLUR / _EARLY_FAST / _DESC) only change what overlaps with what
    and which twin of a kernel runs: every variant must return the default layout's bytes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "variant.py"
    script.write_text(
        "import sys, hashlib, json\n"
        "sys.path.insert(0, %r)\n"
        "import torch\n"
        "from monoorbslam3_amd import synth\n"
        "from monoorbslam3_amd.extractor import ORBExtractor\n"
        "B, W, H = 24, 640, 360\n"
        "fr = torch.from_numpy(synth.make_frames(B, W, H, seed=5)).cuda()\n"
        "ex = ORBExtractor(800, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, variants=json.loads(sys.argv[1]))\n"
        "cap = ex.max_keypoints(W, H)\n"
        "kp = torch.zeros((B, cap, 28), dtype=torch.uint8, device='cuda')\n"
        "de = torch.zeros((B, cap, 32), dtype=torch.uint8, device='cuda')\n"
        "n = torch.zeros(B, dtype=torch.int32, device='cuda')\n"
        "s = torch.cuda.Stream()\n"
        "for _ in range(2):\n"
        "    ex.extract_batch_device(fr.data_ptr(), B, W, H, W, W * H, kp.data_ptr(), de.data_ptr(), cap, n.data_ptr(), s.cuda_stream)\n"
        "s.synchronize()\n"
        "h = hashlib.sha256()\n"
        "nn = n.cpu().numpy()\n"
        "for f in range(B):\n"
        "    h.update(kp[f, :nn[f]].cpu().numpy().tobytes()); h.update(de[f, :nn[f]].cpu().numpy().tobytes())\n"
        "print(int(nn.sum()), h.hexdigest())\n" % root)
    results = {}
    import json
    for name, var in (("default", {}), ("streams3", {"streams": 3}), ("no_side", {"side_blur": 0}),
                      ("blur_after_fast", {"side_blur": 2}), ("blur_beside_orientation", {"side_blur": 3}),
                      ("no_early_fast", {"early_fast": 0}), ("early_blur", {"early_fast": 2}), ("desc_separate", {"desc": "separate"}),
                      ("desc_fused", {"desc": "fused"}), ("desc_fused_streams3", {"desc": "fused", "streams": 3}),
                      ("desc_fused_blur_valu", {"desc": "fused", "blur": "valu"}), ("copy_back", {"zero_copy": 0})):
        results[name] = subprocess.check_output([sys.executable, str(script), json.dumps(var)], text=True).strip().splitlines()[-1]
    assert len(set(results.values())) == 1, results
    assert int(results["default"].split()[0]) > 24 * 500


def test_full_bench_batch_sampled_parity_and_checksum(oracle_mod):
    """The benchmark's own shape: 512 resident 1242x375 frames, 2000 features, through the device entry point.  Two runs
    give the same bytes (checksum over all counts, key points and descriptors), and sampled frames -- the first, the last,
    two in the middle, one of them a noise-perturbed repeat as bench.py makes them -- equal the oracle's output."""
    import hashlib
    import torch
    from monoorbslam3_amd.extractor import ORBExtractor, KP_DTYPE
    w, h, nf, B, n_distinct = 1242, 375, 2000, 512, 16
    base = synth.make_frames(n_distinct, w, h, seed=synth.DEFAULT_SEED + 5)
    rng = np.random.RandomState(7)
    imgs = np.repeat(base[None], B // n_distinct, axis=0).reshape(B, h, w).astype(np.int16)
    imgs[n_distinct:] += rng.randint(-2, 3, imgs[n_distinct:].shape).astype(np.int16)
    imgs = np.clip(imgs, 0, 255).astype(np.uint8)
    ex = ORBExtractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    cap = ex.max_keypoints(w, h)
    d_img = torch.from_numpy(imgs).cuda()
    sums = []
    for _ in range(2):
        d_kp = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
        d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
        ex.extract_batch_device(d_img.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
        ex.synchronize()
        n, kp, desc = d_n.cpu().numpy(), d_kp.cpu().numpy(), d_desc.cpu().numpy()
        hsh = hashlib.sha256()
        hsh.update(n.tobytes())
        for f in range(B):
            hsh.update(kp[f, :n[f]].tobytes())
            hsh.update(desc[f, :n[f]].tobytes())
        sums.append(hsh.hexdigest())
    assert sums[0] == sums[1]
    orc = oracle_mod.Oracle(nf, 1.2, 8, 20, 7)
    for f in (0, 200, 301, B - 1):
        ok, od, _ = orc.extract(imgs[f])
        assert n[f] == len(ok)
        got = kp[f, :n[f]].copy().view(KP_DTYPE).reshape(-1)
        for fld in ("x", "y", "size", "angle", "response", "octave"):
            assert np.array_equal(got[fld], ok[fld]), (f, fld)
        assert np.array_equal(desc[f, :n[f]], od), f


def _random_cases():
    rng = np.random.RandomState(20261004)
    cases = []
    for i in range(14):
        w = int(rng.randint(96, 900))
        h = int(rng.randint(64, 500))
        cases.append((i, w, h, int(rng.choice([300, 800, 1500, 2500])), float(rng.choice([1.1, 1.2, 1.25, 1.44])),
                      int(rng.randint(2, 11)), int(rng.randint(5, 40)), int(rng.randint(2, 20)), int(rng.choice([1, 9, 25]))))
    return cases


@pytest.mark.parametrize("i,w,h,nf,sf,levels,ini,mn,B", _random_cases())
def test_random_shapes_and_arguments(oracle_mod, i, w, h, nf, sf, levels, ini, mn, B):
    """Seeded random frame sizes (odd widths, heights down to 64), feature counts, scale factors, pyramid depths and
    thresholds, as a single call (latency path), 9 frames (per-cell FAST + matrix-pipe Gaussian) or 25 frames (throughput
    path): every stage of the first frame and the records of the last against the oracle."""
    ex, orc = _mk(oracle_mod, nf, w, h, ini=ini, mn=mn, n_levels=levels, sf=sf, batch=B)
    imgs = synth.make_frames(B, w, h, seed=1000 + i)
    if B == 1:
        kps, desc = ex(imgs[0])
        _check_frame(ex, orc, imgs[0], kps, desc, stages=True)
        return
    res = ex.extract_batch(imgs)
    _check_frame(ex, orc, imgs[0], res[0][0], res[0][1], frame=0, stages=True)
    _check_frame(ex, orc, imgs[B - 1], res[B - 1][0], res[B - 1][1], frame=B - 1, stages=False)
