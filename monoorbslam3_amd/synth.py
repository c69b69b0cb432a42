"""Seeded synthetic frames and descriptor sets for parity tests and bench.py.

Follows SURVEY.md section 8(d): corner-rich u8 images (low-pass noise, a few
hundred rectangles / rotated rectangles / discs with +-30..+-120 contrast,
+-2 pixel noise, plus a flat band so that some FAST cells hit the min-threshold
fallback and some stay empty).  The shape count is calibrated so that a
1242x375 frame yields ~15-20 k FAST candidates over the 8 levels (the survey's
"5-20 k candidates -> ~2000 keypoints"): every level still exceeds its quota.
numpy only, so the same seed gives the same bytes here and on the GPU box.
"""
import numpy as np

DEFAULT_SEED = 20261004


def _box_blur(a, r):
    """Separable running-mean blur (float64), edge-replicated."""
    if r <= 0:
        return a
    k = 2 * r + 1
    for axis in (0, 1):
        pad = [(0, 0), (0, 0)]
        pad[axis] = (r + 1, r)
        p = np.pad(a, pad, mode="edge")
        c = np.cumsum(p, axis=axis)
        if axis == 0:
            a = (c[k:, :] - c[:-k, :]) / k
        else:
            a = (c[:, k:] - c[:, :-k]) / k
    return a


def make_canvas(width, height, seed=DEFAULT_SEED, n_shapes=None):
    """One corner-rich u8 canvas of size height x width."""
    rng = np.random.RandomState(seed)
    base = rng.uniform(0.0, 1.0, size=(height, width))
    base = _box_blur(_box_blur(base, 4), 4)
    lo, hi = base.min(), base.max()
    img = 40.0 + (base - lo) / max(hi - lo, 1e-9) * 175.0
    if n_shapes is None:
        n_shapes = int(500 * (width * height) / (1242.0 * 375.0))
    for _ in range(n_shapes):
        kind = rng.randint(0, 3)
        cx = rng.randint(0, width)
        cy = rng.randint(0, height)
        r = rng.randint(4, 30)
        amp = rng.randint(30, 121) * (1 if rng.randint(0, 2) else -1)
        x0, x1 = max(cx - r, 0), min(cx + r + 1, width)
        y0, y1 = max(cy - r, 0), min(cy + r + 1, height)
        if x1 <= x0 or y1 <= y0:
            continue
        if kind == 0:  # axis-aligned rectangle with random aspect
            ry = rng.randint(2, r + 1)
            y0b, y1b = max(cy - ry, 0), min(cy + ry + 1, height)
            img[y0b:y1b, x0:x1] += amp
        else:
            ly, lx = np.mgrid[y0 - cy:y1 - cy, x0 - cx:x1 - cx]
            if kind == 1:  # disc
                m = lx * lx + ly * ly <= r * r
            else:  # rotated rectangle
                th = rng.uniform(0, np.pi)
                c, s = np.cos(th), np.sin(th)
                u = lx * c + ly * s
                v = -lx * s + ly * c
                m = (np.abs(u) <= r * 0.8) & (np.abs(v) <= r * rng.uniform(0.2, 0.7))
            img[y0:y1, x0:x1] += amp * m
    img += rng.randint(-2, 3, size=(height, width))
    # flat band (fallback / empty cells) across ~8% of the height, with faint texture
    b0 = int(height * 0.55)
    b1 = b0 + max(int(height * 0.08), 8)
    img[b0:b1, :] = 128.0 + rng.randint(-1, 2, size=(b1 - b0, width))
    # a low-contrast strip: corners only at the min threshold
    s0 = int(height * 0.30)
    s1 = s0 + max(int(height * 0.06), 8)
    strip = img[s0:s1, :]
    img[s0:s1, :] = 128.0 + (strip - strip.mean()) * 0.12
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_frames(n, width, height, seed=DEFAULT_SEED, n_shapes=None):
    """n distinct frames (n, height, width) u8: shifted crops of one larger canvas
    plus per-frame +-2 noise, cheap enough for batches of hundreds.  n_shapes (default: the corner-rich calibration of
    make_canvas, ~500 per 1242x375) sets the corner density: ~50 / 150 give about a tenth / a third of it."""
    rng = np.random.RandomState(seed + 7)
    mx, my = 64, 48
    if n_shapes is not None:
        n_shapes = int(round(n_shapes * (width + mx) * (height + my) / (1242.0 * 375.0)))
    canvas = make_canvas(width + mx, height + my, seed, n_shapes=n_shapes)
    out = np.empty((n, height, width), dtype=np.uint8)
    for i in range(n):
        ox = rng.randint(0, mx + 1)
        oy = rng.randint(0, my + 1)
        f = canvas[oy:oy + height, ox:ox + width].astype(np.int16)
        if i > 0:
            f = f + rng.randint(-2, 3, size=f.shape).astype(np.int16)
        out[i] = np.clip(f, 0, 255).astype(np.uint8)
    return out


def fast9_corner_fraction(img, threshold=20):
    """Fraction of the pixels of `img` (3-px frame excluded) that are FAST-9/16 corners at `threshold` (numpy, for reporting the
    density of a synthetic workload; not used by any parity check)."""
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
            (-3, 1), (-2, 2), (-1, 3)]
    im = np.asarray(img).astype(np.int16)
    h, w = im.shape
    v = im[3:h - 3, 3:w - 3]
    p = np.stack([im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in ring])
    out = np.zeros(v.shape, bool)
    for fl in (p < v - threshold, p > v + threshold):
        ext = np.concatenate([fl, fl[:8]])
        run = np.ones((16,) + v.shape, bool)
        for i in range(9):
            run &= ext[i:i + 16]
        out |= run.any(axis=0)
    return float(out.mean())


def make_descriptor_pair(n, seed=DEFAULT_SEED, flip_p=0.1):
    """Set A: n x 32 i.i.d. bytes.  Set B: A with each bit flipped w.p. flip_p,
    then row-permuted.  Returns (A, B, perm) with B[i] derived from A[perm[i]]."""
    rng = np.random.RandomState(seed + 13)
    a = rng.randint(0, 256, size=(n, 32)).astype(np.uint8)
    flips = (rng.uniform(size=(n, 256)) < flip_p)
    fb = np.packbits(flips, axis=1, bitorder="little")
    perm = rng.permutation(n)
    b = (a ^ fb)[perm]
    return a, np.ascontiguousarray(b), perm


def feature_vector_by_prefix(desc, bits):
    """Synthetic DBoW2 FeatureVector: bucket descriptors by their first `bits`
    bits (little-endian in byte 0/1).  Returns CSR (node_ids, offsets, indices)
    with indices ascending inside a node (FeatureVector.cpp:31-45 order)."""
    key = (desc[:, 0].astype(np.uint32) | (desc[:, 1].astype(np.uint32) << 8)) & ((1 << bits) - 1)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    node_ids, starts = np.unique(ks, return_index=True)
    offsets = np.concatenate([starts, [len(ks)]]).astype(np.int32)
    return node_ids.astype(np.uint32), offsets, order.astype(np.uint32)


def make_ba_problem(n_poses=20, n_points=3000, seed=DEFAULT_SEED, width=1242, height=375, n_fixed=4, camera="pinhole"):
    """BASELINE config 5: poses on a 10 m arc looking at a 20 x 20 x 10 m box of points, pinhole
    fx = fy = 718.856, cx = 607.19, cy = 185.22; every point observed by every pose where it projects inside
    the image; 1 px Gaussian pixel noise; octave-dependent 1/sigma^2.  Edges are grouped by point.
    camera="fisheye": the same scene through a Kannala-Brandt camera (the reference's Fisheye model, 512 x 512, a
    TUM-VI-like calibration); cam is then (fx, fy, cx, cy, k1, k2, k3, k4)."""
    rng = np.random.RandomState(seed + 29)
    cam = (718.856, 718.856, 607.19, 185.22)
    if camera == "fisheye":
        cam = (190.978477, 190.973307, 254.931706, 256.897442, 0.0034823894, 0.0007150348, -0.0020532361, 0.00020293673)
        width = height = 512

    def proj(pc):
        if len(cam) == 4:
            return cam[0] * pc[0] / pc[2] + cam[2], cam[1] * pc[1] / pc[2] + cam[3]
        a, b = pc[0] / pc[2], pc[1] / pc[2]
        r = np.sqrt(a * a + b * b)
        th = np.arctan(r)
        k = [float(np.float32(v)) for v in cam[4:]]
        thd = th + k[0] * th ** 3 + k[1] * th ** 5 + k[2] * th ** 7 + k[3] * th ** 9
        return cam[0] * thd * a / r + cam[2], cam[1] * thd * b / r + cam[3]
    pts = np.stack([rng.uniform(-10, 10, n_points), rng.uniform(-5, 5, n_points), rng.uniform(8, 28, n_points)], 1)
    R, t = [], []
    for k in range(n_poses):
        a = (k / max(n_poses - 1, 1) - 0.5) * 0.35            # yaw along the arc
        c = np.array([10.0 * np.sin(a), 0.05 * k, 10.0 * (1 - np.cos(a))])   # camera centre
        Rwc = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        Rcw = Rwc.T
        R.append(Rcw)
        t.append(-Rcw @ c)
    R, t = np.array(R), np.array(t)
    ep, el, z, w = [], [], [], []
    for j in range(n_points):
        for k in range(n_poses):
            pc = R[k] @ pts[j] + t[k]
            if pc[2] <= 0.5:
                continue
            u, v = proj(pc)
            if 0 <= u < width and 0 <= v < height:
                octave = rng.randint(0, 8)
                sig = np.float32(1.2) ** octave
                ep.append(k); el.append(j)
                z.append([u + rng.normal(0, 1.0), v + rng.normal(0, 1.0)])
                w.append(float(np.float32(1.0) / np.float32(sig) / np.float32(sig)))
    # perturb the estimates so that residuals and Huber outliers are non-trivial
    pts_est = pts + rng.normal(0, 0.05, pts.shape)
    t_est = t + rng.normal(0, 0.01, t.shape)
    z = np.array(z)
    bad = rng.uniform(size=len(z)) < 0.03
    z[bad] += rng.normal(0, 12.0, (int(bad.sum()), 2))
    fixed = np.zeros(n_poses, np.uint8)
    fixed[:n_fixed] = 1
    return {"cam": cam, "pose_R": R.reshape(n_poses, 9), "pose_t": t_est, "pose_fixed": fixed, "points": pts_est,
            "edge_pose": np.array(ep, np.int32), "edge_point": np.array(el, np.int32), "edge_z": z,
            "edge_inv_sigma2": np.array(w)}


def make_vocabulary(k=10, L=4, seed=DEFAULT_SEED, p_early_leaf=0.02, p_stop=0.02, flip_bits=40, shuffle=True):
    """Synthetic DBoW2 vocabulary tree in loadFromTextFile's node order (TemplatedVocabulary.h:1376-1417): node 0 is the
    root; every node's parent has a smaller id.  A child descriptor is its parent's with `flip_bits` random bits flipped
    (so a descent is decided by real distances); a few inner nodes end early as leaves (k-means clusters that ran out of
    points) and a few words have weight 0 ("stopped").  With shuffle=True the ids of a level are permuted, so children
    are NOT contiguous in id order.  Returns dict(k, L, scoring, weighting, parent, is_leaf, desc, weight)."""
    rng = np.random.RandomState(seed + 41)
    parent, leaf, desc = [np.zeros(1, np.int64)], [np.zeros(1, bool)], [np.zeros((1, 32), np.uint8)]
    prev_ids, prev_desc, next_id = np.zeros(1, np.int64), rng.randint(0, 256, (1, 32)).astype(np.uint8), 1
    for level in range(1, L + 1):
        n = len(prev_ids) * k
        par = np.repeat(prev_ids, k)
        d = np.repeat(prev_desc, k, axis=0)
        flips = np.zeros((n, 256), bool)
        cols = rng.randint(0, 256, (n, flip_bits if level > 1 else 128))
        flips[np.arange(n)[:, None], cols] = True
        d = d ^ np.packbits(flips, axis=1, bitorder="little")
        if shuffle:
            order = rng.permutation(n)
            par, d = par[order], d[order]
        ids = np.arange(next_id, next_id + n)
        next_id += n
        is_leaf = np.full(n, level == L) | (rng.uniform(size=n) < p_early_leaf if level >= 2 else False)
        parent.append(par); leaf.append(is_leaf); desc.append(d)
        keep = ~is_leaf
        prev_ids, prev_desc = ids[keep], d[keep]
    parent = np.concatenate(parent).astype(np.int32)
    is_leaf = np.concatenate(leaf).astype(np.uint8)
    desc = np.ascontiguousarray(np.concatenate(desc))
    # weights as a text file would hold them: 6 significant digits (ostream default, :1447), idf-like magnitudes
    weight = np.array([float("%.6g" % v) for v in rng.uniform(0.5, 12.0, len(parent))])
    weight[(rng.uniform(size=len(parent)) < p_stop) & (is_leaf > 0)] = 0.0
    weight[0] = 0.0
    return dict(k=k, L=L, scoring=0, weighting=0, parent=parent, is_leaf=is_leaf, desc=desc, weight=weight)


def write_vocabulary_text(voc, path, trailing_newline=True):
    """TemplatedVocabulary::saveToTextFile's format (:1429-1447): 'k L  scoring weighting' then per node
    'parent is_leaf d0 ... d31  weight'."""
    with open(path, "w") as f:
        f.write("%d %d  %d %d\n" % (voc["k"], voc["L"], voc["scoring"], voc["weighting"]))
        lines = []
        for i in range(1, len(voc["parent"])):
            lines.append("%d %d %s  %.6g" % (voc["parent"][i], 1 if voc["is_leaf"][i] else 0,
                                            " ".join(str(int(b)) for b in voc["desc"][i]), voc["weight"][i]))
        f.write("\n".join(lines))
        if trailing_newline:
            f.write("\n")


def make_descriptors_near_words(voc, n, seed=DEFAULT_SEED, flip_bits=25):
    """n descriptors, each a random leaf's descriptor with `flip_bits` bits flipped."""
    rng = np.random.RandomState(seed + 43)
    leaves = np.flatnonzero(voc["is_leaf"])
    d = voc["desc"][rng.choice(leaves, n)].copy()
    flips = np.zeros((n, 256), bool)
    flips[np.arange(n)[:, None], rng.randint(0, 256, (n, flip_bits))] = True
    return np.ascontiguousarray(d ^ np.packbits(flips, axis=1, bitorder="little"))
