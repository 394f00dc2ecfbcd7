"""Pins the CPU oracle with the known answers that can be derived from the reference's
own constants and code (no OpenCV is available to compare against): per-level quotas,
the circular-patch table, level sizes, cell grids, the blur kernel, rounding rules, FAST
on hand-built patterns, quadtree invariants."""
import math

import numpy as np
import pytest


def test_quotas_match_reference_formula(oracle):
    # mnFeaturesPerLevel from core/lib/ORB/extractor.cc:428-440, evaluated by hand
    want = {1000: [217, 181, 151, 126, 105, 87, 73, 60],
            1200: [261, 217, 181, 151, 126, 105, 87, 72],
            1500: [326, 271, 226, 189, 157, 131, 109, 91],
            2000: [434, 362, 302, 251, 209, 175, 145, 122],
            10000: [2172, 1810, 1508, 1257, 1047, 873, 727, 606],
            5000: [1086, 905, 754, 628, 524, 436, 364, 303]}
    for nf, q in want.items():
        t = oracle.Extractor(nf, 1.2, 8, 20, 7).tables()
        assert t["quota"].tolist() == q
        assert sum(q) == nf


def test_umax_and_scale_tables(oracle):
    t = oracle.Extractor(1000, 1.2, 8, 20, 7).tables()
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # circular patch has 749 pixels
    assert sum(2 * u + 1 for u in t["umax"][1:]) * 2 + 2 * t["umax"][0] + 1 == 749
    s = np.float32(1.0)
    for l in range(8):
        assert t["scale"][l] == s
        assert t["inv"][l] == np.float32(1.0) / s
        assert t["sigma2"][l] == s * s
        s = np.float32(np.float64(s) * np.float64(np.float32(1.2)))
    # scaledPatchSize = (int)(31 * scale), extractor.cc:840
    assert [int(np.float32(31) * x) for x in t["scale"]] == [31, 37, 44, 53, 64, 77, 92, 111]


@pytest.mark.parametrize("wh,sizes", [
    ((752, 480), [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]),
    ((512, 512), [(512, 512), (427, 427), (356, 356), (296, 296), (247, 247), (206, 206), (171, 171), (143, 143)]),
    ((1241, 376), [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]),
])
def test_level_sizes(oracle, wh, sizes):
    e = oracle.Extractor(100, 1.2, 8, 20, 7)
    img = np.zeros((wh[1], wh[0]), np.uint8)
    assert e.extract(img)[0] == 0
    assert [e.level_dims(l)[:2] for l in range(8)] == sizes


def cells_of(w, h):
    width, height = w - 32, h - 32
    return int(width / 35) * int(height / 35)


def test_cell_counts():
    # extractor.cc:762-768
    sizes = [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]
    assert sum(cells_of(w, h) for w, h in sizes) == 700


def test_cv_round_is_half_to_even(oracle):
    for v, want in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (2.4999, 2), (2.5001, 3), (-2.5, -2)]:
        assert oracle.cv_round(v) == want


def test_fast_atan2(oracle):
    assert oracle.fast_atan2(0.0, 1.0) == 0.0
    for deg in range(0, 360, 7):
        r = math.radians(deg)
        got = oracle.fast_atan2(100 * math.sin(r), 100 * math.cos(r))
        err = abs(got - deg)
        assert min(err, 360 - err) < 0.3  # polynomial accuracy
    assert abs(oracle.fast_atan2(1.0, 1.0) - 45.0) < 0.3
    assert abs(oracle.fast_atan2(-1.0, 0.0) - 270.0) < 0.3
    assert 0.0 <= oracle.fast_atan2(-1e-3, 1.0) < 360.0


def test_resize_of_constant_is_constant(oracle):
    for v in (0, 1, 127, 255):
        src = np.full((120, 200), v, np.uint8)
        assert (oracle.resize(src, 167, 100) == v).all()


def test_resize_interpolates_between_neighbours(oracle):
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (60, 90), dtype=np.uint8)
    dst = oracle.resize(src, 75, 50)
    # every output is within the range of its 2x2 source support
    sx = np.floor((np.arange(75) + 0.5) * 90 / 75 - 0.5).astype(int).clip(0, 88)
    sy = np.floor((np.arange(50) + 0.5) * 60 / 50 - 0.5).astype(int).clip(0, 58)
    quad = np.stack([src[np.ix_(sy + a, sx + b)] for a in (0, 1) for b in (0, 1)])
    assert (dst >= quad.min(0)).all() and (dst <= quad.max(0)).all()


def test_blur_kernel(oracle):
    img = np.zeros((41, 41), np.uint8)
    img[20, 20] = 255
    out = oracle.gauss7(img)
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    assert k.sum() == 256
    want = (255 * np.outer(k, k) + 32768) >> 16
    assert np.array_equal(out[17:24, 17:24], want)
    assert out.sum() == want.sum()
    for v in (0, 9, 200, 255):
        assert (oracle.gauss7(np.full((30, 50), v, np.uint8)) == v).all()


def test_blur_border_is_reflect101(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 56), dtype=np.uint8)
    pad = np.pad(img, 3, mode="reflect")  # numpy 'reflect' == REFLECT_101
    assert np.array_equal(oracle.gauss7(img), oracle.gauss7(pad)[3:-3, 3:-3])


def ring_image(center, ring_vals):
    circle = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3),
              (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    img = np.full((9, 9), center, np.uint8)
    for (dx, dy), v in zip(circle, ring_vals):
        img[4 + dy, 4 + dx] = v
    return img


def test_fast_hand_built_patterns(oracle):
    # 9 contiguous brighter by 50, rest equal -> corner, score = 50 - 1
    img = ring_image(100, [150] * 9 + [100] * 7)
    got = oracle.fast(img, 20)
    assert [tuple(r) for r in got if (r[0], r[1]) == (4, 4)] == [(4, 4, 49)]
    # only 8 contiguous -> not a corner at the centre
    img = ring_image(100, [150] * 8 + [100] * 8)
    assert not any((r[0], r[1]) == (4, 4) for r in oracle.fast(img, 20))
    # darker arc, uneven: score = min over the best 9-arc minus one
    img = ring_image(200, [100, 110, 120, 130, 140, 150, 160, 170, 180] + [200] * 7)
    got = [tuple(r) for r in oracle.fast(img, 10) if (r[0], r[1]) == (4, 4)]
    assert got == [(4, 4, 19)]
    # threshold is strict: difference == threshold is not enough
    img = ring_image(100, [120] * 9 + [100] * 7)
    assert not any((r[0], r[1]) == (4, 4) for r in oracle.fast(img, 20))
    assert any((r[0], r[1]) == (4, 4) for r in oracle.fast(img, 19))


def test_fast_score_is_threshold_independent_and_nms_is_strict(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (60, 60), dtype=np.uint8)
    lo = {(x, y): s for x, y, s in oracle.fast(img, 7)}
    hi = {(x, y): s for x, y, s in oracle.fast(img, 20)}
    assert hi and set(hi) <= set(lo)
    assert all(lo[k] == v and v >= 20 for k, v in hi.items())
    # survivors at the low threshold that are missing at the high one score below it
    assert all(s < 20 for k, s in lo.items() if k not in hi)
    # two equal adjacent maxima suppress each other: mirror-symmetric image
    sym = np.concatenate([img[:, :30], img[:, 29::-1]], axis=1)
    for x, y, s in oracle.fast(sym, 7):
        assert not (x in (29, 30))


def test_octree_invariants(oracle):
    rng = np.random.default_rng(3)
    for n, N in [(0, 50), (1, 50), (30, 50), (500, 60), (5000, 217), (5000, 1000)]:
        pts = set()
        while len(pts) < n:
            pts.add((int(rng.integers(0, 714)), int(rng.integers(0, 442))))
        xys = np.array([[x, y, int(rng.integers(7, 200))] for x, y in sorted(pts, key=lambda p: (p[1] // 38, p[0] // 36, p[1], p[0]))],
                       np.int32).reshape(-1, 3)
        sel = oracle.octree(xys, 16, 736, 16, 464, N)
        assert len(set(sel.tolist())) == len(sel)
        if n <= N:
            assert len(sel) == n
        else:
            assert N <= len(sel) <= N + 3


def test_extract_invariants(oracle):
    import orb_slam3_amd as orbx
    img = orbx.textured_frame(3, 640, 400)
    e = oracle.Extractor(800, 1.2, 8, 20, 7)
    for lap in [(0, 0), (0, 1000), (200, 420)]:
        st, mono, k, d = e.extract(img, lap)
        assert st == 0 and 0 <= mono <= len(k)
        inside = (k["x"] >= lap[0]) & (k["x"] <= lap[1])
        assert not inside[:mono].any() and inside[mono:].all()
        sc = e.tables()["scale"]
        for l in range(8):
            w, h, _ = e.level_dims(l)
            kl, _ = e.keys(l)
            assert ((kl["x"] >= 19) & (kl["x"] <= w - 20) & (kl["y"] >= 19) & (kl["y"] <= h - 20)).all()
        assert (k["size"] == np.float32(31) * sc[k["octave"]] // 1).all()
        assert ((k["angle"] >= 0) & (k["angle"] < 360)).all()
    # the all-stereo fill is the reverse of the all-mono fill
    _, m0, k0, d0 = e.extract(img, (0, 0))
    _, m1, k1, d1 = e.extract(img, (0, 100000))
    assert m0 == len(k0) and m1 == 0
    assert np.array_equal(k0[::-1], k1) and np.array_equal(d0[::-1], d1)


def test_error_statuses(oracle):
    e = oracle.Extractor(100, 1.2, 8, 20, 7)
    assert e.extract(np.zeros((100, 100), np.uint8))[0] == -2   # last level has no 35-px cell
    assert e.extract(np.zeros((480, 752), np.uint8))[2].size == 0


def test_hamming_and_knn_oracle_against_numpy(oracle):
    rng = np.random.default_rng(4)
    q = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t[100] = q[0]
    t[200] = q[0]
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
    assert oracle.hamming(q[3], t[5]) == D[3, 5]
    idx, dist = oracle.knn2(q, t)
    order = np.lexsort((np.broadcast_to(np.arange(300), D.shape), D), axis=1)  # by (dist, idx)
    assert np.array_equal(idx, order[:, :2].astype(np.uint32))
    assert np.array_equal(dist, np.take_along_axis(D, order[:, :2], 1).astype(np.uint16))
    assert idx[0].tolist() == [100, 200] and dist[0].tolist() == [0, 0]
    i1, d1 = oracle.knn2(q[:2], t[:1])
    assert i1[:, 1].tolist() == [0xFFFFFFFF] * 2 and d1[:, 1].tolist() == [0xFFFF] * 2


def test_stereo_match_constant_disparity(oracle):
    """Right image = left image shifted by 12 columns: every surviving match must report a
    disparity of 12 within the sub-pixel step of its pyramid level, depth = mbf / disparity,
    and the median rule (Frame.cc:984-995) must only ever remove matches."""
    from orb_slam3_amd.synth import textured_frame
    wide = textured_frame(3, 752 + 12, 480)
    left = np.ascontiguousarray(wide[:, :752])
    right = np.ascontiguousarray(wide[:, 12:])
    params = (1000, 1.2, 8, 20, 7)
    ol, orr = oracle.Extractor(*params), oracle.Extractor(*params)
    _, _, kl, dl = ol.extract(left, (0, 0))
    _, _, kr, dr = orr.extract(right, (0, 0))
    mbf, mb = np.float32(47.9), np.float32(0.11)
    ur, dp, sad = oracle.stereo_match(ol, orr, kl, dl, kr, dr, mbf, mb)
    ok = dp > 0
    assert ok.sum() > 300
    disp = kl["x"][ok] - ur[ok]
    scale = (np.float32(1.2) ** kl["octave"][ok]).astype(np.float32)
    assert (np.abs(disp - 12) <= 1.01 * scale).all()
    assert np.array_equal(dp[ok], (mbf / disp).astype(np.float32))
    assert ((ur == -1) == (dp == -1)).all()
    assert (sad[ok] >= 0).all()
    removed = (sad >= 0) & ~ok
    if removed.any():
        assert sad[removed].min() >= sad[ok].max()


EUROC_CAM = (458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)


def _distort(cam, xn, yn):
    """Forward radial-tangential model in float64 (what undistortPoints inverts)."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = cam
    r2 = xn * xn + yn * yn
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn * rad + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    return xd * fx + cx, yd * fy + cy


def test_undistort_points_inverts_the_distortion_model(oracle):
    """Distort a lattice of ideal pixels with the forward model, undistort with the restated
    cv::undistortPoints: the ideal pixels come back within the residual of 5 iterations."""
    fx, fy, cx, cy = EUROC_CAM[:4]
    gx, gy = np.meshgrid(np.linspace(20, 730, 15), np.linspace(20, 460, 11))
    ideal = np.stack([gx.ravel(), gy.ravel()], 1)
    ud, vd = _distort(EUROC_CAM, (ideal[:, 0] - cx) / fx, (ideal[:, 1] - cy) / fy)
    back = oracle.undistort_points(EUROC_CAM, np.stack([ud, vd], 1).astype(np.float32))
    err = np.abs(back - ideal).max(axis=1)
    assert err.max() < 0.5                       # five iterations leave a few tenths at the corners
    centre = np.hypot(ideal[:, 0] - cx, ideal[:, 1] - cy) < 150
    assert err[centre].max() < 0.01
    # zero coefficients: identity up to float rounding
    same = oracle.undistort_points(EUROC_CAM[:4] + (0, 0, 0, 0, 0), ideal.astype(np.float32))
    assert np.abs(same - ideal).max() < 1e-3


def test_image_bounds_and_grid_invariants(oracle):
    b = oracle.image_bounds(EUROC_CAM, 752, 480)
    assert b[0] < 0 and b[2] < 0 and b[1] > 752 and b[3] > 480      # barrel distortion widens the frame
    assert np.array_equal(oracle.image_bounds(None, 752, 480), np.array([0, 752, 0, 480], np.float32))
    rng = np.random.default_rng(4)
    k = np.zeros(1500, oracle.KP)
    k["x"] = rng.uniform(0, 752, 1500).astype(np.float32)
    k["y"] = rng.uniform(0, 480, 1500).astype(np.float32)
    k["octave"] = rng.integers(0, 8, 1500)
    un, cs, ent = oracle.frame_grid(EUROC_CAM, b, k)
    assert cs[0] == 0 and (np.diff(cs) >= 0).all() and cs[-1] == len(ent) <= 1500
    assert len(set(ent.tolist())) == len(ent)
    for c in np.nonzero(np.diff(cs) > 1)[0][:50]:
        assert (np.diff(ent[cs[c]:cs[c + 1]]) > 0).all()           # push_back order
    inv_w, inv_h = np.float32(64) / (b[1] - b[0]), np.float32(48) / (b[3] - b[2])
    for c in np.nonzero(np.diff(cs) > 0)[0][:200]:
        i = ent[cs[c]]
        px = int(np.floor((un["x"][i] - b[0]) * inv_w + np.float32(0.5)))
        assert abs(px - c // 48) <= 1
    # a window query returns exactly the brute-force set
    n, out = oracle.features_in_area(b, un, cs, ent, 300.0, 200.0, 40.0, 1, 5, cap=256)
    brute = [i for i in range(1500) if abs(un["x"][i] - 300) < 40 and abs(un["y"][i] - 200) < 40
             and 1 <= un["octave"][i] <= 5 and i in set(ent.tolist())]
    assert n == len(brute) and sorted(out[:n].tolist()) == brute
