"""The C++ oracle against an independent numpy restatement of the same published algorithms
(tests/numpy_spec.py) on small images, stage by stage: pyramid, blur, per-cell FAST candidates,
moments and angles, descriptors.  Two restatements written separately agreeing bit for bit is
what stands in for the missing reference fixtures on the CPU side."""
import numpy as np
import pytest

from tests import numpy_spec as S

PARAMS = (300, 1.2, 4, 20, 7)


@pytest.fixture(scope="module")
def case(oracle):
    import orb_slam3_amd as orbx
    img = orbx.textured_frame(17, 214, 160)
    e = oracle.Extractor(*PARAMS)
    st, mono, k, d = e.extract(img, (0, 0))
    assert st == 0 and len(k) > 100
    levels, padded = S.build_pyramid(img, PARAMS[2], PARAMS[1])
    return dict(img=img, e=e, k=k, d=d, levels=levels, padded=padded)


def test_tables_and_level_sizes(oracle, case):
    scale, inv = S.level_tables(PARAMS[2], PARAMS[1])
    t = case["e"].tables()
    assert np.array_equal(np.array(scale, np.float32), t["scale"])
    assert np.array_equal(np.array(inv, np.float32), t["inv"])
    assert S.umax_table() == t["umax"].tolist()
    for l in range(PARAMS[2]):
        w, h, _ = case["e"].level_dims(l)
        assert (w, h) == S.level_size(214, 160, inv[l])


def test_pyramid_bytes(case):
    for l in range(PARAMS[2]):
        w, h, _ = case["e"].level_dims(l)
        assert np.array_equal(case["e"].padded(l)[:, :w + 38], case["padded"][l]), l


def test_blur_bytes(case):
    for l in range(PARAMS[2]):
        b = case["e"].blurred(l)
        if b is not None:
            assert np.array_equal(b, S.gaussian7(case["levels"][l])), l


def test_resize_and_blur_on_odd_sizes(oracle):
    rng = np.random.default_rng(2)
    for sw, sh, dw, dh in [(67, 50, 56, 42), (101, 77, 84, 64), (40, 33, 20, 17), (35, 35, 34, 34)]:
        src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        assert np.array_equal(oracle.resize(src, dw, dh), S.resize_linear_u8(src, dw, dh))
        assert np.array_equal(oracle.gauss7(src), S.gaussian7(src))


def test_fast_on_random_cells(oracle):
    rng = np.random.default_rng(3)
    import orb_slam3_amd as orbx
    img = orbx.textured_frame(5, 300, 200)
    for i in range(25):
        cw, ch = rng.integers(20, 60, 2)
        x0, y0 = rng.integers(0, 300 - cw), rng.integers(0, 200 - ch)
        roi = np.ascontiguousarray(img[y0:y0 + ch, x0:x0 + cw])
        for th in (7, 20):
            got = [tuple(r) for r in oracle.fast(roi, th).tolist()]
            assert got == S.fast_cell(roi, th), (i, th)
    noise = rng.integers(0, 256, (40, 45), dtype=np.uint8)     # dense corners, many score ties
    assert [tuple(r) for r in oracle.fast(noise, 7).tolist()] == S.fast_cell(noise, 7)


def test_cell_candidates(case):
    for l in range(PARAMS[2]):
        w, h, _ = case["e"].level_dims(l)
        want = S.level_candidates(case["padded"][l], w, h, PARAMS[3], PARAMS[4])
        got = [tuple(r) for r in case["e"].candidates(l).tolist()]
        assert got == want, l
        assert len(got) > 20


def test_moments_angles_and_descriptors(case):
    pattern = S.load_pattern()
    scale, _ = S.level_tables(PARAMS[2], PARAMS[1])
    out = 0
    for l in range(PARAMS[2]):
        kl, mom = case["e"].keys(l)
        blurred = case["e"].blurred(l)
        # the reflected border is part of what the descriptor may sample near the image edge
        bl_pad = np.pad(S.gaussian7(case["levels"][l]), 19, mode="reflect") if blurred is not None else None
        for i in range(len(kl)):
            x, y = int(kl["x"][i]), int(kl["y"][i])
            m10, m01 = S.ic_moments(case["padded"][l], x, y)
            assert (m10, m01) == tuple(mom[i]), (l, i)
            ang = S.fast_atan2(m01, m10)
            assert ang == kl["angle"][i], (l, i)
            # final keypoint and descriptor (vLappingArea (0,0): output order = level order)
            k = case["k"][out]
            assert k["octave"] == l and k["angle"] == ang
            sx = np.float32(x) * scale[l] if l else np.float32(x)
            assert k["x"] == sx
            want = S.brief_descriptor(bl_pad[19:, 19:], x, y, ang, pattern)
            assert np.array_equal(case["d"][out], want), (l, i)
            out += 1
    assert out == len(case["k"])


def test_stereo_association(oracle):
    import orb_slam3_amd as orbx
    params = (400, 1.2, 5, 20, 7)
    left = orbx.textured_frame(23, 320, 240)
    right = orbx.right_view(left, 23, dmin=2.0, dmax=20.0, bands=4)
    el, er = oracle.Extractor(*params), oracle.Extractor(*params)
    _, _, kl, dl = el.extract(left, (0, 0))
    _, _, kr, dr = er.extract(right, (0, 0))
    mbf, mb = np.float32(30.0), np.float32(0.11)
    ur, dp, _ = oracle.stereo_match(el, er, kl, dl, kr, dr, mbf, mb)
    scale, inv = S.level_tables(params[2], params[1])
    lv_l, _ = S.build_pyramid(left, params[2], params[1])
    lv_r, _ = S.build_pyramid(right, params[2], params[1])
    sur, sdp = S.stereo_matches(lv_l, lv_r, scale, inv, kl, dl, kr, dr, mbf, mb)
    assert (dp > 0).sum() > 40
    assert np.array_equal(ur.view(np.uint32), sur.view(np.uint32))
    assert np.array_equal(dp.view(np.uint32), sdp.view(np.uint32))


def test_undistortion_grid_and_window_queries(oracle):
    cam = (458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
    rng = np.random.default_rng(8)
    k = np.zeros(600, oracle.KP)
    k["x"] = rng.uniform(0, 752, 600).astype(np.float32)
    k["y"] = rng.uniform(0, 480, 600).astype(np.float32)
    k["octave"] = rng.integers(0, 8, 600)
    xy = np.stack([k["x"], k["y"]], 1)
    assert np.array_equal(oracle.undistort_points(cam, xy).view(np.uint32), S.undistort_points(cam, xy).view(np.uint32))
    b = oracle.image_bounds(cam, 752, 480)
    un, cs, ent = oracle.frame_grid(cam, b, k)
    grid = S.frame_grid(un, b)
    for c in range(64 * 48):
        assert ent[cs[c]:cs[c + 1]].tolist() == grid.get(c, []), c
    for _ in range(150):
        x, y = rng.uniform(-50, 800), rng.uniform(-50, 530)
        r = float(rng.choice([8.0, 25.0, 70.0]))
        lo, hi = int(rng.choice([-1, 0, 2])), int(rng.choice([-1, 3, 7]))
        n, out = oracle.features_in_area(b, un, cs, ent, x, y, r, lo, hi, cap=600)
        assert out[:n].tolist() == S.features_in_area(un, grid, b, x, y, r, lo, hi)


def _two_frames(oracle, shift=6):
    import orb_slam3_amd as orbx
    params = (500, 1.2, 5, 20, 7)
    wide = orbx.textured_frame(41, 400 + shift, 300)
    a = np.ascontiguousarray(wide[:, shift:])
    b = np.ascontiguousarray(wide[:, :400])           # b(x) = a(x - shift): content moves right
    ea, eb = oracle.Extractor(*params), oracle.Extractor(*params)
    _, _, ka, da = ea.extract(a, (0, 0))
    _, _, kb, db = eb.extract(b, (0, 0))
    bounds = oracle.image_bounds(None, 400, 300)
    un_b, cs, ent = oracle.frame_grid(None, bounds, kb)
    return params, ka, da, un_b, db, bounds, cs, ent


def test_three_maxima(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        sizes = rng.integers(0, rng.choice([3, 12, 200]), 30)
        assert oracle.three_maxima(sizes) == S.three_maxima(sizes.tolist())
    assert oracle.three_maxima([0] * 30) == (-1, -1, -1)
    assert oracle.three_maxima([5] + [0] * 29) == (0, -1, -1)


def test_search_for_initialization(oracle):
    params, ka, da, kb, db, bounds, cs, ent = _two_frames(oracle)
    prev = np.stack([ka["x"], ka["y"]], 1)
    grid = S.frame_grid(kb, bounds)
    for ori in (True, False):
        n, m12, pm = oracle.search_for_initialization(ka, da, kb, db, bounds, cs, ent, prev, 100, 0.9, ori)
        sn, sm12, spm = S.search_for_initialization(ka, da, kb, db, grid, bounds, prev, 100, 0.9, ori)
        assert n == sn and m12.tolist() == sm12 and np.array_equal(pm, spm)
        assert n > 30
        ok = m12 >= 0
        assert (np.abs((kb["x"][m12[ok]] - ka["x"][ok]) - 6) <= 2).mean() > 0.9   # the shift is found


def test_search_by_projection_last(oracle):
    params, ka, da, kb, db, bounds, cs, ent = _two_frames(oracle)
    rng = np.random.default_rng(5)
    n_last = len(ka)
    scale, _ = S.level_tables(params[2], params[1])
    uv = np.stack([ka["x"] + 6 + rng.uniform(-2, 2, n_last), ka["y"] + rng.uniform(-2, 2, n_last)], 1).astype(np.float32)
    uv[::17] += 500                                            # some projections leave the image
    invz = rng.uniform(0.05, 1.0, n_last).astype(np.float32)
    invz[::23] = -0.1                                          # behind the camera
    valid = (rng.random(n_last) < 0.9).astype(np.uint8)
    obs = (rng.random(n_last) < 0.7).astype(np.uint8)          # temporal points have no observations
    state = rng.choice([0, 0, 0, 1, 2], len(kb)).astype(np.uint8)
    u_right = np.where(rng.random(len(kb)) < 0.5, kb["x"] - rng.uniform(1, 30, len(kb)), -1).astype(np.float32)
    grid = S.frame_grid(kb, bounds)
    for forward, backward, ur, th in [(0, 0, None, 15.0), (0, 0, u_right, 7.0), (1, 0, u_right, 15.0), (0, 1, None, 15.0)]:
        n, asg = oracle.search_by_projection_last(valid, uv, invz, ka["octave"], ka["angle"], da, obs, kb, db, ur,
                                                  state, bounds, cs, ent, np.array(scale, np.float32), th,
                                                  forward, backward, 40.0, True)
        sn, sasg = S.search_by_projection_last(valid, uv, invz, ka["octave"], ka["angle"], da, obs, kb, db, ur,
                                               state, grid, bounds, scale, th, forward, backward, 40.0, True)
        assert n == sn and asg.tolist() == sasg, (forward, backward)
        assert (asg >= 0).sum() > 20


def _map_points(rng, ka, da, kb, nlevels, shift=6, jitter=3.0):
    """Map points seen from the first frame, projected into the second (which is shifted)."""
    n = len(ka)
    proj = np.stack([ka["x"] + shift + rng.uniform(-jitter, jitter, n), ka["y"] + rng.uniform(-jitter, jitter, n)],
                    1).astype(np.float32)
    proj_xr = (proj[:, 0] - rng.uniform(1, 30, n)).astype(np.float32)
    level = np.clip(ka["octave"] + rng.integers(-1, 2, n), 0, nlevels - 1).astype(np.int32)
    view_cos = np.where(rng.random(n) < 0.5, 0.9995, rng.uniform(0.5, 0.998, n)).astype(np.float32)
    in_view = (rng.random(n) < 0.9).astype(np.uint8)
    desc = da.copy()
    flip = rng.random(n) < 0.5
    desc[flip, rng.integers(0, 32)] ^= 0x5A
    obs = (rng.random(n) < 0.8).astype(np.uint8)
    state = rng.choice([0, 0, 0, 1, 2], len(kb)).astype(np.uint8)
    u_right = np.where(rng.random(len(kb)) < 0.5, kb["x"] - rng.uniform(1, 30, len(kb)), -1).astype(np.float32)
    return in_view, proj, proj_xr, level, view_cos, desc, obs, state, u_right


def test_search_by_projection_points(oracle):
    params, ka, da, kb, db, bounds, cs, ent = _two_frames(oracle)
    rng = np.random.default_rng(9)
    scale, _ = S.level_tables(params[2], params[1])
    grid = S.frame_grid(kb, bounds)
    in_view, proj, proj_xr, level, view_cos, desc, obs, state, u_right = _map_points(rng, ka, da, kb, params[2])
    worst = 0
    for ur, th, ratio, obs_p in [(None, 1.0, 0.8, 1.0), (u_right, 3.0, 0.8, 1.0), (u_right, 1.0, 0.6, 0.5),
                                 (None, 5.0, 0.95, 0.0), (None, 15.0, 1.0, 0.7)]:
        o = (obs & (rng.random(len(obs)) < obs_p)).astype(np.uint8) if obs_p < 1 else np.ones_like(obs)
        n, asg = oracle.search_by_projection_points(in_view, proj, proj_xr, level, view_cos, desc, o, kb, db, ur, state,
                                                    bounds, cs, ent, np.array(scale, np.float32), th, ratio)
        sn, sasg = S.search_by_projection_points(in_view, proj, proj_xr, level, view_cos, desc, o, kb, db, ur, state,
                                                 grid, bounds, scale, th, ratio)
        assert n == sn and asg.tolist() == sasg, (th, ratio)
        rn, rasg, rounds = S.rounds_search_by_projection_points(in_view, proj, proj_xr, level, view_cos, desc, o, kb,
                                                                db, ur, state, grid, bounds, scale, th, ratio)
        assert (n, asg.tolist()) == (rn, rasg), (th, ratio)
        worst = max(worst, rounds)
        assert (asg >= 0).sum() > 20
    assert worst <= 10, worst


def _rig_cameras(oracle):
    """A last frame and the two cameras of the current one: shifted views of one texture."""
    import orb_slam3_amd as orbx
    params = (400, 1.2, 4, 20, 7)
    pad, w, h = 16, 320, 240
    wide = orbx.textured_frame(47, w + 2 * pad, h)
    ea = oracle.Extractor(*params)
    _, _, ka, da = ea.extract(np.ascontiguousarray(wide[:, pad:pad + w]), (0, 0))
    bounds = oracle.image_bounds(None, w, h)
    cams = []
    for sh in (6, -9):
        e = oracle.Extractor(*params)
        _, _, k, d = e.extract(np.ascontiguousarray(wide[:, pad - sh:pad - sh + w]), (0, 0))
        un, cs, ent = oracle.frame_grid(None, bounds, k)
        cams.append((un, d, cs, ent))
    return params, ka, da, cams, bounds


def test_search_by_projection_last_rig(oracle):
    params, ka, da, cams, bounds = _rig_cameras(oracle)
    (kl, dl, csl, el), (kr, dr, csr, er) = cams
    rng = np.random.default_rng(12)
    scale, _ = S.level_tables(params[2], params[1])
    n = len(ka)
    uvl = np.stack([ka["x"] + 6 + rng.uniform(-2, 2, n), ka["y"] + rng.uniform(-2, 2, n)], 1).astype(np.float32)
    uvr = np.stack([ka["x"] - 9 + rng.uniform(-2, 2, n), ka["y"] + rng.uniform(-2, 2, n)], 1).astype(np.float32)
    uvl[::17] += 500
    uvr[::13] -= 400
    invz = rng.uniform(0.05, 1.0, n).astype(np.float32)
    invz[::23] = -0.1
    valid = (rng.random(n) < 0.9).astype(np.uint8)
    obs = (rng.random(n) < 0.7).astype(np.uint8)
    sl = rng.choice([0, 0, 0, 1, 2], len(kl)).astype(np.uint8)
    sr = rng.choice([0, 0, 0, 1, 2], len(kr)).astype(np.uint8)
    gl, gr = S.frame_grid(kl, bounds), S.frame_grid(kr, bounds)
    for forward, backward, th, ori in [(0, 0, 15.0, True), (1, 0, 15.0, True), (0, 1, 7.0, False), (0, 0, 30.0, True)]:
        want = oracle.search_by_projection_last_rig(valid, uvl, invz, uvr, ka["octave"], ka["angle"], da, obs,
                                                    (kl, dl, sl, csl, el), (kr, dr, sr, csr, er), bounds,
                                                    np.array(scale, np.float32), th, forward, backward, ori)
        got = S.search_by_projection_last_rig(valid, uvl, invz, uvr, ka["octave"], ka["angle"], da, obs,
                                              (kl, dl, sl, gl), (kr, dr, sr, gr), bounds, scale, th, forward,
                                              backward, ori)
        assert (want[0], want[1].tolist(), want[2].tolist()) == got, (forward, backward, th)
        assert (want[1] >= 0).sum() > 20 and (want[2] >= 0).sum() > 20


def test_search_by_projection_points_rig(oracle):
    """Sequential oracle == sequential numpy == rounds form with per-keypoint write lists, with
    mutual partner links and with arbitrary one-way ones (which re-open closed keypoints)."""
    params, ka, da, cams, bounds = _rig_cameras(oracle)
    (kl, dl, csl, el), (kr, dr, csr, er) = cams
    rng = np.random.default_rng(13)
    scale, _ = S.level_tables(params[2], params[1])
    gl, gr = S.frame_grid(kl, bounds), S.frame_grid(kr, bounds)
    rep = np.concatenate([np.arange(len(ka)), np.arange(len(ka))])
    rng.shuffle(rep)
    n = len(rep)
    proj = np.stack([ka["x"][rep] + 6 + rng.uniform(-1.5, 1.5, n), ka["y"][rep] + rng.uniform(-1.5, 1.5, n)], 1).astype(np.float32)
    proj_r = np.stack([ka["x"][rep] - 9 + rng.uniform(-1.5, 1.5, n), ka["y"][rep] + rng.uniform(-1.5, 1.5, n)], 1).astype(np.float32)
    level = ka["octave"][rep].astype(np.int32)
    level_r = np.where(rng.random(n) < 0.1, -1, np.clip(level + rng.integers(-1, 2, n), 0, params[2] - 1)).astype(np.int32)
    view = (rng.random(n) < 0.85).astype(np.uint8)
    view_r = (rng.random(n) < 0.85).astype(np.uint8)
    cos = rng.choice([0.9, 0.9985, 0.9999], n).astype(np.float32)
    cos_r = rng.choice([0.9, 0.9985, 0.9999], n).astype(np.float32)
    desc = da[rep].copy()
    desc[:, 3] ^= rng.integers(0, 8, n).astype(np.uint8)
    sl = rng.choice([0, 0, 0, 1, 2], len(kl)).astype(np.uint8)
    sr = rng.choice([0, 0, 0, 1, 2], len(kr)).astype(np.uint8)
    worst = 0
    for th, ratio, share, obs_p in [(1.0, 0.8, 0.0, 0.7), (4.0, 0.95, 0.3, 0.5), (6.0, 1.0, 0.5, 0.3)]:
        obs = (rng.random(n) < obs_p).astype(np.uint8)
        l2r, r2l = -np.ones(len(kl), np.int32), -np.ones(len(kr), np.int32)
        for i in range(len(kl)):          # mutual links: 15 px apart on the same row and level
            dd = np.abs(kr["x"] - (kl["x"][i] - 15)) + np.abs(kr["y"] - kl["y"][i]) + 100.0 * (kr["octave"] != kl["octave"][i])
            j = int(np.argmin(dd))
            if dd[j] <= 1.5 and r2l[j] < 0:
                l2r[i], r2l[j] = j, i
        xl, xr = rng.random(len(kl)) < share, rng.random(len(kr)) < share
        l2r[xl] = rng.integers(0, len(kr), xl.sum())
        r2l[xr] = rng.integers(0, len(kl), xr.sum())
        want = oracle.search_by_projection_points_rig(view, proj, level, cos, view_r, proj_r, level_r, cos_r, desc, obs,
                                                      (kl, dl, sl, csl, el), (kr, dr, sr, csr, er), l2r, r2l, bounds,
                                                      np.array(scale, np.float32), th, ratio)
        want = (want[0], want[1].tolist(), want[2].tolist())
        pts = (view, proj, level, cos, view_r, proj_r, level_r, cos_r, desc)
        got = S.search_by_projection_points_rig(pts, obs, (kl, dl, sl, gl), (kr, dr, sr, gr), l2r, r2l, bounds, scale,
                                                th, ratio)
        assert want == got, (th, ratio, share)
        rn, ral, rar, rounds = S.rounds_search_by_projection_points_rig(pts, obs, (kl, dl, sl, gl), (kr, dr, sr, gr),
                                                                        l2r, r2l, bounds, scale, th, ratio)
        assert want == (rn, ral, rar), (th, ratio, share)
        worst = max(worst, rounds)
        assert (np.array(got[1]) >= 0).sum() > 20 and (np.array(got[2]) >= 0).sum() > 20
    assert worst <= 16, worst


def test_search_by_bow(oracle):
    from tests import voc_util
    params, ka, da, kb, db, bounds, cs, ent = _two_frames(oracle)
    rng = np.random.default_rng(10)
    for k, L, levelsup, ratio, ori in [(6, 3, 2, 0.7, True), (10, 2, 1, 0.9, True), (4, 3, 4, 0.75, False),
                                       (5, 4, 2, 1.0, True)]:
        arrays = voc_util.make_vocabulary(k, L, seed=k * 10 + L)
        voc = oracle.Vocabulary(arrays=(k, L) + arrays + (0,))
        _, _, an, as_, af = voc.transform(da, levelsup)       # the keyframe = frame a
        _, _, bn, bs, bf = voc.transform(db, levelsup)
        valid = (rng.random(len(ka)) < 0.85).astype(np.uint8)
        n, m = oracle.search_by_bow(valid, ka["angle"], da, an, as_, af, kb["angle"], db, bn, bs, bf, ratio, ori)
        kf_fv = {int(an[r]): af[as_[r]:as_[r + 1]].tolist() for r in range(len(an))}
        f_fv = {int(bn[r]): bf[bs[r]:bs[r + 1]].tolist() for r in range(len(bn))}
        sn, sm = S.search_by_bow(valid, ka["angle"], da, kf_fv, kb["angle"], db, f_fv, ratio, ori)
        assert n == sn and m.tolist() == sm, (k, L, levelsup)
        assert (m >= 0).sum() > 20
        ok = m >= 0
        assert (np.abs((kb["x"][ok] - ka["x"][m[ok]]) - 6) <= 2).mean() > 0.8     # the shift is found


def test_guided_matchers_as_rounds_equal_the_greedy_loops(oracle):
    """The data-parallel formulation planned for the device (rounds of independent decisions that
    only look at lower-indexed queries' claims) reproduces the reference's sequential loops, in a
    handful of rounds."""
    params, ka, da, kb, db, bounds, cs, ent = _two_frames(oracle)
    rng = np.random.default_rng(6)
    n_last = len(ka)
    scale, _ = S.level_tables(params[2], params[1])
    grid = S.frame_grid(kb, bounds)
    uv = np.stack([ka["x"] + 6 + rng.uniform(-3, 3, n_last), ka["y"] + rng.uniform(-3, 3, n_last)], 1).astype(np.float32)
    invz = rng.uniform(0.05, 1.0, n_last).astype(np.float32)
    valid = np.ones(n_last, np.uint8)
    state = rng.choice([0, 0, 0, 0, 1, 2], len(kb)).astype(np.uint8)
    worst = 0
    for obs_p, th in [(1.0, 15.0), (0.6, 25.0), (0.0, 15.0)]:
        obs = (rng.random(n_last) < obs_p).astype(np.uint8)
        n, asg = oracle.search_by_projection_last(valid, uv, invz, ka["octave"], ka["angle"], da, obs, kb, db, None,
                                                  state, bounds, cs, ent, np.array(scale, np.float32), th, 0, 0, 40.0, True)
        rn, rasg, rounds = S.rounds_search_by_projection_last(valid, uv, invz, ka["octave"], ka["angle"], da, obs, kb,
                                                              db, None, state, grid, bounds, scale, th, 0, 0, 40.0, True)
        assert (n, asg.tolist()) == (rn, rasg)
        worst = max(worst, rounds)
    prev = np.stack([ka["x"], ka["y"]], 1)
    for window in (30, 100):
        n, m12, _ = oracle.search_for_initialization(ka, da, kb, db, bounds, cs, ent, prev, window, 0.9, True)
        rn, rm12, rounds = S.rounds_search_for_initialization(ka, da, kb, db, grid, bounds, prev, window, 0.9, True)
        assert (n, m12.tolist()) == (rn, rm12)
        worst = max(worst, rounds)
    assert worst <= 8, worst


@pytest.mark.parametrize("w,h,params", [(180, 140, (200, 1.1, 3, 20, 7)), (231, 173, (300, 1.5, 3, 25, 5)),
                                        (256, 200, (250, 2.0, 2, 12, 12)), (199, 151, (150, 1.2, 4, 40, 10))])
def test_other_sizes_and_parameters(oracle, w, h, params):
    """Pyramid, blur and per-cell candidates for other scale factors, thresholds and odd sizes."""
    import orb_slam3_amd as orbx
    img = orbx.textured_frame(w + h, w, h)
    e = oracle.Extractor(*params)
    st, _, k, _ = e.extract(img, (0, 0))
    assert st == 0 and len(k) > 20
    levels, padded = S.build_pyramid(img, params[2], params[1])
    for l in range(params[2]):
        lw, lh, _ = e.level_dims(l)
        assert np.array_equal(e.padded(l)[:, :lw + 38], padded[l]), l
        b = e.blurred(l)
        if b is not None:
            assert np.array_equal(b, S.gaussian7(levels[l])), l
        want = S.level_candidates(padded[l], lw, lh, params[3], params[4])
        assert [tuple(r) for r in e.candidates(l).tolist()] == want, l
