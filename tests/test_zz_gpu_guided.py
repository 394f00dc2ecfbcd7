"""GPU parity of the guided matchers (ORBmatcher::SearchByProjection(CurrentFrame, LastFrame) and
SearchForInitialization, core/lib/ORB/matcher.cc:1591-1808, 629-733) against the CPU oracle's
sequential loops: assignments, cleared entries, match counts and the updated vbPrevMatched are
identical.  The device runs rounds of independent decisions; the number of rounds is reported
and bounded here.

Collected after the other GPU tests on purpose: these kernels are the newest code."""
import numpy as np
import pytest

import orb_slam3_amd as orbx
from orb_slam3_amd import grid, guided

pytestmark = pytest.mark.gpu

PARAMS = (1000, 1.2, 8, 20, 7)


def two_frames(w=752, h=480, shift=6, params=PARAMS, seed=41):
    """Frames a and b with b(x) = a(x - shift); b is extracted on the device with its grid."""
    wide = orbx.textured_frame(seed, w + shift, h)
    a = np.ascontiguousarray(wide[:, shift:])
    b = np.ascontiguousarray(wide[:, :w])
    exa = orbx.ORBextractor(*params)
    _, ka, da = exa(a, None, (0, 0))
    exb = orbx.ORBextractor(*params)
    _, kb, db = exb(b, None, (0, 0))
    bounds = grid.image_bounds(None, w, h)
    grid.frame_grid_batch(exb, 1, None, bounds)
    un, cs, ent = grid.frame_grid_download(exb, 0)
    return exb, ka, da, un, db, bounds, cs, ent


def projection_inputs(rng, ka, kb, shift=6, jitter=2.0):
    n_last = len(ka)
    uv = np.stack([ka["x"] + shift + rng.uniform(-jitter, jitter, n_last),
                   ka["y"] + rng.uniform(-jitter, jitter, n_last)], 1).astype(np.float32)
    uv[::17] += 700                                            # some projections leave the image
    invz = rng.uniform(0.05, 1.0, n_last).astype(np.float32)
    invz[::23] = -0.1                                          # behind the camera
    valid = (rng.random(n_last) < 0.9).astype(np.uint8)
    obs = (rng.random(n_last) < 0.7).astype(np.uint8)          # temporal points have no observations
    state = rng.choice([0, 0, 0, 1, 2], len(kb)).astype(np.uint8)
    u_right = np.where(rng.random(len(kb)) < 0.5, kb["x"] - rng.uniform(1, 30, len(kb)), -1).astype(np.float32)
    return uv, invz, valid, obs, state, u_right


@pytest.mark.parametrize("forward,backward,stereo,th", [(0, 0, False, 15.0), (0, 0, True, 7.0), (1, 0, True, 15.0),
                                                        (0, 1, False, 15.0), (0, 0, False, 30.0)])
def test_search_by_projection_last_matches_oracle(oracle, forward, backward, stereo, th):
    exb, ka, da, kb, db, bounds, cs, ent = two_frames()
    rng = np.random.default_rng(5)
    uv, invz, valid, obs, state, u_right = projection_inputs(rng, ka, kb)
    ur = u_right if stereo else None
    sf = exb.GetScaleFactors()
    for ori in (True, False):
        want_n, want = oracle.search_by_projection_last(valid, uv, invz, ka["octave"], ka["angle"], da, obs, kb, db,
                                                        ur, state, bounds, cs, ent, np.array(sf, np.float32), th,
                                                        forward, backward, 40.0, ori)
        n, asg, rounds = guided.search_by_projection_last(exb, 0, valid, uv, invz, ka["octave"], ka["angle"], da,
                                                          obs, ur, state, th, forward, backward, 40.0, ori)
        assert n == want_n and np.array_equal(asg, want), (forward, backward, stereo, ori)
        assert 1 <= rounds <= 16, rounds
        assert (asg >= 0).sum() > 100


def test_search_by_projection_contention(oracle):
    """Many queries land on the same keypoints (large radius, all map points with observations,
    then none): long dependency chains, more rounds, same result as the greedy loop."""
    exb, ka, da, kb, db, bounds, cs, ent = two_frames(376, 240, params=(400, 1.2, 4, 20, 7))
    rng = np.random.default_rng(8)
    n_last = len(ka)
    uv = np.stack([ka["x"] + 6 + rng.uniform(-6, 6, n_last), ka["y"] + rng.uniform(-6, 6, n_last)], 1).astype(np.float32)
    invz = np.full(n_last, 0.5, np.float32)
    valid = np.ones(n_last, np.uint8)
    sf = np.array(exb.GetScaleFactors(), np.float32)
    for obs_p in (1.0, 0.5, 0.0):
        obs = (rng.random(n_last) < obs_p).astype(np.uint8)
        want_n, want = oracle.search_by_projection_last(valid, uv, invz, ka["octave"], ka["angle"], da, obs, kb, db,
                                                        None, np.zeros(len(kb), np.uint8), bounds, cs, ent, sf, 60.0,
                                                        0, 0, 40.0, True)
        n, asg, rounds = guided.search_by_projection_last(exb, 0, valid, uv, invz, ka["octave"], ka["angle"], da,
                                                          obs, None, None, 60.0, 0, 0, 40.0, True)
        assert n == want_n and np.array_equal(asg, want), obs_p
        assert rounds <= n_last + 2


def rig_frames(w=512, h=512, shift_l=6, shift_r=-9, params=PARAMS, seed=47):
    """A last frame a and the current frame's two cameras, each a shifted view of the same
    texture: left(x) = a(x - shift_l), right(x) = a(x - shift_r)."""
    pad = 16
    wide = orbx.textured_frame(seed, w + 2 * pad, h)
    a = np.ascontiguousarray(wide[:, pad:pad + w])
    exa = orbx.ORBextractor(*params)
    _, ka, da = exa(a, None, (0, 0))
    bounds = grid.image_bounds(None, w, h)
    cams = []
    for sh in (shift_l, shift_r):
        img = np.ascontiguousarray(wide[:, pad - sh:pad - sh + w])
        ex = orbx.ORBextractor(*params)
        _, k, d = ex(img, None, (0, 0))
        grid.frame_grid_batch(ex, 1, None, bounds)
        un, cs, ent = grid.frame_grid_download(ex, 0)
        cams.append((ex, un, d, cs, ent))
    return ka, da, cams, bounds


@pytest.mark.parametrize("forward,backward,th,jitter", [(0, 0, 15.0, 2.0), (1, 0, 15.0, 2.0), (0, 1, 7.0, 2.0),
                                                         (0, 0, 40.0, 6.0)])
def test_search_by_projection_last_rig_matches_oracle(oracle, forward, backward, th, jitter):
    """Two-camera current frame (CurrentFrame.Nleft != -1): the right camera's search is gated by
    the left one's, both feed one rotation histogram."""
    ka, da, cams, bounds = rig_frames()
    (exl, kl, dl, csl, el), (exr, kr, dr, csr, er) = cams
    rng = np.random.default_rng(11)
    n_last = len(ka)
    uvl = np.stack([ka["x"] + 6 + rng.uniform(-jitter, jitter, n_last),
                    ka["y"] + rng.uniform(-jitter, jitter, n_last)], 1).astype(np.float32)
    uvr = np.stack([ka["x"] - 9 + rng.uniform(-jitter, jitter, n_last),
                    ka["y"] + rng.uniform(-jitter, jitter, n_last)], 1).astype(np.float32)
    uvl[::17] += 700                     # left projection outside: the right camera is not searched either
    uvr[::13] -= 600                     # right projection outside the image: not checked, its window is just empty
    invz = rng.uniform(0.05, 1.0, n_last).astype(np.float32)
    invz[::23] = -0.1
    valid = (rng.random(n_last) < 0.9).astype(np.uint8)
    obs = (rng.random(n_last) < 0.7).astype(np.uint8)
    sl = rng.choice([0, 0, 0, 1, 2], len(kl)).astype(np.uint8)
    sr = rng.choice([0, 0, 0, 1, 2], len(kr)).astype(np.uint8)
    sf = np.array(exl.GetScaleFactors(), np.float32)
    for ori in (True, False):
        want_n, want_l, want_r = oracle.search_by_projection_last_rig(
            valid, uvl, invz, uvr, ka["octave"], ka["angle"], da, obs, (kl, dl, sl, csl, el), (kr, dr, sr, csr, er),
            bounds, sf, th, forward, backward, ori)
        n, al, ar, rounds = guided.search_by_projection_last_rig(exl, exr, 0, valid, uvl, invz, uvr, ka["octave"],
                                                                 ka["angle"], da, obs, sl, sr, th, forward, backward,
                                                                 ori)
        assert n == want_n and np.array_equal(al, want_l) and np.array_equal(ar, want_r), (forward, backward, ori)
        assert 1 <= rounds <= 24, rounds
        assert (al >= 0).sum() > 100 and (ar >= 0).sum() > 100
    # without states, and with an empty right camera window everywhere
    want_n, want_l, want_r = oracle.search_by_projection_last_rig(
        valid, uvl, invz, uvr - 5000, ka["octave"], ka["angle"], da, obs, (kl, dl, sl * 0, csl, el),
        (kr, dr, sr * 0, csr, er), bounds, sf, th, forward, backward, True)
    n, al, ar, rounds = guided.search_by_projection_last_rig(exl, exr, 0, valid, uvl, invz, uvr - 5000, ka["octave"],
                                                             ka["angle"], da, obs, None, None, th, forward, backward)
    assert n == want_n and np.array_equal(al, want_l) and np.array_equal(ar, want_r) and (ar == -1).all()


def stereo_partners(kl, kr, disparity, rng, random_share=0.0):
    """mvLeftToRightMatch / mvRightToLeftMatch stand-ins: mutual links between keypoints of the two
    cameras that sit `disparity` apart on the same row and level; a share of extra one-way links
    to arbitrary keypoints exercises writes that the reference's code allows but its stereo
    matcher would not produce."""
    l2r = -np.ones(len(kl), np.int32)
    r2l = -np.ones(len(kr), np.int32)
    for i in range(len(kl)):
        d = np.abs(kr["x"] - (kl["x"][i] - disparity)) + np.abs(kr["y"] - kl["y"][i]) + 100.0 * (kr["octave"] != kl["octave"][i])
        j = int(np.argmin(d))
        if d[j] <= 1.5 and r2l[j] < 0:
            l2r[i], r2l[j] = j, i
    extra_l = rng.random(len(kl)) < random_share
    l2r[extra_l] = rng.integers(0, len(kr), extra_l.sum())
    extra_r = rng.random(len(kr)) < random_share
    r2l[extra_r] = rng.integers(0, len(kl), extra_r.sum())
    return l2r, r2l


@pytest.mark.parametrize("th,nnratio,random_share,obs_p", [(1.0, 0.8, 0.0, 0.7), (3.0, 0.8, 0.0, 0.5),
                                                           (4.0, 0.95, 0.3, 0.5), (6.0, 1.0, 0.5, 0.3)])
def test_search_by_projection_points_rig_matches_oracle(oracle, th, nnratio, random_share, obs_p):
    """Two-camera frame (F.Nleft != -1): matches propagate to stereo partners, which can re-open a
    closed keypoint; a failed ratio test on the left skips the right camera."""
    ka, da, cams, bounds = rig_frames()
    (exl, kl, dl, csl, el), (exr, kr, dr, csr, er) = cams
    rng = np.random.default_rng(21)
    l2r, r2l = stereo_partners(kl, kr, 15.0, rng, random_share)
    assert (l2r >= 0).sum() > 100
    # every last-frame keypoint twice, as two map points with slightly different projections
    rep = np.concatenate([np.arange(len(ka)), np.arange(len(ka))])
    rng.shuffle(rep)
    n_pts = len(rep)
    jit = 1.5
    proj = np.stack([ka["x"][rep] + 6 + rng.uniform(-jit, jit, n_pts), ka["y"][rep] + rng.uniform(-jit, jit, n_pts)],
                    1).astype(np.float32)
    proj_r = np.stack([ka["x"][rep] - 9 + rng.uniform(-jit, jit, n_pts), ka["y"][rep] + rng.uniform(-jit, jit, n_pts)],
                      1).astype(np.float32)
    level = ka["octave"][rep].astype(np.int32)
    level_r = np.where(rng.random(n_pts) < 0.1, -1, np.clip(level + rng.integers(-1, 2, n_pts), 0, 7)).astype(np.int32)
    view = (rng.random(n_pts) < 0.85).astype(np.uint8)
    view_r = (rng.random(n_pts) < 0.85).astype(np.uint8)
    cos = rng.choice([0.9, 0.9985, 0.9999], n_pts).astype(np.float32)
    cos_r = rng.choice([0.9, 0.9985, 0.9999], n_pts).astype(np.float32)
    desc = da[rep].copy()
    desc[:, 3] ^= rng.integers(0, 8, n_pts).astype(np.uint8)
    obs = (rng.random(n_pts) < obs_p).astype(np.uint8)
    sl = rng.choice([0, 0, 0, 1, 2], len(kl)).astype(np.uint8)
    sr = rng.choice([0, 0, 0, 1, 2], len(kr)).astype(np.uint8)
    sf = np.array(exl.GetScaleFactors(), np.float32)
    want_n, want_l, want_r = oracle.search_by_projection_points_rig(
        view, proj, level, cos, view_r, proj_r, level_r, cos_r, desc, obs, (kl, dl, sl, csl, el),
        (kr, dr, sr, csr, er), l2r, r2l, bounds, sf, th, nnratio)
    n, al, ar, rounds = guided.search_by_projection_points_rig(exl, exr, 0, view, proj, level, cos, view_r, proj_r,
                                                               level_r, cos_r, desc, obs, l2r, r2l, sl, sr, th, nnratio)
    assert n == want_n and np.array_equal(al, want_l) and np.array_equal(ar, want_r), (th, nnratio, random_share)
    assert 1 <= rounds <= 40, rounds
    assert (al >= 0).sum() > 100 and (ar >= 0).sum() > 100
    # one camera's view only, and no partners at all
    none_l, none_r = -np.ones(len(kl), np.int32), -np.ones(len(kr), np.int32)
    for v, vr in ((view, view_r * 0), (view * 0, view_r)):
        want_n, want_l, want_r = oracle.search_by_projection_points_rig(
            v, proj, level, cos, vr, proj_r, level_r, cos_r, desc, obs, (kl, dl, sl * 0, csl, el),
            (kr, dr, sr * 0, csr, er), none_l, none_r, bounds, sf, th, nnratio)
        n, al, ar, rounds = guided.search_by_projection_points_rig(exl, exr, 0, v, proj, level, cos, vr, proj_r,
                                                                   level_r, cos_r, desc, obs, none_l, none_r, None,
                                                                   None, th, nnratio)
        assert n == want_n and np.array_equal(al, want_l) and np.array_equal(ar, want_r)


def test_search_by_projection_empty_inputs(oracle):
    exb, ka, da, kb, db, bounds, cs, ent = two_frames(376, 240, params=(300, 1.2, 4, 20, 7))
    n, asg, rounds = guided.search_by_projection_last(exb, 0, np.zeros(0, np.uint8), np.zeros((0, 2), np.float32),
                                                      np.zeros(0), np.zeros(0), np.zeros(0), np.zeros((0, 32)),
                                                      np.zeros(0))
    assert n == 0 and len(asg) == len(kb) and (asg == -1).all()
    # nothing valid
    m = len(ka)
    n, asg, rounds = guided.search_by_projection_last(exb, 0, np.zeros(m, np.uint8), np.zeros((m, 2), np.float32),
                                                      np.ones(m), ka["octave"], ka["angle"], da, np.ones(m))
    assert n == 0 and (asg == -1).all() and rounds == 1


@pytest.mark.parametrize("window,ori", [(100, True), (100, False), (30, True), (250, True)])
def test_search_for_initialization_matches_oracle(oracle, window, ori):
    exb, ka, da, kb, db, bounds, cs, ent = two_frames(seed=43, params=(5000, 1.2, 8, 20, 7))
    prev = np.stack([ka["x"], ka["y"]], 1)
    want_n, want, want_pm = oracle.search_for_initialization(ka, da, kb, db, bounds, cs, ent, prev, window, 0.9, ori)
    n, m12, pm, rounds = guided.search_for_initialization(exb, 0, ka, da, prev, window, 0.9, ori)
    assert n == want_n and np.array_equal(m12, want), (window, ori)
    assert np.array_equal(pm.view(np.uint32), want_pm.view(np.uint32))
    assert 1 <= rounds <= 16, rounds
    if window >= 100:
        ok = m12 >= 0
        assert ok.sum() > 100
        assert (np.abs((kb["x"][m12[ok]] - ka["x"][ok]) - 6) <= 2).mean() > 0.9   # the shift is found


def test_search_for_initialization_repeated_descriptors(oracle):
    """Few distinct descriptors: distance ties and keypoints claimed several times over."""
    exb, ka, da, kb, db, bounds, cs, ent = two_frames(376, 240, params=(600, 1.2, 2, 20, 7))
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (12, 32), dtype=np.uint8)
    da2 = base[rng.integers(0, 12, len(ka))]
    da2[:, 0] ^= rng.integers(0, 4, len(ka)).astype(np.uint8)
    # the current frame's descriptors live on the device: match frame 1's synthetic
    # descriptors against them anyway, and against close copies of them
    pick = rng.integers(0, len(kb), len(ka))
    da3 = db[pick].copy()
    da3[:, 5] ^= rng.integers(0, 16, len(ka)).astype(np.uint8)
    prev = np.stack([kb["x"][pick] + rng.uniform(-20, 20, len(ka)), kb["y"][pick] + rng.uniform(-20, 20, len(ka))],
                    1).astype(np.float32)
    ka0 = ka.copy()
    ka0["octave"] = 0
    for d1 in (da2, da3):
        want_n, want, want_pm = oracle.search_for_initialization(ka0, d1, kb, db, bounds, cs, ent, prev, 60, 0.95, True)
        n, m12, pm, rounds = guided.search_for_initialization(exb, 0, ka0, d1, prev, 60, 0.95, True)
        assert n == want_n and np.array_equal(m12, want)
        assert np.array_equal(pm.view(np.uint32), want_pm.view(np.uint32))


def test_guided_argument_errors():
    from orb_slam3_amd._lib import OrbxError, ORBX_ERR_BAD_ARG
    ex = orbx.ORBextractor(300, 1.2, 4, 20, 7)
    ex(orbx.textured_frame(1, 376, 240), None, (0, 0))
    with pytest.raises(OrbxError) as e:     # no grid yet
        guided.search_by_projection_last(ex, 0, np.ones(1, np.uint8), np.zeros((1, 2)), np.ones(1), np.zeros(1),
                                         np.zeros(1), np.zeros((1, 32)), np.ones(1))
    assert e.value.status == ORBX_ERR_BAD_ARG
    grid.frame_grid_batch(ex, 1, None, grid.image_bounds(None, 376, 240))
    with pytest.raises(OrbxError) as e:     # octave outside the pyramid
        guided.search_by_projection_last(ex, 0, np.ones(1, np.uint8), np.zeros((1, 2)), np.ones(1),
                                         np.array([9]), np.zeros(1), np.zeros((1, 32)), np.ones(1))
    assert e.value.status == ORBX_ERR_BAD_ARG


def test_rig_argument_errors():
    from orb_slam3_amd._lib import OrbxError, ORBX_ERR_BAD_ARG
    ka, da, cams, bounds = rig_frames(376, 240, params=(300, 1.2, 4, 20, 7))
    (exl, kl, dl, csl, el), (exr, kr, dr, csr, er) = cams
    m = len(ka)
    last = (np.ones(m, np.uint8), np.zeros((m, 2), np.float32), np.ones(m, np.float32), np.zeros((m, 2), np.float32),
            ka["octave"], ka["angle"], da, np.ones(m, np.uint8))
    with pytest.raises(OrbxError) as e:      # one handle for both cameras
        guided.search_by_projection_last_rig(exl, exl, 0, *last)
    assert e.value.status == ORBX_ERR_BAD_ARG
    other = orbx.ORBextractor(300, 1.2, 4, 20, 7)
    other(orbx.textured_frame(3, 376, 240), None, (0, 0))
    grid.frame_grid_batch(other, 1, None, grid.image_bounds(None, 376, 200))
    with pytest.raises(OrbxError) as e:      # the two cameras' grids disagree on the image bounds
        guided.search_by_projection_last_rig(exl, other, 0, *last)
    assert e.value.status == ORBX_ERR_BAD_ARG
    pts = (np.ones(m, np.uint8), np.zeros((m, 2)), ka["octave"], np.ones(m), np.ones(m, np.uint8), np.zeros((m, 2)),
           ka["octave"], np.ones(m), da, np.ones(m, np.uint8))
    l2r, r2l = -np.ones(len(kl), np.int32), -np.ones(len(kr), np.int32)
    l2r[0] = len(kr)                         # partner index outside the right camera
    with pytest.raises(OrbxError) as e:
        guided.search_by_projection_points_rig(exl, exr, 0, *pts, l2r, r2l)
    assert e.value.status == ORBX_ERR_BAD_ARG
    l2r[0] = -1
    lvl_r = ka["octave"].copy()
    lvl_r[0] = 9                             # right level outside the pyramid (-1 alone means none)
    with pytest.raises(OrbxError) as e:
        guided.search_by_projection_points_rig(exl, exr, 0, *pts[:6], lvl_r, *pts[7:], l2r, r2l)
    assert e.value.status == ORBX_ERR_BAD_ARG
    lvl_r[0] = -1
    n, al, ar, _ = guided.search_by_projection_points_rig(exl, exr, 0, *pts[:6], lvl_r, *pts[7:], l2r, r2l)
    assert n >= 0 and len(al) == len(kl) and len(ar) == len(kr)


def test_results_derived_from_an_extraction_expire_with_it():
    """A new extraction on the handle invalidates the frame grid (and with it the guided matchers)
    until the grid call has run again."""
    from orb_slam3_amd._lib import OrbxError, ORBX_ERR_BAD_ARG
    ex = orbx.ORBextractor(300, 1.2, 4, 20, 7)
    img = orbx.textured_frame(2, 376, 240)
    ex(img, None, (0, 0))
    b = grid.image_bounds(None, 376, 240)
    grid.frame_grid_batch(ex, 1, None, b)
    grid.features_in_area(ex, 0, np.array([[100, 100, 20, -1, -1]], np.float32))
    ex(img, None, (0, 0))
    with pytest.raises(OrbxError) as e:
        grid.features_in_area(ex, 0, np.array([[100, 100, 20, -1, -1]], np.float32))
    assert e.value.status == ORBX_ERR_BAD_ARG
    grid.frame_grid_batch(ex, 1, None, b)
    grid.features_in_area(ex, 0, np.array([[100, 100, 20, -1, -1]], np.float32))
