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
