"""GPU parity of the local-map matcher, ORBmatcher::SearchByProjection(F, vpMapPoints, th, ...)
(core/lib/ORB/matcher.cc:41-142, monocular / rectified case), and of the relocalisation matcher,
SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:1810-1923), against the CPU
oracle's sequential loops: assignments and match counts identical.

Collected after the other GPU tests on purpose: this kernel is the newest code."""
import numpy as np
import pytest

import orb_slam3_amd as orbx
from orb_slam3_amd import guided
from tests.test_zz_gpu_guided import two_frames

pytestmark = pytest.mark.gpu


def map_points(rng, ka, da, kb, nlevels, shift=6, jitter=3.0, copies=1):
    """Map points seen from frame a, projected into frame b; `copies` > 1 repeats every point with
    its own jitter, so that several map points compete for each keypoint."""
    ka = np.concatenate([ka] * copies)
    da = np.concatenate([da] * copies)
    n = len(ka)
    proj = np.stack([ka["x"] + shift + rng.uniform(-jitter, jitter, n), ka["y"] + rng.uniform(-jitter, jitter, n)],
                    1).astype(np.float32)
    proj[::19] += 900                                          # some land outside the grid
    proj_xr = (proj[:, 0] - rng.uniform(1, 30, n)).astype(np.float32)
    level = np.clip(ka["octave"] + rng.integers(-1, 2, n), 0, nlevels - 1).astype(np.int32)
    view_cos = np.where(rng.random(n) < 0.5, 0.9995, rng.uniform(0.5, 0.998, n)).astype(np.float32)
    in_view = (rng.random(n) < 0.9).astype(np.uint8)
    desc = da.copy()
    flip = rng.random(n) < 0.5
    desc[flip, int(rng.integers(0, 32))] ^= 0x5A
    obs = (rng.random(n) < 0.8).astype(np.uint8)
    state = rng.choice([0, 0, 0, 1, 2], len(kb)).astype(np.uint8)
    u_right = np.where(rng.random(len(kb)) < 0.5, kb["x"] - rng.uniform(1, 30, len(kb)), -1).astype(np.float32)
    return in_view, proj, proj_xr, level, view_cos, desc, obs, state, u_right


@pytest.mark.parametrize("stereo,th,ratio", [(False, 1.0, 0.8), (True, 3.0, 0.8), (True, 1.0, 0.6), (False, 5.0, 0.95),
                                             (False, 15.0, 1.0)])
def test_search_by_projection_points_matches_oracle(oracle, stereo, th, ratio):
    exb, ka, da, kb, db, bounds, cs, ent = two_frames()
    rng = np.random.default_rng(12)
    in_view, proj, proj_xr, level, view_cos, desc, obs, state, u_right = map_points(rng, ka, da, kb, 8)
    ur = u_right if stereo else None
    sf = np.array(exb.GetScaleFactors(), np.float32)
    want_n, want = oracle.search_by_projection_points(in_view, proj, proj_xr, level, view_cos, desc, obs, kb, db, ur,
                                                      state, bounds, cs, ent, sf, th, ratio)
    n, asg, rounds = guided.search_by_projection_points(exb, 0, in_view, proj, proj_xr, level, view_cos, desc, obs,
                                                        ur, state, th, ratio)
    assert n == want_n and np.array_equal(asg, want), (stereo, th, ratio)
    assert 1 <= rounds <= 16, rounds
    assert (asg >= 0).sum() > 100


def test_search_by_projection_points_contention(oracle):
    """Three map points per keypoint, wide windows: long chains of 'the keypoint I wanted is
    taken', with and without observations (without, later points overwrite earlier ones)."""
    exb, ka, da, kb, db, bounds, cs, ent = two_frames(376, 240, params=(400, 1.2, 4, 20, 7))
    rng = np.random.default_rng(13)
    in_view, proj, proj_xr, level, view_cos, desc, obs, state, u_right = map_points(rng, ka, da, kb, 4, jitter=6.0,
                                                                                    copies=3)
    sf = np.array(exb.GetScaleFactors(), np.float32)
    for obs_p in (1.0, 0.5, 0.0):
        o = (rng.random(len(obs)) < obs_p).astype(np.uint8)
        want_n, want = oracle.search_by_projection_points(in_view, proj, proj_xr, level, view_cos, desc, o, kb, db,
                                                          None, np.zeros(len(kb), np.uint8), bounds, cs, ent, sf,
                                                          10.0, 0.9)
        n, asg, rounds = guided.search_by_projection_points(exb, 0, in_view, proj, proj_xr, level, view_cos, desc, o,
                                                            None, None, 10.0, 0.9)
        assert n == want_n and np.array_equal(asg, want), obs_p
        assert rounds <= len(in_view) + 2
        if obs_p == 0.0:
            assert n > (asg >= 0).sum()          # overwritten assignments still counted (matcher.cc:137)


def test_search_by_projection_points_empty_and_errors(oracle):
    from orb_slam3_amd._lib import OrbxError, ORBX_ERR_BAD_ARG
    exb, ka, da, kb, db, bounds, cs, ent = two_frames(376, 240, params=(300, 1.2, 4, 20, 7))
    z = np.zeros(0)
    n, asg, rounds = guided.search_by_projection_points(exb, 0, np.zeros(0, np.uint8), np.zeros((0, 2)), z, z, z,
                                                        np.zeros((0, 32)), z)
    assert n == 0 and len(asg) == len(kb) and (asg == -1).all()
    m = len(ka)
    n, asg, rounds = guided.search_by_projection_points(exb, 0, np.zeros(m, np.uint8), np.zeros((m, 2)), np.zeros(m),
                                                        ka["octave"], np.ones(m), da, np.ones(m))
    assert n == 0 and (asg == -1).all() and rounds == 1
    with pytest.raises(OrbxError) as e:       # a level the extractor does not have
        guided.search_by_projection_points(exb, 0, np.ones(1, np.uint8), np.zeros((1, 2)), np.zeros(1), [4], [1.0],
                                           np.zeros((1, 32)), [1])
    assert e.value.status == ORBX_ERR_BAD_ARG
    with pytest.raises(OrbxError):            # no such frame
        guided.search_by_projection_points(exb, 1, np.ones(1, np.uint8), np.zeros((1, 2)), np.zeros(1), [0], [1.0],
                                           np.zeros((1, 32)), [1])


@pytest.mark.parametrize("th,orb_dist,ori,taken_p", [(10.0, 100, True, 0.3), (3.0, 64, True, 0.0), (10.0, 64, False, 0.3),
                                                     (25.0, 255, True, 0.1), (10.0, 0, True, 0.0)])
def test_search_by_projection_keyframe_matches_oracle(oracle, th, orb_dist, ori, taken_p):
    exb, ka, da, kb, db, bounds, cs, ent = two_frames()
    rng = np.random.default_rng(21)
    n = len(ka)
    uv = np.stack([ka["x"] + 6 + rng.uniform(-3, 3, n), ka["y"] + rng.uniform(-3, 3, n)], 1).astype(np.float32)
    uv[::17] += 700
    level = np.clip(ka["octave"] + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
    valid = (rng.random(n) < 0.85).astype(np.uint8)
    desc = da.copy()
    flip = rng.random(n) < 0.5
    desc[flip, 7] ^= 0x3C
    has_point = (rng.random(len(kb)) < taken_p).astype(np.uint8)
    sf = np.array(exb.GetScaleFactors(), np.float32)
    want_n, want = oracle.search_by_projection_keyframe(valid, uv, level, ka["angle"], desc, kb, db, has_point, bounds,
                                                        cs, ent, sf, th, orb_dist, ori)
    nm, asg, rounds = guided.search_by_projection_keyframe(exb, 0, valid, uv, level, ka["angle"], desc,
                                                           has_point if taken_p else None, th, orb_dist, ori)
    assert nm == want_n and np.array_equal(asg, want), (th, orb_dist, ori)
    assert 1 <= rounds <= 24, rounds
    if orb_dist >= 64:
        assert (asg >= 0).sum() > 100
        assert not (has_point.astype(bool) & (asg != -1)).any()      # keypoints with a map point stay untouched
    with pytest.raises(orbx.OrbxError):
        guided.search_by_projection_keyframe(exb, 0, valid, uv, level, ka["angle"], desc, None, th, 256, ori)
