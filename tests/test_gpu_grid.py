"""GPU parity of the frame grid (UndistortKeyPoints, AssignFeaturesToGrid, GetFeaturesInArea;
core/src/Frame.cc:432-459, 682-829) against the CPU oracle: undistorted coordinates are
identical floats (double arithmetic without contraction on both sides), cell lists identical
including their order, window queries identical including their order."""
import numpy as np
import pytest

import orb_slam3_amd as orbx
from orb_slam3_amd import grid

pytestmark = pytest.mark.gpu

EUROC = (458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
K3CAM = (435.2, 435.2, 367.2, 252.2, -0.31, 0.11, 0.0007, -0.0003, -0.02)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("cam", [EUROC, K3CAM, None, EUROC[:4] + (0, 0, 0, 0, 0)])
def test_host_undistort_and_bounds_match_oracle(oracle, cam):
    rng = np.random.default_rng(0)
    xy = rng.uniform(-50, 800, (3000, 2)).astype(np.float32)
    if cam is not None:
        got = grid.undistort_points(grid.pinhole(*cam), xy)
        assert np.array_equal(bits(got), bits(oracle.undistort_points(cam, xy)))
    b = grid.image_bounds(grid.pinhole(*cam) if cam is not None else None, 752, 480)
    assert np.array_equal(bits(b), bits(oracle.image_bounds(cam, 752, 480)))


@pytest.mark.parametrize("cam", [EUROC, K3CAM, None])
def test_frame_grid_batch_matches_oracle(oracle, cam):
    params = (1200, 1.2, 8, 20, 7)
    B = 5
    frames = [orbx.textured_frame(60 + i, 752, 480) for i in range(B)]
    frames[2] = np.zeros_like(frames[2])                 # a frame without keypoints
    ex = orbx.ORBextractor(*params, max_batch=B)
    res = ex.extract_batch(frames, (0, 0))
    pc = grid.pinhole(*cam) if cam is not None else None
    b = grid.image_bounds(pc, 752, 480)
    grid.frame_grid_batch(ex, B, pc, b)
    rng = np.random.default_rng(7)
    for f in range(B):
        un, cs, ent = grid.frame_grid_download(ex, f)
        run, rcs, rent = oracle.frame_grid(cam, b, res[f][1])
        assert len(un) == len(run)
        assert np.array_equal(un.view(np.uint8), run.view(np.uint8)), f
        assert np.array_equal(cs, rcs) and np.array_equal(ent, rent), f
        if len(un) == 0:
            continue
        # window queries: around keypoints (guided matching) and at random places, with and
        # without level limits, including windows that leave the grid
        q = np.zeros((300, 5), np.float32)
        pick = rng.integers(0, len(un), 200)
        q[:200, 0] = un["x"][pick] + rng.uniform(-3, 3, 200)
        q[:200, 1] = un["y"][pick] + rng.uniform(-3, 3, 200)
        q[200:, 0] = rng.uniform(-100, 900, 100)
        q[200:, 1] = rng.uniform(-100, 600, 100)
        q[:, 2] = rng.choice([5.0, 15.0, 40.0, 120.0], 300)
        q[:, 3] = rng.choice([-1, 0, 1, 3], 300)
        q[:, 4] = rng.choice([-1, 2, 4, 7], 300)
        cnt, idx = grid.features_in_area(ex, f, q, per_query_cap=48)
        for i in range(300):
            n, out = oracle.features_in_area(b, run, rcs, rent, float(q[i, 0]), float(q[i, 1]), float(q[i, 2]),
                                             int(q[i, 3]), int(q[i, 4]), cap=48)
            assert cnt[i] == n, (f, i)
            m = min(n, 48)
            assert np.array_equal(idx[i, :m], out[:m]), (f, i)
        assert cnt.max() > 48 or cnt.max() > 5


def test_frame_grid_argument_errors():
    from orb_slam3_amd._lib import OrbxError
    ex = orbx.ORBextractor(500, 1.2, 8, 20, 7)
    with pytest.raises(OrbxError):                      # nothing extracted yet
        grid.frame_grid_batch(ex, 1, None, [0, 752, 0, 480])
    ex(orbx.textured_frame(1, 320, 240), None, (0, 0))
    with pytest.raises(OrbxError):                      # empty bounds
        grid.frame_grid_batch(ex, 1, None, [0, 0, 0, 240])
    # distortion that folds all four corners onto the principal point: bounds without extent
    folded = grid.pinhole(155.4, 157.99, 132.5, 203.0, -0.336549, -0.053964, -0.000485, 0.000557, 0.01)
    b = grid.image_bounds(folded, 259, 410)
    assert not (b[1] > b[0] and b[3] > b[2])
    with pytest.raises(OrbxError):
        grid.frame_grid_batch(ex, 1, folded, b)
    with pytest.raises(OrbxError):                      # more frames than extracted
        grid.frame_grid_batch(ex, 2, None, [0, 320, 0, 240])
    with pytest.raises(OrbxError):                      # query before any grid call
        grid.features_in_area(ex, 0, np.zeros((1, 5), np.float32))
    grid.frame_grid_batch(ex, 1, None, [0, 320, 0, 240])
    cnt, _ = grid.features_in_area(ex, 0, np.array([[160, 120, 400, -1, -1]], np.float32), 8)
    assert cnt[0] == len(grid.frame_grid_download(ex, 0)[2])
