"""GPU parity of Frame::ComputeStereoFromRGBD (core/src/Frame.cc:998-1016) against the CPU oracle:
mvuRight / mvDepth identical floats (one IEEE division and one subtraction on both sides).

Collected after the other GPU tests on purpose: this kernel is among the newest code."""
import numpy as np
import pytest

import orb_slam3_amd as orbx
from orb_slam3_amd import grid

pytestmark = pytest.mark.gpu

EUROC = (458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)


def depth_images(rng, b, h, stride):
    d = rng.uniform(0.3, 12.0, (b, h, stride)).astype(np.float32)
    d[rng.random(d.shape) < 0.2] = 0.0             # holes of the sensor
    d[rng.random(d.shape) < 0.02] = -1.0
    return d


@pytest.mark.parametrize("cam,stride", [(EUROC, 752), (None, 768)])
def test_stereo_from_rgbd_matches_oracle(oracle, cam, stride):
    import torch
    from tests.conftest import TORCH_DEVICE
    B, w, h = 3, 752, 480
    rng = np.random.default_rng(31)
    frames = [orbx.textured_frame(80 + i, w, h) for i in range(B)]
    frames[1] = np.zeros_like(frames[1])                          # a frame without keypoints
    ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    res = ex.extract_batch(frames, (0, 0))
    pc = grid.pinhole(*cam) if cam is not None else None
    bounds = grid.image_bounds(pc, w, h)
    grid.frame_grid_batch(ex, B, pc, bounds)
    depth = depth_images(rng, B, h, stride)
    d_depth = torch.from_numpy(depth).to(TORCH_DEVICE)
    mbf = 40.0
    orbx.compute_stereo_from_rgbd_batch(ex, B, d_depth.data_ptr(), stride, mbf)
    total = 0
    for f in range(B):
        ur, dp, sad = orbx.stereo_download(ex, f, with_l1=True)
        k = res[f][1]
        un, _, _ = oracle.frame_grid(cam, bounds, k)
        rur, rdp = oracle.stereo_from_rgbd(k, un, depth[f], mbf)
        assert len(ur) == len(k)
        assert np.array_equal(ur.view(np.uint32), rur.view(np.uint32)), f
        assert np.array_equal(dp.view(np.uint32), rdp.view(np.uint32)), f
        assert (sad == -1).all()
        total += int((dp > 0).sum())
        assert ((dp > 0) == (ur != -1)).all() or (ur[dp > 0] == -1).sum() < 3   # uRight = -1 only by coincidence
    assert total > 1000
    # without the grid call (no undistorted keypoints for this extraction) the call is refused
    ex.extract_batch(frames, (0, 0))
    with pytest.raises(orbx.OrbxError):
        orbx.compute_stereo_from_rgbd_batch(ex, B, d_depth.data_ptr(), stride, mbf)
    with pytest.raises(orbx.OrbxError):
        orbx.stereo_download(ex, 0)
