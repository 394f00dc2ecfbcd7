"""GPU parity of the device-resident form of the fisheye pair's descriptor search
(Frame::ComputeStereoFishEyeMatches, core/src/Frame.cc:1157-1185: 2-NN of the left lapping-area
descriptors against the right ones) against the CPU oracle's 2-NN on the downloaded subsets.

Collected after the other GPU tests on purpose: this entry point is among the newest code."""
import numpy as np
import pytest

import orb_slam3_amd as orbx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lap_l,lap_r,shift", [((100, 511), (0, 400), 9), ((0, 511), (0, 511), 3), ((300, 320), (250, 270), 40)])
def test_lapping_knn_matches_oracle(oracle, lap_l, lap_r, shift):
    B = 2
    frames_l = [orbx.textured_frame(31 + i, 512, 512) for i in range(B)]
    frames_r = [np.roll(f, shift, axis=1) for f in frames_l]
    exl = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_batch=B)
    exr = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_batch=B)
    rl = exl.extract_batch(frames_l, lap_l)
    rr = exr.extract_batch(frames_r, lap_r)
    for f in range(B):
        ml, kl, dl = rl[f]
        mr, kr, dr = rr[f]
        idx, dist, gml, gmr = orbx.lapping_knn_match2(exl, exr, f)
        assert (gml, gmr) == (ml, mr)
        ridx, rdist = oracle.knn2(dl[ml:], dr[mr:])
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), f
        assert len(idx) == len(kl) - ml
    assert len(idx) > 10


def test_lapping_knn_edge_cases():
    from orb_slam3_amd._lib import OrbxError
    img = orbx.textured_frame(5, 376, 240)
    exl = orbx.ORBextractor(300, 1.2, 4, 20, 7)
    exr = orbx.ORBextractor(300, 1.2, 4, 20, 7)
    exl(img, None, (1000, 2000))                      # no keypoint in the lapping area: nothing to match
    exr(img, None, (0, 1000))
    idx, dist, ml, mr = orbx.lapping_knn_match2(exl, exr, 0)
    assert len(idx) == 0 and mr == 0
    exl(img, None, (0, 1000))
    exr(np.zeros_like(img), None, (0, 1000))          # empty right frame: no neighbours
    idx, dist, ml, mr = orbx.lapping_knn_match2(exl, exr, 0)
    assert len(idx) > 100 and (idx == 0xFFFFFFFF).all() and (dist == 0xFFFF).all()
    with pytest.raises(OrbxError):
        orbx.lapping_knn_match2(exl, exl, 0)
    with pytest.raises(OrbxError):
        orbx.lapping_knn_match2(exl, exr, 1)
