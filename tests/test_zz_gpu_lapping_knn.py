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


def test_lapping_knn_batch_equals_per_pair():
    """One launch for the whole batch (results on the device) against the blocking per-pair call,
    frames with different subset sizes, one of them with next to no lapping keypoint on the left and
    one with an empty right frame."""
    B = 5
    frames_l = [orbx.textured_frame(61 + i, 512, 512) for i in range(B)]
    frames_r = [np.roll(f, 7, axis=1) for f in frames_l]
    frames_l[2] = frames_l[2].copy()
    frames_l[2][:, 200:] = 90                       # no texture inside the left lapping area
    frames_r[3] = np.zeros_like(frames_r[3])         # nothing on the right
    exl = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_batch=B)
    exr = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_batch=B)
    exl.extract_batch(frames_l, (200, 511))
    exr.extract_batch(frames_r, (0, 350))
    per_pair = [orbx.lapping_knn_match2(exl, exr, f) for f in range(B)]
    orbx.lapping_knn_match2_batch(exl, exr, B)
    sizes = []
    for f in range(B):
        idx, dist, ml, mr = orbx.lapping_knn_download(exl, f)
        ridx, rdist, rml, rmr = per_pair[f]
        assert (ml, mr) == (rml, rmr)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), f
        sizes.append(len(idx))
    assert sizes[2] < 20 and sizes[0] > 100 and len(set(sizes)) > 2
    assert (per_pair[3][0] == 0xFFFFFFFF).all()
    from orb_slam3_amd._lib import OrbxError
    with pytest.raises(OrbxError):
        orbx.lapping_knn_download(exl, B)            # beyond the batch call
    exl.extract_batch(frames_l, (200, 511))          # a new extraction invalidates the results
    with pytest.raises(OrbxError):
        orbx.lapping_knn_download(exl, 0)


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
