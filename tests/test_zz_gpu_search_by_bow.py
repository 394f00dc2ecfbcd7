"""GPU parity of ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (core/lib/ORB/matcher.cc:
222-405, monocular / rectified case) against the CPU oracle: matches, cleared entries and the
return value identical.  The frame's feature vector is the one the device transform left in
device memory; the keyframe's comes from the oracle's transform of the other image.

Collected after the other GPU tests on purpose: these kernels are the newest code."""
import numpy as np
import pytest

import orb_slam3_amd as orbx
from orb_slam3_amd import bow, guided
from tests import voc_util

pytestmark = pytest.mark.gpu


def pair(w=752, h=480, shift=6, params=(1000, 1.2, 8, 20, 7), seed=51):
    wide = orbx.textured_frame(seed, w + shift, h)
    a = np.ascontiguousarray(wide[:, shift:])
    b = np.ascontiguousarray(wide[:, :w])
    exa = orbx.ORBextractor(*params)
    _, ka, da = exa(a, None, (0, 0))
    exb = orbx.ORBextractor(*params)
    _, kb, db = exb(b, None, (0, 0))
    return exb, ka, da, kb, db


@pytest.mark.parametrize("k,L,levelsup,ratio,ori", [(10, 4, 2, 0.7, True), (10, 3, 4, 0.7, True), (6, 3, 1, 0.9, False),
                                                    (3, 2, 1, 0.75, True), (17, 2, 0, 1.0, True)])
def test_search_by_bow_matches_oracle(oracle, k, L, levelsup, ratio, ori):
    exb, ka, da, kb, db = pair()
    arrays = voc_util.make_vocabulary(k, L, seed=7 * k + L)
    voc = bow.ORBVocabulary.from_arrays(k, L, *arrays)
    ovoc = oracle.Vocabulary(arrays=(k, L) + arrays + (0,))
    bow.transform_batch(exb, voc, 1, levelsup)
    _, _, an, as_, af = ovoc.transform(da, levelsup)            # keyframe = frame a
    _, _, bn, bs, bf = ovoc.transform(db, levelsup)
    rng = np.random.default_rng(k + L)
    valid = (rng.random(len(ka)) < 0.85).astype(np.uint8)
    want_n, want = oracle.search_by_bow(valid, ka["angle"], da, an, as_, af, kb["angle"], db, bn, bs, bf, ratio, ori)
    n, m = guided.search_by_bow(exb, 0, valid, ka["angle"], da, an, as_, af, ratio, ori)
    assert n == want_n and np.array_equal(m, want), (k, L, levelsup)
    ok = m >= 0
    assert ok.sum() > 50
    assert (np.abs((kb["x"][ok] - ka["x"][m[ok]]) - 6) <= 2).mean() > 0.8       # the shift is found
    if ori:
        assert (m == -2).sum() >= 0 and n == ok.sum()


def test_search_by_bow_one_node_holds_everything(oracle):
    """levelsup >= L: every feature sits in node 0, so one wave walks lists of a thousand entries
    (positions beyond one pass of 64 lanes) and every keyframe feature competes with all others."""
    exb, ka, da, kb, db = pair(376, 240, params=(600, 1.2, 4, 20, 7))
    k, L = 5, 2
    arrays = voc_util.make_vocabulary(k, L, seed=3)
    voc = bow.ORBVocabulary.from_arrays(k, L, *arrays)
    ovoc = oracle.Vocabulary(arrays=(k, L) + arrays + (0,))
    bow.transform_batch(exb, voc, 1, 4)
    _, _, an, as_, af = ovoc.transform(da, 4)
    _, _, bn, bs, bf = ovoc.transform(db, 4)
    assert len(an) == 1 and len(bf) > 300
    valid = np.ones(len(ka), np.uint8)
    for ratio in (0.7, 1.0):
        want_n, want = oracle.search_by_bow(valid, ka["angle"], da, an, as_, af, kb["angle"], db, bn, bs, bf, ratio, True)
        n, m = guided.search_by_bow(exb, 0, valid, ka["angle"], da, an, as_, af, ratio, True)
        assert n == want_n and np.array_equal(m, want), ratio
    # duplicates in the keyframe: the second copy of a feature finds its keypoint taken
    da2 = np.concatenate([da, da])
    ka2 = np.concatenate([ka, ka])
    _, _, an2, as2, af2 = ovoc.transform(da2, 4)
    valid2 = np.ones(len(ka2), np.uint8)
    want_n, want = oracle.search_by_bow(valid2, ka2["angle"], da2, an2, as2, af2, kb["angle"], db, bn, bs, bf, 0.9, False)
    n, m = guided.search_by_bow(exb, 0, valid2, ka2["angle"], da2, an2, as2, af2, 0.9, False)
    assert n == want_n and np.array_equal(m, want)


def test_search_by_bow_empty_and_errors(oracle):
    from orb_slam3_amd._lib import OrbxError, ORBX_ERR_BAD_ARG
    exb, ka, da, kb, db = pair(376, 240, params=(300, 1.2, 4, 20, 7))
    k, L = 6, 2
    arrays = voc_util.make_vocabulary(k, L, seed=4)
    voc = bow.ORBVocabulary.from_arrays(k, L, *arrays)
    z32, zi = np.zeros(0, np.uint32), np.zeros(1, np.int32)
    with pytest.raises(OrbxError) as e:          # the frame has no bag of words yet
        guided.search_by_bow(exb, 0, np.zeros(0, np.uint8), np.zeros(0), np.zeros((0, 32)), z32, zi, z32)
    assert e.value.status == ORBX_ERR_BAD_ARG
    bow.transform_batch(exb, voc, 1, 1)
    n, m = guided.search_by_bow(exb, 0, np.zeros(0, np.uint8), np.zeros(0), np.zeros((0, 32)), z32, zi, z32)
    assert n == 0 and len(m) == len(kb) and (m == -1).all()
    ovoc = oracle.Vocabulary(arrays=(k, L) + arrays + (0,))
    _, _, an, as_, af = ovoc.transform(da, 1)
    n, m = guided.search_by_bow(exb, 0, np.zeros(len(ka), np.uint8), ka["angle"], da, an, as_, af)   # no map points
    assert n == 0 and (m == -1).all()
    bad = af.copy()
    bad[0] = len(ka)                               # a feature index beyond the keyframe's keypoints
    with pytest.raises(OrbxError):
        guided.search_by_bow(exb, 0, np.ones(len(ka), np.uint8), ka["angle"], da, an, as_, bad)
    with pytest.raises(OrbxError):                 # node ids must ascend
        guided.search_by_bow(exb, 0, np.ones(len(ka), np.uint8), ka["angle"], da, an[::-1].copy(), as_, af)
    exb(np.zeros((240, 376), np.uint8), None, (0, 0))    # a new extraction invalidates the bag of words
    with pytest.raises(OrbxError):
        guided.search_by_bow(exb, 0, np.ones(len(ka), np.uint8), ka["angle"], da, an, as_, af)
