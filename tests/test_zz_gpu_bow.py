"""GPU parity of the bag-of-words transform (Frame::ComputeBoW, core/src/Frame.cc:764-769, on the
reference's vocabulary text format, core/lib/ORB/vocabulary.cc:96-160) against the CPU oracle:
word ids, node ids, feature lists identical; word values identical doubles (the device adds and
normalises in the reference's order).

Collected after the other GPU tests on purpose: these kernels are the newest code."""
import numpy as np
import pytest

import orb_slam3_amd as orbx
from orb_slam3_amd import bow
from tests import voc_util

pytestmark = pytest.mark.gpu


def check_frame(ex, f, ovoc, desc, levelsup):
    wi, wv, ni, ns, ft = bow.download(ex, f)
    if desc is None:
        assert len(wi) == 0 and len(ni) == 0 and ns.tolist() == [0]
        return 0
    rwi, rwv, rni, rns, rft = ovoc.transform(desc, levelsup)
    assert np.array_equal(wi, rwi) and np.array_equal(ni, rni)
    assert np.array_equal(wv.view(np.uint64), rwv.view(np.uint64))      # identical doubles
    assert np.array_equal(ns, rns) and np.array_equal(ft, rft)
    return len(wi)


@pytest.mark.parametrize("k,L,ragged,levelsup,weighting", [(10, 4, False, 2, bow.TF_IDF), (8, 3, True, 1, bow.TF_IDF),
                                                           (17, 2, True, 4, bow.TF_IDF), (10, 3, False, 2, bow.TF)])
def test_bow_batch_matches_oracle(oracle, tmp_path, k, L, ragged, levelsup, weighting):
    arrays = voc_util.make_vocabulary(k, L, seed=100 + k + L, ragged=ragged)
    path = tmp_path / "voc.txt"
    voc_util.write_text(path, k, L, arrays, weighting=weighting)
    voc = bow.ORBVocabulary(path)
    ovoc = oracle.Vocabulary(path=path)
    assert voc.info() == ovoc.info()
    B = 4
    frames = [orbx.textured_frame(300 + i, 752, 480) for i in range(B)]
    frames[1] = np.zeros_like(frames[1])                       # a frame without keypoints
    ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    res = ex.extract_batch(frames, (0, 0))
    bow.transform_batch(ex, voc, B, levelsup)
    words = 0
    for f in range(B):
        words += check_frame(ex, f, ovoc, res[f][2], levelsup)
    assert words > 500
    # a second call on the same handle (buffers reused), fewer frames
    bow.transform_batch(ex, voc, 2, levelsup)
    check_frame(ex, 0, ovoc, res[0][2], levelsup)


def test_bow_vocabulary_from_arrays_and_repeated_words(oracle):
    """A tiny vocabulary (9 words): every word collects dozens of features, so the repeated
    additions and the normalisation are exercised on long runs."""
    k, L = 3, 2
    arrays = voc_util.make_vocabulary(k, L, seed=5, stop_fraction=0.3)
    voc = bow.ORBVocabulary.from_arrays(k, L, *arrays)
    ovoc = oracle.Vocabulary(arrays=(k, L) + arrays + (0,))
    ex = orbx.ORBextractor(2000, 1.2, 8, 20, 7)
    _, kps, desc = ex(orbx.textured_frame(9, 752, 480), None, (0, 0))
    bow.transform_batch(ex, voc, 1, 4)
    n = check_frame(ex, 0, ovoc, desc, 4)
    assert 1 <= n <= 9


def test_bow_large_capacity(oracle):
    """10000 features per frame: the sort keys no longer fit LDS and live in a device workspace."""
    k, L = 10, 3
    arrays = voc_util.make_vocabulary(k, L, seed=77)
    voc = bow.ORBVocabulary.from_arrays(k, L, *arrays)
    ovoc = oracle.Vocabulary(arrays=(k, L) + arrays + (0,))
    B = 2
    ex = orbx.ORBextractor(10000, 1.2, 8, 20, 7, max_batch=B)
    assert ex.max_keypoints() > 8192
    frames = [orbx.textured_frame(500 + i, 1241, 376) for i in range(B)]
    res = ex.extract_batch(frames, (0, 0))
    bow.transform_batch(ex, voc, B, 2)
    for f in range(B):
        assert len(res[f][2]) > 4000
        assert check_frame(ex, f, ovoc, res[f][2], 2) > 300


def test_bow_argument_errors(tmp_path):
    from orb_slam3_amd._lib import OrbxError, ORBX_ERR_BAD_ARG, ORBX_ERR_UNSUPPORTED, ORBX_ERR_BATCH
    arrays = voc_util.make_vocabulary(4, 2, seed=1)
    with pytest.raises(OrbxError) as e:      # IDF weighting is not what transform() implements here
        bow.ORBVocabulary.from_arrays(4, 2, *arrays, weighting=2)
    assert e.value.status == ORBX_ERR_UNSUPPORTED
    bad = (arrays[0].copy(),) + arrays[1:]
    bad[0][3] = 50                            # names a node that comes later in the file
    with pytest.raises(OrbxError) as e:
        bow.ORBVocabulary.from_arrays(4, 2, *bad)
    assert e.value.status == ORBX_ERR_BAD_ARG
    with pytest.raises(OrbxError):
        bow.ORBVocabulary(tmp_path / "missing.txt")
    # a damaged line in the middle of a text file fails the load (it is not taken for the end of
    # a shorter vocabulary); a file with or without a trailing newline loads all its nodes
    good = tmp_path / "good.txt"
    for nl in (True, False):
        voc_util.write_text(good, 4, 2, arrays, trailing_newline=nl)
        assert bow.ORBVocabulary(good).info()[2] == len(arrays[0]) + 1
    lines = good.read_text().split("\n")
    lines[len(lines) // 2] = "3 x" + lines[len(lines) // 2][3:]
    (tmp_path / "damaged.txt").write_text("\n".join(lines))
    with pytest.raises(OrbxError) as e:
        bow.ORBVocabulary(tmp_path / "damaged.txt")
    assert e.value.status == ORBX_ERR_BAD_ARG
    voc = bow.ORBVocabulary.from_arrays(4, 2, *arrays)
    ex = orbx.ORBextractor(300, 1.2, 4, 20, 7)
    ex(orbx.textured_frame(1, 376, 240), None, (0, 0))
    with pytest.raises(OrbxError) as e:
        bow.transform_batch(ex, voc, 2, 4)    # only one frame was extracted
    assert e.value.status == ORBX_ERR_BATCH


@pytest.mark.needs_real_device
def test_cv_adaptor_bow_binary(tmp_path):
    """orbx_cv::ORBVocabulary / ComputeBoW of include/orbx_cv.hpp fill DBoW2-shaped maps with exactly
    what the C ABI returns.  The binary is built by __graft_entry__.build() against the test-only
    OpenCV stand-in in tests/cpp/cvstub and links the real library."""
    import os
    import subprocess
    arrays = voc_util.make_vocabulary(10, 3, seed=3)
    path = tmp_path / "voc.txt"
    voc_util.write_text(path, 10, 3, arrays)
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "_build", "test_adaptor_bow")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build_adaptor_test()
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
