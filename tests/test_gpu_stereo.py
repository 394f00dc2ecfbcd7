"""GPU parity of the device-resident Frame::ComputeStereoMatches (core/src/Frame.cc:831-996)
against the CPU oracle: mvuRight / mvDepth must be identical floats (every operation is a single
IEEE float op on both sides, so the tolerance is zero), the L1 distances identical integers."""
import numpy as np
import pytest

import orb_slam3_amd as orbx
from tests.stereo_util import right_view

pytestmark = pytest.mark.gpu

MBF, MB = 435.2 * 0.11, 0.11   # EuRoC-like: fx * baseline, baseline -> maxD = fx


def oracle_stereo(oracle, left, right, params, lap=(0, 0)):
    ol, orr = oracle.Extractor(*params), oracle.Extractor(*params)
    _, _, kl, dl = ol.extract(left, lap)
    _, _, kr, dr = orr.extract(right, lap)
    return oracle.stereo_match(ol, orr, kl, dl, kr, dr, MBF, MB)


def same_floats(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("size,params,seed", [((752, 480), (1200, 1.2, 8, 20, 7), 5),
                                              ((1241, 376), (2000, 1.2, 8, 20, 7), 6),
                                              ((640, 480), (800, 1.5, 5, 20, 7), 7)])
def test_stereo_single_frame_matches_oracle(oracle, size, params, seed):
    w, h = size
    left = orbx.textured_frame(seed, w, h)
    right = right_view(left, seed)
    exl, exr = orbx.ORBextractor(*params), orbx.ORBextractor(*params)
    exl(left, None, (0, 0))
    exr(right, None, (0, 0))
    ur, dp = orbx.compute_stereo_matches(exl, exr, MBF, MB)
    rur, rdp, rsad = oracle_stereo(oracle, left, right, params)
    assert len(ur) == len(rur)
    assert same_floats(ur, rur) and same_floats(dp, rdp)
    ur2, dp2, sad = orbx.stereo_download(exl, 0, with_l1=True)
    assert same_floats(ur2, ur) and same_floats(dp2, dp)
    assert np.array_equal(sad, rsad)
    good = dp > 0
    assert good.sum() > 150                      # a shifted copy matches plentifully
    assert ((sad >= 0) & ~good).sum() > 0        # and the median rule removed some


def test_stereo_batch_matches_per_frame(oracle):
    params = (1200, 1.2, 8, 20, 7)
    B = 6
    lefts = [orbx.textured_frame(40 + i, 752, 480) for i in range(B)]
    rights = [right_view(l, 40 + i) for i, l in enumerate(lefts)]
    rights[3] = np.zeros_like(rights[3])         # a right image without keypoints
    lefts[4] = np.zeros_like(lefts[4])           # a left image without keypoints
    exl = orbx.ORBextractor(*params, max_batch=B)
    exr = orbx.ORBextractor(*params, max_batch=B)
    exl.extract_batch(lefts, (0, 0))
    exr.extract_batch(rights, (0, 0))
    orbx.compute_stereo_matches_batch(exl, exr, B, MBF, MB)
    for f in range(B):
        ur, dp, sad = orbx.stereo_download(exl, f, with_l1=True)
        rur, rdp, rsad = oracle_stereo(oracle, lefts[f], rights[f], params)
        assert same_floats(ur, rur) and same_floats(dp, rdp) and np.array_equal(sad, rsad), f
    assert len(orbx.stereo_download(exl, 4)[0]) == 0
    assert (orbx.stereo_download(exl, 3)[1] == -1).all()


def test_stereo_identical_images_median_zero(oracle):
    """Right == left: every L1 distance is 0, so the median rule's threshold is 0 and the
    `first < thDist` test (Frame.cc:989) fails for every match: all of them are erased."""
    params = (1000, 1.2, 8, 20, 7)
    img = orbx.textured_frame(77, 752, 480)
    exl, exr = orbx.ORBextractor(*params), orbx.ORBextractor(*params)
    exl(img, None, (0, 0))
    exr(img, None, (0, 0))
    ur, dp, sad = None, None, None
    orbx.compute_stereo_matches_batch(exl, exr, 1, MBF, MB)
    ur, dp, sad = orbx.stereo_download(exl, 0, with_l1=True)
    rur, rdp, rsad = oracle_stereo(oracle, img, img, params)
    assert same_floats(ur, rur) and same_floats(dp, rdp) and np.array_equal(sad, rsad)
    assert (sad == 0).sum() > 100 and (dp == -1).all()


def test_stereo_zero_shift_with_noise_clamps_disparity(oracle):
    """Right = left + small noise: disparities scatter around 0; negative ones are rejected,
    exact zeros are replaced by 0.01 (Frame.cc:972-976)."""
    params = (1000, 1.2, 8, 20, 7)
    img = orbx.textured_frame(78, 752, 480)
    rng = np.random.default_rng(1)
    right = np.clip(img.astype(np.int16) + rng.integers(-3, 4, img.shape), 0, 255).astype(np.uint8)
    exl, exr = orbx.ORBextractor(*params), orbx.ORBextractor(*params)
    _, kl, _ = exl(img, None, (0, 0))
    exr(right, None, (0, 0))
    ur, dp = orbx.compute_stereo_matches(exl, exr, MBF, MB)
    rur, rdp, _ = oracle_stereo(oracle, img, right, params)
    assert same_floats(ur, rur) and same_floats(dp, rdp)
    assert (dp > 0).sum() > 50
    clamped = dp == np.float32(MBF) / np.float32(0.01)
    assert same_floats(ur[clamped], (kl["x"][clamped].astype(np.float64) - 0.01).astype(np.float32))


def test_stereo_argument_errors():
    from orb_slam3_amd._lib import OrbxError, ORBX_ERR_BAD_ARG, ORBX_ERR_BATCH
    a = orbx.ORBextractor(500, 1.2, 8, 20, 7)
    b = orbx.ORBextractor(500, 1.2, 8, 20, 7)
    c = orbx.ORBextractor(500, 1.2, 8, 20, 7)
    img = orbx.textured_frame(1, 320, 240)
    a(img, None, (0, 0))
    with pytest.raises(OrbxError) as e:          # right handle has not extracted anything
        orbx.compute_stereo_matches(a, b, MBF, MB)
    assert e.value.status == ORBX_ERR_BAD_ARG
    b(img, None, (0, 0))
    c(orbx.textured_frame(1, 400, 240), None, (0, 0))
    with pytest.raises(OrbxError) as e:          # different image size
        orbx.compute_stereo_matches(a, c, MBF, MB)
    assert e.value.status == ORBX_ERR_BAD_ARG
    with pytest.raises(OrbxError) as e:          # same handle twice
        orbx.compute_stereo_matches(a, a, MBF, MB)
    assert e.value.status == ORBX_ERR_BAD_ARG
    with pytest.raises(OrbxError) as e:
        orbx.compute_stereo_matches_batch(a, b, 2, MBF, MB)
    assert e.value.status == ORBX_ERR_BATCH
    with pytest.raises(OrbxError):
        orbx.compute_stereo_matches(a, b, MBF, 0.0)
    orbx.compute_stereo_matches(a, b, MBF, MB)


@pytest.mark.needs_real_device
def test_cv_adaptor_binary():
    """include/orbx_cv.hpp (ORBextractor / DescriptorDistance / ComputeStereoMatches / KnnMatch2
    with OpenCV-style types) gives the same bytes as the C ABI it wraps.  The binary is built by
    __graft_entry__.build() against the test-only OpenCV stand-in in tests/cpp/cvstub."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "_build", "test_adaptor")
    if not os.path.exists(exe):       # normally prebuilt; a two-second host compile otherwise
        import __graft_entry__
        __graft_entry__.build_adaptor_test()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_left_and_right_extraction_from_two_threads(oracle):
    """The stereo Frame constructor runs the two extractions on two threads
    (core/src/Frame.cc:120-181: threadLeft / threadRight, then ComputeStereoMatches).  Two handles
    called at the same time from two threads, many frames in a row, must give what the same
    calls give one after the other, and the association that follows must equal the oracle's."""
    import threading
    from tests.conftest import HOSTSIM
    if HOSTSIM:
        pytest.skip("the kernel host simulation runs one launch at a time from one thread")
    params = (1200, 1.2, 8, 20, 7)
    lefts = [orbx.textured_frame(810 + i, 752, 480) for i in range(6)]
    rights = [right_view(l, 810 + i) for i, l in enumerate(lefts)]
    exl, exr = orbx.ORBextractor(*params), orbx.ORBextractor(*params)
    serial = [(exl(l, None, (0, 0)), exr(r, None, (0, 0))) for l, r in zip(lefts, rights)]
    failures = []

    def work(ex, imgs, side, out):
        try:
            for rep in range(8):
                for i, img in enumerate(imgs):
                    mono, k, d = ex(img, None, (0, 0))
                    ref = serial[i][side]
                    if mono != ref[0] or not np.array_equal(k, ref[1]) or not np.array_equal(d, ref[2]):
                        failures.append((side, rep, i))
            out.append(True)
        except Exception as e:   # noqa: BLE001 - reported by the assertion below
            failures.append(repr(e))

    done = []
    threads = [threading.Thread(target=work, args=(exl, lefts, 0, done)),
               threading.Thread(target=work, args=(exr, rights, 1, done))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures and len(done) == 2, failures[:5]
    # both handles now hold the last pair
    ur, dp = orbx.compute_stereo_matches(exl, exr, MBF, MB)
    rur, rdp, _ = oracle_stereo(oracle, lefts[-1], rights[-1], params)
    assert same_floats(ur, rur) and same_floats(dp, rdp)
