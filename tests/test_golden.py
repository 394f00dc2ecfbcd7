"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from
the CPU oracle).  CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them
without the oracle being present."""
import glob
import os
import zlib

import numpy as np
import pytest

import orb_slam3_amd as orbx

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(HERE, "*.npz"))
               if not os.path.basename(p).startswith(("knn2", "pair_match", "guided", "bow")))


def load(path):
    g = np.load(path)
    prm = g["params"]
    params = (int(prm[0]), float(prm[1]), int(prm[2]), int(prm[3]), int(prm[4]))
    img = orbx.textured_frame(int(g["seed"]), int(g["w"]), int(g["h"]))
    assert zlib.crc32(img.tobytes()) == int(g["image_crc32"])
    return g, params, img, tuple(int(v) for v in g["lap"])


def test_fixtures_exist():
    assert len(CASES) == 4


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_oracle_reproduces_golden(oracle, path):
    g, params, img, lap = load(path)
    e = oracle.Extractor(*params)
    st, mono, k, d = e.extract(img, lap)
    assert st == 0 and mono == int(g["mono"])
    assert np.array_equal(k, g["keypoints"]) and np.array_equal(d, g["descriptors"])
    crc = [zlib.crc32(e.padded(l)[:, :e.level_dims(l)[0] + 38].tobytes()) for l in range(params[2])]
    assert crc == g["pyramid_crc32"].tolist()


def test_oracle_knn_golden(oracle):
    g = np.load(os.path.join(HERE, "knn2_200x3000.npz"))
    idx, dist = oracle.knn2(g["q"], g["t"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_hip_reproduces_golden(path):
    g, params, img, lap = load(path)
    ex = orbx.ORBextractor(*params)
    mono, k, d = ex(img, None, lap)
    gk = g["keypoints"]
    assert mono == int(g["mono"]) and len(k) == len(gk)
    for fld in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(k[fld], gk[fld]), fld
    dang = np.abs(k["angle"] - gk["angle"])
    assert np.minimum(dang, 360 - dang).max() <= np.degrees(1e-4)
    bad = (np.unpackbits(d ^ g["descriptors"], axis=1).sum(axis=1) > 0)
    assert bad.mean() <= 1e-3
    for l in range(params[2]):
        w = ex.level_dims(l)[0]
        assert zlib.crc32(np.ascontiguousarray(ex.debug_padded(0, l)[:, :w + 38]).tobytes()) == int(g["pyramid_crc32"][l])


@pytest.mark.gpu
def test_hip_knn_golden():
    g = np.load(os.path.join(HERE, "knn2_200x3000.npz"))
    idx, dist = orbx.knn_match2(g["q"], g["t"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])


def _pair(g):
    from tests.stereo_util import right_view
    prm = g["params"]
    params = (int(prm[0]), float(prm[1]), int(prm[2]), int(prm[3]), int(prm[4]))
    left = orbx.textured_frame(int(g["seed"]), 752, 480)
    right = right_view(left, int(g["seed"]))
    assert zlib.crc32(right.tobytes()) == int(g["right_crc32"])
    return params, left, right


def test_oracle_pair_match_golden(oracle):
    g = np.load(os.path.join(HERE, "pair_match_752x480.npz"))
    params, left, right = _pair(g)
    el, er = oracle.Extractor(*params), oracle.Extractor(*params)
    _, _, kl, dl = el.extract(left, (0, 0))
    _, _, kr, dr = er.extract(right, (0, 0))
    ur, dp, l1 = oracle.stereo_match(el, er, kl, dl, kr, dr, float(g["mbf"]), float(g["mb"]))
    assert np.array_equal(ur, g["uright"]) and np.array_equal(dp, g["depth"]) and np.array_equal(l1, g["l1"])


@pytest.mark.gpu
def test_hip_pair_match_golden():
    g = np.load(os.path.join(HERE, "pair_match_752x480.npz"))
    params, left, right = _pair(g)
    exl, exr = orbx.ORBextractor(*params), orbx.ORBextractor(*params)
    exl(left, None, (0, 0))
    exr(right, None, (0, 0))
    orbx.compute_stereo_matches_batch(exl, exr, 1, float(g["mbf"]), float(g["mb"]))
    ur, dp, l1 = orbx.stereo_download(exl, 0, with_l1=True)
    assert np.array_equal(ur, g["uright"]) and np.array_equal(dp, g["depth"]) and np.array_equal(l1, g["l1"])
