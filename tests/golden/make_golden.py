#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the CPU oracle.

The reference ships no fixtures for this path and cannot be built here (it needs OpenCV),
so these vectors pin GPU <-> oracle parity and guard the oracle against drift; they do not
pin oracle <-> OpenCV parity.  Inputs are produced by the deterministic generator
orbx_synth_frame (seed stored in the fixture) or stored verbatim when small.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import orb_slam3_amd as orbx  # noqa: E402  (host-only helpers: synthetic frames)
from tests import oracle_lib as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, seed, w, h, (nfeatures, scale, nlevels, ini, min), lapping
    ("mono_752x480", 0, 752, 480, (1000, 1.2, 8, 20, 7), (0, 1000)),
    ("stereo_752x480", 1, 752, 480, (1200, 1.2, 8, 20, 7), (0, 0)),
    ("fisheye_512x512", 2, 512, 512, (1500, 1.2, 8, 20, 7), (100, 400)),
    ("small_320x240", 3, 320, 240, (300, 1.2, 5, 20, 7), (0, 0)),
]


def main():
    for name, seed, w, h, prm, lap in CASES:
        img = orbx.textured_frame(seed, w, h)
        e = O.Extractor(*prm)
        st, mono, k, d = e.extract(img, lap)
        assert st == 0
        pyr_crc = [zlib.crc32(e.padded(l)[:, :e.level_dims(l)[0] + 38].tobytes()) for l in range(prm[2])]
        ncand = [len(e.candidates(l)) for l in range(prm[2])]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=seed, w=w, h=h,
                            params=np.array(prm, np.float64), lap=np.array(lap), mono=mono,
                            keypoints=k, descriptors=d, pyramid_crc32=np.array(pyr_crc, np.uint32),
                            ncand=np.array(ncand), image_crc32=zlib.crc32(img.tobytes()))
        print(name, len(k), mono, ncand)
    # stereo association (Frame::ComputeStereoMatches) on a synthetic rectified pair
    from tests.stereo_util import right_view
    prm, seed, mbf, mb = (1200, 1.2, 8, 20, 7), 11, np.float32(435.2 * 0.11), np.float32(0.11)
    left = orbx.textured_frame(seed, 752, 480)
    right = right_view(left, seed)
    el, er = O.Extractor(*prm), O.Extractor(*prm)
    _, _, kl, dl = el.extract(left, (0, 0))
    _, _, kr, dr = er.extract(right, (0, 0))
    ur, dp, sad = O.stereo_match(el, er, kl, dl, kr, dr, mbf, mb)
    np.savez_compressed(os.path.join(HERE, "pair_match_752x480.npz"), seed=seed, params=np.array(prm, np.float64),
                        mbf=mbf, mb=mb, right_crc32=zlib.crc32(right.tobytes()), uright=ur, depth=dp, l1=sad)
    print("pair_match", int((dp > 0).sum()), "matches of", len(kl))
    rng = np.random.default_rng(2024)
    t = rng.integers(0, 256, (3000, 32), dtype=np.uint8)
    t[1500:1600] = t[:100]  # duplicates: exact distance ties
    q = t[rng.integers(0, 3000, 200)].copy()
    q[::2] ^= (rng.integers(0, 256, (100, 32), dtype=np.uint8) & 0x21)
    idx, dist = O.knn2(q, t)
    np.savez_compressed(os.path.join(HERE, "knn2_200x3000.npz"), q=q, t=t, idx=idx, dist=dist)
    print("knn2", idx[:3].tolist(), dist[:3].tolist())


if __name__ == "__main__":
    main()
