#!/usr/bin/env python3
"""Randomised differential run of the whole path against the CPU oracle.

Every case draws an image size, content, extractor parameters, thresholds and a lapping area,
compares the extraction stage by stage (pyramid bytes, candidates, quadtree selections, blurred
bytes, keypoints, descriptors), then the stereo association, the frame grid with window queries,
both guided matchers and the bag-of-words transform on what was extracted, and a brute-force 2-NN
on random descriptor sets - all bit-exact except
the angle (1e-4 rad).  On a GPU box it runs the real library; with ORBX_HOSTSIM=1 and no GPU it
runs the kernels under tests/hostsim (guarded memory), which is how it is used where there is no
device.  The case number is the seed: `--only N` repeats one case.

    ORBX_HOSTSIM=1 python tools/fuzz_parity.py --cases 200 [--start 0] [--only N] [--max-pixels 400000]
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import conftest  # noqa: E402  (loads the host simulation when asked to)
import orb_slam3_amd as orbx  # noqa: E402
from orb_slam3_amd import bow, grid, guided  # noqa: E402
from tests import oracle_lib as oracle  # noqa: E402
from tests import voc_util  # noqa: E402
from tests.test_gpu_extract import compare_full  # noqa: E402


def content(rng, kind, w, h):
    if kind == "textured":
        return orbx.textured_frame(int(rng.integers(0, 1 << 30)), w, h)
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "faint":
        base, amp = int(rng.integers(20, 200)), int(rng.integers(8, 40))
        return (base + rng.integers(0, amp + 1, (h, w))).astype(np.uint8)
    if kind == "binary":
        lo, d = int(rng.integers(0, 200)), int(rng.integers(6, 56))
        return np.where(rng.random((h, w)) < rng.uniform(0.2, 0.8), lo, lo + d).astype(np.uint8)
    if kind == "checker":
        s = int(rng.integers(2, 24))
        yy, xx = np.mgrid[0:h, 0:w]
        img = (((yy // s + xx // s) & 1) * int(rng.integers(30, 256))).astype(np.int32)
        return np.clip(img + rng.integers(0, 3, (h, w)), 0, 255).astype(np.uint8)
    if kind == "dots":
        img = np.full((h, w), int(rng.integers(0, 120)), np.uint8)
        n = int(rng.integers(5, 400))
        img[rng.integers(0, h, n), rng.integers(0, w, n)] = 255
        return img
    if kind == "ramp":
        g = np.linspace(0, 255, w)[None, :] * np.linspace(0.2, 1, h)[:, None]
        return np.clip(g + rng.integers(-8, 9, (h, w)), 0, 255).astype(np.uint8)
    if kind == "patchwork":
        # textured on one side, faint noise elsewhere, a flat band: cells of every kind
        img = orbx.textured_frame(int(rng.integers(0, 1 << 30)), w, h).copy()
        x0 = int(rng.integers(w // 4, 3 * w // 4))
        img[:, x0:] = (90 + rng.integers(0, int(rng.integers(10, 30)), (h, w - x0))).astype(np.uint8)
        y0 = int(rng.integers(0, h - 8))
        img[y0:y0 + int(rng.integers(8, 80)), :] = int(rng.integers(0, 256))
        return img
    raise ValueError(kind)


KINDS = ["textured", "textured", "textured", "noise", "faint", "binary", "checker", "dots", "ramp", "patchwork",
         "patchwork"]


def draw_case(seed, max_pixels):
    rng = np.random.default_rng(seed)
    while True:
        w, h = int(rng.integers(70, 1000)), int(rng.integers(70, 700))
        if w * h > max_pixels:
            continue
        # mostly shapes the reference handles; one in ten draws may be a tall or a very wide strip
        if rng.random() < 0.9 and ((w - 32) * 2 <= (h - 32) or (w - 32) > 16 * (h - 32)):
            continue
        break
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.25, 1.3, 1.5, 2.0]))
    # every level must keep one 35-pixel cell: min(w, h) / sf^(nl-1) >= 67
    nl_max = 1
    while nl_max < 8 and round(min(w, h) / sf ** nl_max) >= 67:
        nl_max += 1
    nl = int(rng.integers(1, nl_max + 1))
    nf = int(rng.choice([int(rng.integers(1, 60)), int(rng.integers(60, 1500)), int(rng.integers(1500, 6000))],
                        p=[0.15, 0.7, 0.15]))
    ini = int(rng.integers(5, 60))
    mn = int(rng.integers(1, ini + 1)) if rng.random() < 0.85 else int(rng.integers(ini, ini + 20))
    lap = [(0, 0), (0, 1000), (int(rng.integers(0, w)), int(rng.integers(0, 2 * w)))][int(rng.integers(0, 3))]
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    return rng, (w, h), (nf, sf, nl, ini, mn), lap, kind


def both_refuse(img, params, lap):
    """Shapes the reference has no defined result for (a level without a 35-pixel cell, or a level
    more than twice as tall as wide, where extractor.cc:549-551 divides by nIni = 0) must be
    refused by both sides with the same status; strips wider than 16.5 times their height are
    beyond the quadtree kernel's root table and refused by the library alone."""
    st = oracle.Extractor(*params).extract(img, lap)[0]
    try:
        orbx.ORBextractor(*params)(img, None, lap)
        got = 0
    except orbx.OrbxError as e:
        got = e.status
    h, w = img.shape
    sf, nl = params[1], params[2]
    widest = max((round(w / sf ** l) - 32) / (round(h / sf ** l) - 32) for l in range(nl))
    if got == -7 and st == 0 and widest > 16.4:
        return "refused by the library as unsupported (more than 16 quadtree roots)"
    assert got == st, "status %d, oracle %d" % (got, st)
    return "refused by both (status %d)" % st if st else None


def same_floats(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def stereo_part(rng, img, params):
    w = img.shape[1]
    d = int(rng.integers(0, 30))
    right = np.ascontiguousarray(np.roll(img, -d, axis=1))
    right = np.clip(right.astype(np.int32) + rng.integers(-2, 3, right.shape), 0, 255).astype(np.uint8)
    mb = float(rng.uniform(0.05, 0.5))
    mbf = float(rng.uniform(0.3, 1.2)) * w * mb
    exl, exr = orbx.ORBextractor(*params), orbx.ORBextractor(*params)
    exl(img, None, (0, 0))
    exr(right, None, (0, 0))
    orbx.compute_stereo_matches_batch(exl, exr, 1, mbf, mb)
    ur, dp, sad = orbx.stereo_download(exl, 0, with_l1=True)
    ol, orr = oracle.Extractor(*params), oracle.Extractor(*params)
    _, _, kl, dl = ol.extract(img, (0, 0))
    _, _, kr, dr = orr.extract(right, (0, 0))
    rur, rdp, rsad = oracle.stereo_match(ol, orr, kl, dl, kr, dr, mbf, mb)
    assert same_floats(ur, rur) and same_floats(dp, rdp) and np.array_equal(sad, rsad), "stereo"
    return int((dp > 0).sum())


def grid_and_matchers(rng, img, params):
    h, w = img.shape
    ex = orbx.ORBextractor(*params)
    _, k, d = ex(img, None, (0, 0))
    if len(k) == 0:
        return "no keypoints"
    cam = None
    if rng.random() < 0.5:
        cam = (0.6 * w, 0.61 * w, 0.5 * w + 3, 0.5 * h - 2, float(rng.uniform(-0.35, 0.1)), float(rng.uniform(-0.1, 0.12)),
               float(rng.uniform(-1e-3, 1e-3)), float(rng.uniform(-1e-3, 1e-3)), float(rng.choice([0.0, -0.02, 0.01])))
    pc = grid.pinhole(*cam) if cam is not None else None
    b = grid.image_bounds(pc, w, h)
    assert same_floats(b, oracle.image_bounds(cam, w, h)), "bounds"
    if not (b[1] > b[0] and b[3] > b[2]):
        # distortion so strong that the five undistortion iterations send all four corners to one
        # point: the reference would go on with an infinite cell size; the library refuses
        from orb_slam3_amd._lib import OrbxError, ORBX_ERR_BAD_ARG
        try:
            grid.frame_grid_batch(ex, 1, pc, b)
        except OrbxError as e:
            assert e.status == ORBX_ERR_BAD_ARG
            return "degenerate image bounds, refused"
        raise AssertionError("degenerate bounds accepted")
    grid.frame_grid_batch(ex, 1, pc, b)
    un, cs, ent = grid.frame_grid_download(ex, 0)
    run, rcs, rent = oracle.frame_grid(cam, b, k)
    assert np.array_equal(un.view(np.uint8), run.view(np.uint8)), "undistorted keypoints"
    assert np.array_equal(cs, rcs) and np.array_equal(ent, rent), "grid"
    nq = 120
    q = np.zeros((nq, 5), np.float32)
    pick = rng.integers(0, len(un), nq)
    q[:, 0] = un["x"][pick] + rng.uniform(-30, 30, nq)
    q[:, 1] = un["y"][pick] + rng.uniform(-30, 30, nq)
    q[:, 2] = rng.choice([1.0, 5.0, 15.0, 40.0, 300.0], nq)
    q[:, 3] = rng.choice([-1, 0, 1, 3], nq)
    q[:, 4] = rng.choice([-1, 0, 2, 7], nq)
    cnt, idx = grid.features_in_area(ex, 0, q, per_query_cap=32)
    for i in range(nq):
        n, out = oracle.features_in_area(b, run, rcs, rent, float(q[i, 0]), float(q[i, 1]), float(q[i, 2]),
                                         int(q[i, 3]), int(q[i, 4]), cap=32)
        assert cnt[i] == n and np.array_equal(idx[i, :min(n, 32)], out[:min(n, 32)]), ("window query", i)

    # the "other frame": the same keypoints moved a little, descriptors with flipped bits
    n1 = int(rng.integers(1, len(k) + 1))
    sel = rng.permutation(len(k))[:n1]
    ka = un[sel].copy()
    da = d[sel].copy()
    flips = int(rng.integers(0, 60))
    for _ in range(flips):
        da[np.arange(n1), rng.integers(0, 32, n1)] ^= (1 << rng.integers(0, 8, n1)).astype(np.uint8)
    jit = float(rng.choice([0.5, 3.0, 12.0]))
    uv = np.stack([ka["x"] + rng.uniform(-jit, jit, n1), ka["y"] + rng.uniform(-jit, jit, n1)], 1).astype(np.float32)
    invz = rng.uniform(-0.05, 1.0, n1).astype(np.float32)
    valid = (rng.random(n1) < 0.9).astype(np.uint8)
    obs = (rng.random(n1) < rng.choice([0.0, 0.6, 1.0])).astype(np.uint8)
    state = rng.choice([0, 0, 0, 1, 2], len(k)).astype(np.uint8)
    ur = np.where(rng.random(len(k)) < 0.5, un["x"] - rng.uniform(0, 30, len(k)), -1).astype(np.float32)
    sf = np.array(ex.GetScaleFactors(), np.float32)
    th = float(rng.choice([3.0, 7.0, 15.0, 40.0]))
    fwd, bwd = [(0, 0), (1, 0), (0, 1)][int(rng.integers(0, 3))]
    use_ur = ur if rng.random() < 0.5 else None
    use_state = state if rng.random() < 0.7 else None
    ori = bool(rng.random() < 0.7)
    want_n, want = oracle.search_by_projection_last(valid, uv, invz, ka["octave"], ka["angle"], da, obs, un, d, use_ur,
                                                    use_state if use_state is not None else np.zeros(len(k), np.uint8),
                                                    b, cs, ent, sf, th, fwd, bwd, 40.0, ori)
    n, asg, rounds = guided.search_by_projection_last(ex, 0, valid, uv, invz, ka["octave"], ka["angle"], da, obs, use_ur,
                                                      use_state, th, fwd, bwd, 40.0, ori)
    assert n == want_n and np.array_equal(asg, want), "search by projection"

    # the same queries as a keyframe's map points during relocalisation (matcher.cc:1810-1923)
    has_point = (rng.random(len(k)) < rng.choice([0.0, 0.2])).astype(np.uint8)
    lvl_kf = np.clip(ka["octave"] + rng.integers(-1, 2, n1), 0, params[2] - 1).astype(np.int32)
    orb_dist = int(rng.choice([0, 50, 64, 100, 255]))
    want_nk, want_k = oracle.search_by_projection_keyframe(valid, uv, lvl_kf, ka["angle"], da, un, d, has_point, b, cs,
                                                           ent, sf, th, orb_dist, ori)
    nk, asgk, _ = guided.search_by_projection_keyframe(ex, 0, valid, uv, lvl_kf, ka["angle"], da, has_point, th,
                                                       orb_dist, ori)
    assert nk == want_nk and np.array_equal(asgk, want_k), "search by projection from a keyframe"

    # the same keypoints as map points in view (several per keypoint)
    copies = int(rng.integers(1, 4))
    kp3, dp3 = np.concatenate([ka] * copies), np.concatenate([da] * copies)
    n3 = len(kp3)
    proj = np.stack([kp3["x"] + rng.uniform(-jit, jit, n3), kp3["y"] + rng.uniform(-jit, jit, n3)], 1).astype(np.float32)
    proj_xr = (proj[:, 0] - rng.uniform(0, 30, n3)).astype(np.float32)
    level = np.clip(kp3["octave"] + rng.integers(-1, 2, n3), 0, params[2] - 1).astype(np.int32)
    vcos = np.where(rng.random(n3) < 0.5, 0.9995, rng.uniform(0.5, 0.998, n3)).astype(np.float32)
    in_view = (rng.random(n3) < 0.9).astype(np.uint8)
    obs3 = (rng.random(n3) < rng.choice([0.0, 0.6, 1.0])).astype(np.uint8)
    th3, ratio3 = float(rng.choice([1.0, 3.0, 5.0, 15.0])), float(rng.choice([0.6, 0.8, 0.9, 1.0]))
    want_n, want = oracle.search_by_projection_points(in_view, proj, proj_xr, level, vcos, dp3, obs3, un, d, use_ur,
                                                      use_state if use_state is not None
                                                      else np.zeros(len(k), np.uint8), b, cs, ent, sf, th3, ratio3)
    n3m, asg3, rounds3 = guided.search_by_projection_points(ex, 0, in_view, proj, proj_xr, level, vcos, dp3, obs3,
                                                            use_ur, use_state, th3, ratio3)
    assert n3m == want_n and np.array_equal(asg3, want), "search by projection of map points"

    # the frame as the left camera of a rig (Nleft != -1): the right camera sees the same content
    # a few pixels to the side; both matchers in their two-camera forms
    shift = int(rng.integers(2, 12))
    right = np.ascontiguousarray(np.roll(img, -shift, axis=1))
    exr = orbx.ORBextractor(*params)
    _, kr_raw, drr = exr(right, None, (0, 0))
    if len(kr_raw) > 0:
        grid.frame_grid_batch(exr, 1, pc, b)
        unr, csr, entr = grid.frame_grid_download(exr, 0)
        state_r = rng.choice([0, 0, 0, 1, 2], len(unr)).astype(np.uint8)
        st_l = use_state if use_state is not None else np.zeros(len(k), np.uint8)
        uv_r = (uv - np.array([shift, 0], np.float32) + rng.uniform(-jit, jit, uv.shape)).astype(np.float32)
        want_n, want_l, want_r = oracle.search_by_projection_last_rig(
            valid, uv, invz, uv_r, ka["octave"], ka["angle"], da, obs, (un, d, st_l, cs, ent),
            (unr, drr, state_r, csr, entr), b, sf, th, fwd, bwd, ori)
        nrig, al, ar, _ = guided.search_by_projection_last_rig(ex, exr, 0, valid, uv, invz, uv_r, ka["octave"],
                                                               ka["angle"], da, obs, use_state, state_r, th, fwd, bwd,
                                                               ori)
        assert nrig == want_n and np.array_equal(al, want_l) and np.array_equal(ar, want_r), "rig search by projection"
        l2r = np.where(rng.random(len(un)) < 0.4, rng.integers(0, len(unr), len(un)), -1).astype(np.int32)
        r2l = np.where(rng.random(len(unr)) < 0.4, rng.integers(0, len(un), len(unr)), -1).astype(np.int32)
        proj_r = (proj - np.array([shift, 0], np.float32) + rng.uniform(-jit, jit, proj.shape)).astype(np.float32)
        level_r = np.where(rng.random(n3) < 0.1, -1, np.clip(level + rng.integers(-1, 2, n3), 0, params[2] - 1)).astype(np.int32)
        vcos_r = np.where(rng.random(n3) < 0.5, 0.9995, rng.uniform(0.5, 0.998, n3)).astype(np.float32)
        in_view_r = (rng.random(n3) < 0.8).astype(np.uint8)
        want_n, want_l, want_r = oracle.search_by_projection_points_rig(
            in_view, proj, level, vcos, in_view_r, proj_r, level_r, vcos_r, dp3, obs3, (un, d, st_l, cs, ent),
            (unr, drr, state_r, csr, entr), l2r, r2l, b, sf, th3, ratio3)
        nrig3, al, ar, _ = guided.search_by_projection_points_rig(ex, exr, 0, in_view, proj, level, vcos, in_view_r,
                                                                  proj_r, level_r, vcos_r, dp3, obs3, l2r, r2l,
                                                                  use_state, state_r, th3, ratio3)
        assert nrig3 == want_n and np.array_equal(al, want_l) and np.array_equal(ar, want_r), "rig search of map points"

    ka0 = ka.copy()
    if rng.random() < 0.7:
        ka0["octave"] = np.where(rng.random(n1) < 0.8, 0, ka0["octave"])
    prev = uv.copy()
    window = int(rng.choice([10, 30, 100, 250]))
    ratio = float(rng.choice([0.6, 0.9, 0.95, 1.0]))
    want_n, want, want_pm = oracle.search_for_initialization(ka0, da, un, d, b, cs, ent, prev, window, ratio, ori)
    n2, m12, pm, rounds2 = guided.search_for_initialization(ex, 0, ka0, da, prev, window, ratio, ori)
    assert n2 == want_n and np.array_equal(m12, want), "search for initialization"
    assert same_floats(pm, want_pm), "vbPrevMatched"

    kk, L = int(rng.integers(2, 17)), int(rng.integers(1, 5))
    while kk ** L > 20000:
        L -= 1
    arrays = voc_util.make_vocabulary(kk, L, seed=int(rng.integers(0, 1 << 30)), ragged=bool(rng.random() < 0.5),
                                      stop_fraction=float(rng.choice([0.0, 0.2])))
    voc = bow.ORBVocabulary.from_arrays(kk, L, *arrays)
    ovoc = oracle.Vocabulary(arrays=(kk, L) + arrays + (0,))
    levelsup = int(rng.integers(0, 6))
    bow.transform_batch(ex, voc, 1, levelsup)
    wi, wv, ni, ns, ft = bow.download(ex, 0)
    rwi, rwv, rni, rns, rft = ovoc.transform(d, levelsup)
    assert np.array_equal(wi, rwi) and np.array_equal(ni, rni), "bow ids"
    assert np.array_equal(wv.view(np.uint64), rwv.view(np.uint64)), "bow values"
    assert np.array_equal(ns, rns) and np.array_equal(ft, rft), "feature vector"
    # the "other frame" as a keyframe: its feature vector from the oracle's transform
    _, _, an, as_, af = ovoc.transform(da, levelsup)
    ratio_b = float(rng.choice([0.6, 0.75, 0.9, 1.0]))
    want_nb, want_b = oracle.search_by_bow(valid, ka["angle"], da, an, as_, af, k["angle"], d, rni, rns, rft, ratio_b, ori)
    nb, mb_ = guided.search_by_bow(ex, 0, valid, ka["angle"], da, an, as_, af, ratio_b, ori)
    assert nb == want_nb and np.array_equal(mb_, want_b), "search by bag of words"
    return "proj %d in %d rounds, points %d in %d rounds, init %d in %d rounds, %d words, %d by bow" % (
        n, rounds, n3m, rounds3, n2, rounds2, len(wi), nb)


def batch_part(rng, img, params, lap):
    """The frame inside a batch of three with other content before and after it: every frame of
    the batch equals its single-frame extraction."""
    h, w = img.shape
    others = [content(rng, KINDS[int(rng.integers(0, len(KINDS)))], w, h) for _ in range(2)]
    frames = [others[0], img, others[1]]
    ex = orbx.ORBextractor(*params, max_batch=3)
    res = ex.extract_batch(frames, lap)
    single = orbx.ORBextractor(*params)
    for f in range(3):
        mono, k, d = single(frames[f], None, lap)
        bm, bk, bd = res[f]
        assert bm == mono and np.array_equal(bk, k), ("batch frame", f)
        assert (d is None and bd is None) or np.array_equal(bd, d), ("batch descriptors", f)


def knn_part(rng):
    """Brute-force 2-NN on random sizes; a small pool of distinct rows makes distance ties the rule."""
    nq, nt = int(rng.integers(1, 1300)), int(rng.choice([0, 1, 2, int(rng.integers(3, 6000))], p=[.03, .03, .04, .9]))
    pool = rng.integers(0, 256, (int(rng.choice([3, 40, 100000])), 32), dtype=np.uint8)
    t = pool[rng.integers(0, len(pool), nt)] if len(pool) < 100000 else rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = pool[rng.integers(0, len(pool), nq)].copy() if len(pool) < 100000 else rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    q[:, int(rng.integers(0, 32))] ^= rng.integers(0, 4, nq).astype(np.uint8)
    idx, dist = orbx.knn_match2(q, np.ascontiguousarray(t))
    ridx, rdist = oracle.knn2(q, np.ascontiguousarray(t))
    assert np.array_equal(dist, rdist) and np.array_equal(idx, ridx), "2-NN %d x %d" % (nq, nt)
    return "2-NN %dx%d" % (nq, nt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=50)
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--max-pixels", type=int, default=400000)
    ap.add_argument("--keep-going", action="store_true")
    args = ap.parse_args()
    seeds = [args.only] if args.only is not None else range(args.start, args.start + args.cases)
    failures = 0
    for seed in seeds:
        rng, (w, h), params, lap, kind = draw_case(seed, args.max_pixels)
        t0 = time.time()
        head = "case %d: %dx%d %s params %s lap %s" % (seed, w, h, kind, params, lap)
        try:
            knn = knn_part(rng)
            img = content(rng, kind, w, h)
            refused = both_refuse(img, params, lap)
            if refused:
                print("%s: %s" % (head, refused), flush=True)
                continue
            bad = compare_full(oracle, img, params, lap)
            if seed % 3 == 0:
                batch_part(rng, img, params, lap)
            msg = "%d descriptor rows differ" % bad if bad else "extraction equal"
            nstereo = stereo_part(rng, img, params)
            rest = grid_and_matchers(rng, img, params)
            print("%s: %s, %d stereo matches, %s, %s (%.1f s)" % (head, msg, nstereo, rest, knn, time.time() - t0),
                  flush=True)
        except Exception:
            failures += 1
            print("%s: FAILED" % head, flush=True)
            traceback.print_exc()
            if not args.keep_going:
                return 1
    print("%d cases, %d failures" % (len(list(seeds)), failures), flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
