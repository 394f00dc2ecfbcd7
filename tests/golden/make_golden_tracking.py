#!/usr/bin/env python3
"""Generates tests/golden/guided_376x240.npz and bow_376x240.npz from the CPU oracle: inputs and
expected outputs of SearchByProjection(Frame, LastFrame), SearchByProjection(Frame, map points),
SearchForInitialization and
ComputeBoW on a seeded synthetic frame pair.  Like the other fixtures they pin GPU <-> oracle
parity and guard the oracle against drift.

Run from the repo root:  python tests/golden/make_golden_tracking.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import orb_slam3_amd as orbx  # noqa: E402  (host-only helpers: synthetic frames)
from tests import oracle_lib as O  # noqa: E402
from tests import voc_util  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED, W, H, SHIFT = 41, 376, 240, 6
PARAMS = (400, 1.2, 4, 20, 7)


def frames():
    wide = orbx.textured_frame(SEED, W + SHIFT, H)
    return np.ascontiguousarray(wide[:, SHIFT:]), np.ascontiguousarray(wide[:, :W])


def main():
    a, b = frames()
    ea, eb = O.Extractor(*PARAMS), O.Extractor(*PARAMS)
    _, _, ka, da = ea.extract(a, (0, 0))
    _, _, kb, db = eb.extract(b, (0, 0))
    bounds = O.image_bounds(None, W, H)
    un, cs, ent = O.frame_grid(None, bounds, kb)
    sf = np.array(eb.tables()["scale"], np.float32)
    rng = np.random.default_rng(17)
    n_last = len(ka)
    uv = np.stack([ka["x"] + SHIFT + rng.uniform(-2, 2, n_last), ka["y"] + rng.uniform(-2, 2, n_last)], 1).astype(np.float32)
    uv[::19] += 500
    invz = rng.uniform(0.05, 1.0, n_last).astype(np.float32)
    invz[::23] = -0.1
    valid = (rng.random(n_last) < 0.9).astype(np.uint8)
    obs = (rng.random(n_last) < 0.7).astype(np.uint8)
    state = rng.choice([0, 0, 0, 1, 2], len(kb)).astype(np.uint8)
    u_right = np.where(rng.random(len(kb)) < 0.5, kb["x"] - rng.uniform(1, 30, len(kb)), -1).astype(np.float32)
    th, mbf = 15.0, 40.0
    nm, asg = O.search_by_projection_last(valid, uv, invz, ka["octave"], ka["angle"], da, obs, un, db, u_right, state,
                                          bounds, cs, ent, sf, th, 0, 0, mbf, True)
    prev = np.stack([ka["x"], ka["y"]], 1).astype(np.float32)
    ni, m12, pm = O.search_for_initialization(ka, da, un, db, bounds, cs, ent, prev, 60, 0.9, True)
    # the same keypoints as map points of the local map, two per keypoint (matcher.cc:41-142)
    rng2 = np.random.default_rng(18)
    kp2, dp2 = np.concatenate([ka, ka]), np.concatenate([da, da])
    n2 = len(kp2)
    pt_proj = np.stack([kp2["x"] + SHIFT + rng2.uniform(-4, 4, n2), kp2["y"] + rng2.uniform(-4, 4, n2)], 1).astype(np.float32)
    pt_proj_xr = (pt_proj[:, 0] - rng2.uniform(1, 30, n2)).astype(np.float32)
    pt_level = np.clip(kp2["octave"] + rng2.integers(-1, 2, n2), 0, PARAMS[2] - 1).astype(np.int32)
    pt_view_cos = np.where(rng2.random(n2) < 0.5, 0.9995, rng2.uniform(0.5, 0.998, n2)).astype(np.float32)
    pt_in_view = (rng2.random(n2) < 0.9).astype(np.uint8)
    pt_obs = (rng2.random(n2) < 0.7).astype(np.uint8)
    pt_th, pt_ratio = 3.0, 0.8
    np_, pasg = O.search_by_projection_points(pt_in_view, pt_proj, pt_proj_xr, pt_level, pt_view_cos, dp2, pt_obs, un, db,
                                              u_right, state, bounds, cs, ent, sf, pt_th, pt_ratio)
    print("guided", np_, "map-point matches of", n2)
    # relocalisation: the keypoints of frame a as a keyframe's map points (matcher.cc:1810-1923)
    rng4 = np.random.default_rng(20)
    kf_level = np.clip(ka["octave"] + rng4.integers(-1, 2, n_last), 0, PARAMS[2] - 1).astype(np.int32)
    kf_has_point = (rng4.random(len(kb)) < 0.2).astype(np.uint8)
    kf_n, kf_asg = O.search_by_projection_keyframe(valid, uv, kf_level, ka["angle"], da, un, db, kf_has_point, bounds,
                                                   cs, ent, sf, 10.0, 64, True)
    print("guided", kf_n, "keyframe projection matches")
    np.savez_compressed(os.path.join(HERE, "guided_376x240.npz"), kf_level=kf_level, kf_has_point=kf_has_point,
                        kf_th=10.0, kf_orb_dist=64, kf_nmatches=kf_n, kf_assigned=kf_asg, pt_in_view=pt_in_view, pt_proj=pt_proj,
                        pt_proj_xr=pt_proj_xr, pt_level=pt_level, pt_view_cos=pt_view_cos, pt_obs=pt_obs, pt_th=pt_th,
                        pt_ratio=pt_ratio, pt_nmatches=np_, pt_assigned=pasg, seed=SEED, shift=SHIFT, w=W, h=H,
                        params=np.array(PARAMS, np.float64), a_crc32=zlib.crc32(a.tobytes()),
                        b_crc32=zlib.crc32(b.tobytes()), last_keypoints=ka, last_descriptors=da, uv=uv, invz=invz,
                        valid=valid, obs=obs, state=state, u_right=u_right, th=th, mbf=mbf, nmatches=nm,
                        assigned=asg, window=60, nnratio=0.9, init_nmatches=ni, matches12=m12, prev_matched=pm)
    print("guided", nm, "projection matches,", ni, "initialisation matches of", n_last)
    k, L = 6, 3
    arrays = voc_util.make_vocabulary(k, L, seed=9)
    voc = O.Vocabulary(arrays=(k, L) + arrays + (0,))
    wi, wv, nodes, ns, ft = voc.transform(db, 2)
    # SearchByBoW with frame a as the keyframe (matcher.cc:222-405)
    _, _, kf_nodes, kf_start, kf_feats = voc.transform(da, 2)
    rng3 = np.random.default_rng(19)
    kf_valid = (rng3.random(len(ka)) < 0.85).astype(np.uint8)
    sb_n, sb_m = O.search_by_bow(kf_valid, ka["angle"], da, kf_nodes, kf_start, kf_feats, kb["angle"], db, nodes, ns, ft,
                                 0.7, True)
    print("bow", sb_n, "matches by bag of words")
    np.savez_compressed(os.path.join(HERE, "bow_376x240.npz"), kf_angle=ka["angle"], kf_desc=da, kf_valid=kf_valid,
                        kf_nodes=kf_nodes, kf_start=kf_start, kf_feats=kf_feats, sb_ratio=0.7, sb_nmatches=sb_n,
                        sb_matches=sb_m, seed=SEED, shift=SHIFT, w=W, h=H,
                        params=np.array(PARAMS, np.float64), b_crc32=zlib.crc32(b.tobytes()), k=k, L=L, levelsup=2,
                        voc_parent=arrays[0], voc_leaf=arrays[1], voc_desc=arrays[2], voc_weight=arrays[3],
                        word_ids=wi, word_values=wv, node_ids=nodes, node_start=ns, features=ft)
    print("bow", len(wi), "words,", len(nodes), "nodes,", len(ft), "features of", len(kb))


if __name__ == "__main__":
    main()
