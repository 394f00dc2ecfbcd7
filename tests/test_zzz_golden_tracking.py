"""Committed golden vectors of the guided matchers and the bag-of-words transform
(tests/golden/guided_376x240.npz, bow_376x240.npz, made by tests/golden/make_golden_tracking.py
from the CPU oracle).  CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them
without the oracle being present."""
import os
import zlib

import numpy as np
import pytest

import orb_slam3_amd as orbx

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _frames(g):
    prm = g["params"]
    params = (int(prm[0]), float(prm[1]), int(prm[2]), int(prm[3]), int(prm[4]))
    w, h, shift = int(g["w"]), int(g["h"]), int(g["shift"])
    wide = orbx.textured_frame(int(g["seed"]), w + shift, h)
    a, b = np.ascontiguousarray(wide[:, shift:]), np.ascontiguousarray(wide[:, :w])
    assert zlib.crc32(b.tobytes()) == int(g["b_crc32"])
    return params, a, b, w, h


def test_oracle_reproduces_guided_golden(oracle):
    g = np.load(os.path.join(HERE, "guided_376x240.npz"))
    params, a, b, w, h = _frames(g)
    eb = oracle.Extractor(*params)
    _, _, kb, db = eb.extract(b, (0, 0))
    bounds = oracle.image_bounds(None, w, h)
    un, cs, ent = oracle.frame_grid(None, bounds, kb)
    ka, da = g["last_keypoints"], g["last_descriptors"]
    nm, asg = oracle.search_by_projection_last(g["valid"], g["uv"], g["invz"], ka["octave"], ka["angle"], da, g["obs"], un,
                                               db, g["u_right"], g["state"], bounds, cs, ent,
                                               np.array(eb.tables()["scale"], np.float32), float(g["th"]), 0, 0,
                                               float(g["mbf"]), True)
    assert nm == int(g["nmatches"]) and np.array_equal(asg, g["assigned"])
    prev = np.stack([ka["x"], ka["y"]], 1).astype(np.float32)
    ni, m12, pm = oracle.search_for_initialization(ka, da, un, db, bounds, cs, ent, prev, int(g["window"]),
                                                   float(g["nnratio"]), True)
    assert ni == int(g["init_nmatches"]) and np.array_equal(m12, g["matches12"]) and np.array_equal(pm, g["prev_matched"])
    dp2 = np.concatenate([da, da])
    npm, pasg = oracle.search_by_projection_points(g["pt_in_view"], g["pt_proj"], g["pt_proj_xr"], g["pt_level"],
                                                   g["pt_view_cos"], dp2, g["pt_obs"], un, db, g["u_right"], g["state"],
                                                   bounds, cs, ent, np.array(eb.tables()["scale"], np.float32),
                                                   float(g["pt_th"]), float(g["pt_ratio"]))
    assert npm == int(g["pt_nmatches"]) and np.array_equal(pasg, g["pt_assigned"])
    nk, kasg = oracle.search_by_projection_keyframe(g["valid"], g["uv"], g["kf_level"], ka["angle"], da, un, db,
                                                    g["kf_has_point"], bounds, cs, ent,
                                                    np.array(eb.tables()["scale"], np.float32), float(g["kf_th"]),
                                                    int(g["kf_orb_dist"]), True)
    assert nk == int(g["kf_nmatches"]) and np.array_equal(kasg, g["kf_assigned"])


def test_oracle_reproduces_bow_golden(oracle):
    g = np.load(os.path.join(HERE, "bow_376x240.npz"))
    params, a, b, w, h = _frames(g)
    _, _, kb, db = oracle.Extractor(*params).extract(b, (0, 0))
    voc = oracle.Vocabulary(arrays=(int(g["k"]), int(g["L"]), g["voc_parent"], g["voc_leaf"], g["voc_desc"],
                                    g["voc_weight"], 0))
    wi, wv, ni, ns, ft = voc.transform(db, int(g["levelsup"]))
    assert np.array_equal(wi, g["word_ids"]) and np.array_equal(wv.view(np.uint64), g["word_values"].view(np.uint64))
    assert np.array_equal(ni, g["node_ids"]) and np.array_equal(ns, g["node_start"]) and np.array_equal(ft, g["features"])
    n, m = oracle.search_by_bow(g["kf_valid"], g["kf_angle"], g["kf_desc"], g["kf_nodes"], g["kf_start"], g["kf_feats"],
                                kb["angle"], db, ni, ns, ft, float(g["sb_ratio"]), True)
    assert n == int(g["sb_nmatches"]) and np.array_equal(m, g["sb_matches"])


@pytest.mark.gpu
def test_hip_reproduces_guided_golden():
    from orb_slam3_amd import grid, guided
    g = np.load(os.path.join(HERE, "guided_376x240.npz"))
    params, a, b, w, h = _frames(g)
    ex = orbx.ORBextractor(*params)
    ex(b, None, (0, 0))
    grid.frame_grid_batch(ex, 1, None, grid.image_bounds(None, w, h))
    ka, da = g["last_keypoints"], g["last_descriptors"]
    nm, asg, _ = guided.search_by_projection_last(ex, 0, g["valid"], g["uv"], g["invz"], ka["octave"], ka["angle"], da,
                                                  g["obs"], g["u_right"], g["state"], float(g["th"]), 0, 0,
                                                  float(g["mbf"]), True)
    assert nm == int(g["nmatches"]) and np.array_equal(asg, g["assigned"])
    prev = np.stack([ka["x"], ka["y"]], 1).astype(np.float32)
    ni, m12, pm, _ = guided.search_for_initialization(ex, 0, ka, da, prev, int(g["window"]), float(g["nnratio"]), True)
    assert ni == int(g["init_nmatches"]) and np.array_equal(m12, g["matches12"]) and np.array_equal(pm, g["prev_matched"])
    npm, pasg, _ = guided.search_by_projection_points(ex, 0, g["pt_in_view"], g["pt_proj"], g["pt_proj_xr"], g["pt_level"],
                                                      g["pt_view_cos"], np.concatenate([da, da]), g["pt_obs"],
                                                      g["u_right"], g["state"], float(g["pt_th"]), float(g["pt_ratio"]))
    assert npm == int(g["pt_nmatches"]) and np.array_equal(pasg, g["pt_assigned"])
    nk, kasg, _ = guided.search_by_projection_keyframe(ex, 0, g["valid"], g["uv"], g["kf_level"], ka["angle"], da,
                                                       g["kf_has_point"], float(g["kf_th"]), int(g["kf_orb_dist"]), True)
    assert nk == int(g["kf_nmatches"]) and np.array_equal(kasg, g["kf_assigned"])


@pytest.mark.gpu
def test_hip_reproduces_bow_golden():
    from orb_slam3_amd import bow
    g = np.load(os.path.join(HERE, "bow_376x240.npz"))
    params, a, b, w, h = _frames(g)
    ex = orbx.ORBextractor(*params)
    ex(b, None, (0, 0))
    voc = bow.ORBVocabulary.from_arrays(int(g["k"]), int(g["L"]), g["voc_parent"], g["voc_leaf"], g["voc_desc"],
                                        g["voc_weight"])
    bow.transform_batch(ex, voc, 1, int(g["levelsup"]))
    wi, wv, ni, ns, ft = bow.download(ex, 0)
    assert np.array_equal(wi, g["word_ids"]) and np.array_equal(wv.view(np.uint64), g["word_values"].view(np.uint64))
    assert np.array_equal(ni, g["node_ids"]) and np.array_equal(ns, g["node_start"]) and np.array_equal(ft, g["features"])
    from orb_slam3_amd import guided
    n, m = guided.search_by_bow(ex, 0, g["kf_valid"], g["kf_angle"], g["kf_desc"], g["kf_nodes"], g["kf_start"],
                                g["kf_feats"], float(g["sb_ratio"]), True)
    assert n == int(g["sb_nmatches"]) and np.array_equal(m, g["sb_matches"])
