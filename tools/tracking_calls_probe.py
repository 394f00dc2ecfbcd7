"""Blocking latency of the per-frame tracking calls around one extraction (752x480, 1000 keypoints):
frame grid, SearchByProjection(CurrentFrame, LastFrame), SearchByProjection(F, map points),
SearchForInitialization.  Medians over 200 calls.  Usage (GPU box): python tools/tracking_calls_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam3_amd as orbx
from orb_slam3_amd import grid, guided

W, H, SHIFT = 752, 480, 6
wide = orbx.textured_frame(41, W + SHIFT, H)
a = np.ascontiguousarray(wide[:, SHIFT:]); b = np.ascontiguousarray(wide[:, :W])
exa = orbx.ORBextractor(1000, 1.2, 8, 20, 7); exb = orbx.ORBextractor(1000, 1.2, 8, 20, 7)
_, ka, da = exa(a, None, (0, 0))
_, kb, db = exb(b, None, (0, 0))
bounds = grid.image_bounds(None, W, H)
rng = np.random.default_rng(5)
n_last = len(ka)
uv = np.stack([ka["x"] + SHIFT + rng.uniform(-2, 2, n_last), ka["y"] + rng.uniform(-2, 2, n_last)], 1).astype(np.float32)
invz = rng.uniform(0.05, 1.0, n_last).astype(np.float32)
valid = np.ones(n_last, np.uint8); obs = np.ones(n_last, np.uint8)
state = np.zeros(len(kb), np.uint8)


def med(fn, reps=200):
    for _ in range(20):
        fn()
    per = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); per.append(time.perf_counter() - t0)
    return np.median(per) * 1e3


print("extraction (orbx_extract):                %.3f ms" % med(lambda: exb(b, None, (0, 0))))
print("frame grid (orbx_frame_grid_batch_device + sync through a download of the counts): %.3f ms"
      % med(lambda: (grid.frame_grid_batch(exb, 1, None, bounds), exb.sync())))
grid.frame_grid_batch(exb, 1, None, bounds)
r = guided.search_by_projection_last(exb, 0, valid, uv, invz, ka["octave"], ka["angle"], da, obs, None, state, 15.0, 0, 0, 40.0, True)
print("SearchByProjection(last frame): %d matches in %d rounds" % (int((r[1] >= 0).sum()), r[2]))
print("SearchByProjection(last frame):           %.3f ms" % med(
    lambda: guided.search_by_projection_last(exb, 0, valid, uv, invz, ka["octave"], ka["angle"], da, obs, None, state,
                                             15.0, 0, 0, 40.0, True)))
print("whole tracking step (extraction, grid, SearchByProjection): %.3f ms" % med(
    lambda: (exb(b, None, (0, 0)), grid.frame_grid_batch(exb, 1, None, bounds),
             guided.search_by_projection_last(exb, 0, valid, uv, invz, ka["octave"], ka["angle"], da, obs, None, state,
                                              15.0, 0, 0, 40.0, True))))
# SearchByProjection(F, local map points): the projections of the other frame's keypoints, twice over
npts = 2 * n_last
proj = np.concatenate([uv, uv + 0.7]).astype(np.float32)
r = guided.search_by_projection_points(exb, 0, np.ones(npts, np.uint8), proj, np.full(npts, -1, np.float32),
                                       np.concatenate([ka["octave"], ka["octave"]]), np.full(npts, 0.999, np.float32),
                                       np.concatenate([da, da]), np.ones(npts, np.uint8), None, state, 1.0, 0.8)
print("SearchByProjection(map points): %d points, %d matches in %d rounds" % (npts, r[0], r[2]))
print("SearchByProjection(map points):           %.3f ms" % med(
    lambda: guided.search_by_projection_points(exb, 0, np.ones(npts, np.uint8), proj, np.full(npts, -1, np.float32),
                                               np.concatenate([ka["octave"], ka["octave"]]), np.full(npts, 0.999, np.float32),
                                               np.concatenate([da, da]), np.ones(npts, np.uint8), None, state, 1.0, 0.8)))
pm = np.stack([ka["x"], ka["y"]], 1).astype(np.float32)
r = guided.search_for_initialization(exb, 0, ka, da, pm, 100, 0.9, True)
print("SearchForInitialization: %d matches in %d rounds" % (r[0], r[3]))
print("SearchForInitialization:                  %.3f ms" % med(lambda: guided.search_for_initialization(exb, 0, ka, da, pm, 100, 0.9, True)))
# ComputeBoW of the frame and SearchByBoW against a keyframe (TrackReferenceKeyFrame): a synthetic
# tree with k = 10, L = 5 and the feature vector two levels below the root (100 nodes, as ORBvoc.txt's
# k = 10, L = 6 with levelsup = 4 gives)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import voc_util
from orb_slam3_amd import bow
arrays = voc_util.make_vocabulary(10, 5, seed=3)
voc = bow.ORBVocabulary.from_arrays(10, 5, *arrays)
exa2 = orbx.ORBextractor(1000, 1.2, 8, 20, 7)
exa2(a, None, (0, 0)); bow.transform_batch(exa2, voc, 1, 3)
kf = bow.download(exa2, 0)
print("ComputeBoW (orbx_bow_transform_batch_device + download): %.3f ms" % med(
    lambda: (bow.transform_batch(exb, voc, 1, 3), bow.download(exb, 0))))
valid1 = np.ones(len(ka), np.uint8)
print("bow.download returns:", [type(x).__name__ + str(getattr(x, "shape", "")) for x in kf])
_, _, kn, ks, kfeat = kf
r = guided.search_by_bow(exb, 0, valid1, ka["angle"], da, kn, ks, kfeat, 0.7, True)
print("SearchByBoW: %d matches" % r[0])
print("SearchByBoW:                              %.3f ms" % med(
    lambda: guided.search_by_bow(exb, 0, valid1, ka["angle"], da, kn, ks, kfeat, 0.7, True)))
