"""Host-pointer calls with a few frames (a stereo pair, a rig of four cameras): time per call of
orbx_extract_batch against the same frames resident on the device.
Usage (GPU box): python tools/small_batch_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import orb_slam3_amd as orbx
from orb_slam3_amd._lib import lib, check
W, H = 752, 480
frames = np.stack([orbx.textured_frame(100 + i, W, H) for i in range(8)])
d = torch.from_numpy(frames).cuda()
for B in (1, 2, 4, 8):
    ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_batch=B)
    cap = ex.max_keypoints()
    ptrs = (C.c_void_p * B)(*[frames[i].ctypes.data for i in range(B)])
    k = np.zeros((B, cap * 28), np.uint8); de = np.zeros((B, cap * 32), np.uint8)
    n = np.zeros(B, np.int32); mono = np.zeros(B, np.int32)
    i32p = C.POINTER(C.c_int32)
    def host():
        check(lib().orbx_extract_batch(ex._h, B, ptrs, W, H, W, 0, 1000, k.ctypes.data_as(C.c_void_p),
                                       de.ctypes.data_as(C.c_void_p), cap, n.ctypes.data_as(i32p),
                                       mono.ctypes.data_as(i32p)), "orbx_extract_batch")
    def dev():
        ex.extract_batch_device(d.data_ptr(), B, W, H, W, (0, 1000)); ex.sync()
    out = []
    for fn in (host, dev, host):
        for _ in range(30):
            fn()
        per = []
        for _ in range(200):
            t0 = time.perf_counter(); fn(); per.append(time.perf_counter() - t0)
        out.append((np.median(per) * 1e3, np.mean(per) * 1e3, max(per) * 1e3))
    print("%d frames: host call median %.3f ms (mean %.3f, slowest %.2f; again after the device-resident loop: median %.3f, "
          "mean %.3f, slowest %.2f), device-resident chain %.3f ms"
          % (B, out[0][0], out[0][1], out[0][2], out[2][0], out[2][1], out[2][2], out[1][0]), flush=True)

# one stereo frame the way Frame's stereo constructor makes it (core/src/Frame.cc:120-181):
# left and right extraction on two handles, then ComputeStereoMatches
from orb_slam3_amd import matcher
L = orbx.textured_frame(300, W, H)
R = np.roll(L, -12, axis=1)
exl = orbx.ORBextractor(1200, 1.2, 8, 20, 7, device=0)
exr = orbx.ORBextractor(1200, 1.2, 8, 20, 7, device=0)
def frame():
    exl(L, None, (0, 0)); exr(R, None, (0, 0))
    return matcher.compute_stereo_matches(exl, exr, 47.9, 0.11)
for part, fn in (("both extractions + association", frame),
                 ("association alone", lambda: matcher.compute_stereo_matches(exl, exr, 47.9, 0.11))):
    for _ in range(30):
        fn()
    per = []
    for _ in range(200):
        t0 = time.perf_counter(); fn(); per.append(time.perf_counter() - t0)
    print("stereo frame, %s: median %.3f ms" % (part, np.median(per) * 1e3), flush=True)
ur, dp = frame()
print("matched %d of %d" % (int((ur >= 0).sum()), len(ur)))
# the same with the two extractions on two threads, as the reference's constructor runs them
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(2)
def frame_threads():
    a = pool.submit(exl, L, None, (0, 0)); b = pool.submit(exr, R, None, (0, 0))
    a.result(); b.result()
    return matcher.compute_stereo_matches(exl, exr, 47.9, 0.11)
for _ in range(30):
    frame_threads()
per = []
for _ in range(200):
    t0 = time.perf_counter(); frame_threads(); per.append(time.perf_counter() - t0)
print("stereo frame, extractions on two threads + association: median %.3f ms" % (np.median(per) * 1e3), flush=True)

# one fisheye stereo frame (core/src/Frame.cc:1159-1177): two extractions with lapping areas, then
# the 2-NN of the left lapping descriptors against the right ones, blocking
exl = orbx.ORBextractor(1500, 1.2, 8, 20, 7, device=0)
exr = orbx.ORBextractor(1500, 1.2, 8, 20, 7, device=0)
Lf = orbx.textured_frame(400, 512, 512); Rf = orbx.textured_frame(401, 512, 512)
def fisheye():
    exl(Lf, None, (0, 511)); exr(Rf, None, (0, 511))
    return matcher.lapping_knn_match2(exl, exr, 0)
for part, fn in (("both extractions + 2-NN", fisheye), ("2-NN alone", lambda: matcher.lapping_knn_match2(exl, exr, 0))):
    for _ in range(30):
        fn()
    per = []
    for _ in range(200):
        t0 = time.perf_counter(); fn(); per.append(time.perf_counter() - t0)
    print("fisheye frame, %s: median %.3f ms" % (part, np.median(per) * 1e3), flush=True)
