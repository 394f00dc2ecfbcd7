"""Where the blocking single-frame call spends its time: the kernel chain alone (device-resident
frame, one sync) against the whole host-pointer call.  Usage (GPU box): python tools/single_frame_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import orb_slam3_amd as orbx
W, H = 752, 480
img = orbx.textured_frame(1, W, H)
d = torch.from_numpy(img).cuda()
ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, device=0)
for _ in range(20):
    ex.extract_batch_device(d.data_ptr(), 1, W, H, W, (0, 1000)); ex.sync()
t0 = time.perf_counter()
for _ in range(200):
    ex.extract_batch_device(d.data_ptr(), 1, W, H, W, (0, 1000)); ex.sync()
print("kernel chain + sync: %.3f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
for _ in range(20):
    ex(img, None, (0, 1000))
t0 = time.perf_counter()
for _ in range(200):
    ex(img, None, (0, 1000))
print("host call (python wrapper included): %.3f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
# the C entry point itself, image in ordinary and in page-locked memory
import ctypes as C
from orb_slam3_amd._lib import lib, check
cap = ex.max_keypoints()
k1 = np.zeros(cap * 28, np.uint8); d1 = np.zeros(cap * 32, np.uint8)
n1, m1 = C.c_int32(), C.c_int32()
pin = C.c_void_p()
check(lib().orbx_host_alloc(W * H, C.byref(pin)), "orbx_host_alloc")
C.memmove(pin, img.ctypes.data, W * H)
for name, addr in (("ordinary", img.ctypes.data), ("page-locked", pin.value), ("ordinary", img.ctypes.data), ("page-locked", pin.value)):
    def one():
        check(lib().orbx_extract(ex._h, C.c_void_p(addr), W, H, W, 0, 1000, k1.ctypes.data_as(C.c_void_p),
                                 d1.ctypes.data_as(C.c_void_p), cap, C.byref(n1), C.byref(m1)), "orbx_extract")
    for _ in range(20):
        one()
    per_call = []
    for _ in range(300):
        t0 = time.perf_counter()
        one()
        per_call.append(time.perf_counter() - t0)
    print("orbx_extract, image in %s memory: median %.3f ms, mean %.3f ms, slowest call %.1f ms (%d keypoints)"
          % (name, np.median(per_call) * 1e3, np.mean(per_call) * 1e3, max(per_call) * 1e3, n1.value))
check(lib().orbx_host_free(pin), "orbx_host_free")
ex.set_profiling(True)
acc = {}
for _ in range(50):
    ex.extract_batch_device(d.data_ptr(), 1, W, H, W, (0, 1000)); ex.sync()
    for k, v in ex.stage_times_ms().items():
        acc[k] = acc.get(k, 0.0) + v / 50
print("stages, one frame, serialised (ms):", {k: round(v, 4) for k, v in acc.items()})
