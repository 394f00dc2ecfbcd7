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
ex.set_profiling(True)
acc = {}
for _ in range(50):
    ex.extract_batch_device(d.data_ptr(), 1, W, H, W, (0, 1000)); ex.sync()
    for k, v in ex.stage_times_ms().items():
        acc[k] = acc.get(k, 0.0) + v / 50
print("stages, one frame, serialised (ms):", {k: round(v, 4) for k, v in acc.items()})
