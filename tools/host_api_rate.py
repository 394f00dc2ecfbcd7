#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (orbx_extract_batch, orbx_extract):
images start in host memory and keypoints/descriptors end there."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_amd as orbx  # noqa: E402

B = 64
frames = [orbx.textured_frame(i, 752, 480) for i in range(B)]
ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
ex.extract_batch(frames, (0, 1000))
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    ex.extract_batch(frames, (0, 1000))
dt = time.perf_counter() - t0
print("orbx_extract_batch (host in/out, %d frames per call): %.0f frames/s" % (B, B * reps / dt))
ex1 = orbx.ORBextractor(1000, 1.2, 8, 20, 7)
ex1(frames[0], None, (0, 1000))
t0 = time.perf_counter()
for i in range(200):
    ex1(frames[i % B], None, (0, 1000))
dt = time.perf_counter() - t0
print("orbx_extract (one frame per call, blocking): %.0f frames/s, %.3f ms latency" % (200 / dt, dt / 200 * 1e3))
