"""Latency of one single-frame extraction split into its parts (device work, download, full call)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orb_slam3_amd as orbx
frames = [orbx.textured_frame(i, 752, 480) for i in range(8)]
ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7)
d = torch.from_numpy(frames[0]).cuda()
def t(fn, n=100):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
def dev():
    ex.extract_batch_device(d.data_ptr(), 1, 752, 480, 752, (0, 1000)); ex.sync()
print("device 1 frame + sync: %.3f ms" % t(dev))
print("download: %.3f ms" % t(lambda: ex.download(0)))
print("full __call__: %.3f ms" % t(lambda: ex(frames[0], None, (0, 1000))))
h = torch.from_numpy(frames[0])
print("torch H2D copy of one frame: %.3f ms" % t(lambda: (d.copy_(h), torch.cuda.synchronize())))
hp = h.pin_memory()
print("torch H2D pinned: %.3f ms" % t(lambda: (d.copy_(hp), torch.cuda.synchronize())))
