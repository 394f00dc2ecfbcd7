#!/usr/bin/env python3
"""Stage durations (each stage alone on the chip, one stream) of a device-resident batch at a given
image size: python tools/stage_times_probe.py W H NFEATURES BATCH"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import orb_slam3_amd as orbx

w, h, nf, batch = (int(a) for a in sys.argv[1:5])
imgs = np.stack([orbx.textured_frame(7000 + i, w, h) for i in range(4)])
d = torch.from_numpy(np.tile(imgs, ((batch + 3) // 4, 1, 1))[:batch].copy()).cuda()
ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_batch=batch)
ex.extract_batch_device(d.data_ptr(), batch, w, h, w, (0, 0))
ex.sync()
ex.set_profiling(True)
acc = {}
for _ in range(3):
    ex.extract_batch_device(d.data_ptr(), batch, w, h, w, (0, 0))
    ex.sync()
    for k, v in ex.stage_times_ms().items():
        acc[k] = acc.get(k, 0.0) + v / 3
print("%dx%d, %d features, batch %d: stage ms" % (w, h, nf, batch), {k: round(v, 4) for k, v in acc.items()},
      "sum %.3f" % sum(acc.values()))
