"""Small batches with the quadtree built by one wave per (frame, level) or by a team of waves:
time per device-resident call for a range of batch sizes.  The library reads ORBX_OCT_TEAM when a
handle is created (largest batch that uses the team kernel; 0 = never).
Usage (GPU box): python tools/team_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import orb_slam3_amd as orbx
W, H = 752, 480
frames = np.stack([orbx.textured_frame(100 + i, W, H) for i in range(128)])
d = torch.from_numpy(frames).cuda()
for B in (1, 2, 4, 8, 16, 32, 64, 128):
    row = []
    for team in (0, 1 << 20):
        os.environ["ORBX_OCT_TEAM"] = str(team)
        ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_batch=B)
        for _ in range(10):
            ex.extract_batch_device(d.data_ptr(), B, W, H, W, (0, 1000)); ex.sync()
        t0 = time.perf_counter()
        for _ in range(100):
            ex.extract_batch_device(d.data_ptr(), B, W, H, W, (0, 1000)); ex.sync()
        row.append((time.perf_counter() - t0) / 100 * 1e3)
        del ex
    print("batch %4d: one wave %.3f ms, team %.3f ms" % (B, row[0], row[1]), flush=True)
