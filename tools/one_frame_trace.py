"""300 blocking single-frame calls and nothing else, for a kernel trace of the latency path.
Usage (GPU box): rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/one_frame_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_amd as orbx
img = orbx.textured_frame(1, 752, 480)
ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, device=0)
for _ in range(300):
    ex(img, None, (0, 1000))
