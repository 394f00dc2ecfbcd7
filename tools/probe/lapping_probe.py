"""Times orbx_lapping_knn2 before and after a large 2-NN has warmed the stream-ordered pool."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import orb_slam3_amd as orbx
from orb_slam3_amd._lib import lib, check
S = 512
a = orbx.textured_frame(800, S, S); b = np.roll(a, 9, axis=1)
exl = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_batch=128); exr = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_batch=128)
da = torch.from_numpy(np.stack([a] * 128)).cuda(); db = torch.from_numpy(np.stack([b] * 128)).cuda()
exl.extract_batch_device(da.data_ptr(), 128, S, S, S, (100, 511)); exr.extract_batch_device(db.data_ptr(), 128, S, S, S, (0, 400))
exl.sync(); exr.sync()
def timeit(tag):
    ts = []
    for i in range(40):
        t0 = time.perf_counter(); orbx.lapping_knn_match2(exl, exr, i % 32); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(tag, "min %.3f median %.3f max %.3f ms" % (ts.min(), np.median(ts), ts.max()), flush=True)
timeit("cold")
q = torch.randint(0, 256, (1 << 18, 32), dtype=torch.uint8, device="cuda"); t = torch.randint(0, 256, (1 << 18, 32), dtype=torch.uint8, device="cuda")
idx = torch.empty((1 << 18, 2), dtype=torch.int32, device="cuda"); dst = torch.empty((1 << 18, 2), dtype=torch.int16, device="cuda")
check(lib().orbx_knn2_device(C.c_void_p(q.data_ptr()), 1 << 18, C.c_void_p(t.data_ptr()), 1 << 18, 0, C.c_void_p(idx.data_ptr()), C.c_void_p(dst.data_ptr()), None))
torch.cuda.synchronize()
timeit("after a large 2-NN")
