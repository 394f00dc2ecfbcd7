"""Times the brute-force Hamming 2-NN kernel on device-resident descriptors."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orb_slam3_amd._lib import lib, check
for nq, nt in [(4096, 65536), (65536, 262144), (131072, 1 << 20), (1 << 20, 1250000)]:
    q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device="cuda")
    t = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device="cuda")
    idx = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    dst = torch.empty((nq, 2), dtype=torch.int16, device="cuda")
    def run():
        check(lib().orbx_knn2_device(C.c_void_p(q.data_ptr()), nq, C.c_void_p(t.data_ptr()), nt, 0,
                                     C.c_void_p(idx.data_ptr()), C.c_void_p(dst.data_ptr()), None))
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(nq, nt, "%.3f ms  %.3f Tpairs/s  (%.1f%% of 2.46 Tpairs/s VALU roof)" % (dt * 1e3, nq * nt / dt / 1e12, nq * nt / dt / 2.46e12 * 100))
