"""The allocation-free form of the device 2-NN (orbx_knn2_device_noalloc: per-segment partial
results in a fixed device array inside the library) against orbx_knn2_device, bit for bit, on two
calls in a row, the second with a smaller input - leftovers of the first call in the shared array
must not leak into the second."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(fn, q, t, off, torch, dev):
    nq = q.shape[0]
    idx = torch.full((max(nq, 1), 2), -7, dtype=torch.int32, device=dev)
    dst = torch.full((max(nq, 1), 2), -7, dtype=torch.int16, device=dev)
    st = fn(C.c_void_p(q.data_ptr()), nq, C.c_void_p(t.data_ptr()), t.shape[0], off,
            C.c_void_p(idx.data_ptr()), C.c_void_p(dst.data_ptr()), None)
    if dev == "cuda":
        torch.cuda.synchronize()
    return st, idx[:nq].cpu().numpy().view(np.uint32), dst[:nq].cpu().numpy().view(np.uint16)


def test_noalloc_equals_allocating_entry_point_twice():
    import torch
    from orb_slam3_amd._lib import lib, ORBX_OK
    from tests.conftest import TORCH_DEVICE as dev
    L = lib()
    rng = np.random.default_rng(77)
    # (queries, train rows): the first call is cut into many segments (few query blocks), the
    # second is smaller in both directions and still segmented
    # (the host simulation emulates every matrix instruction lane by lane: smaller sizes there,
    # still several segments per call)
    from tests.conftest import HAS_GPU
    sizes = ((3000, 60000, 0), (700, 9000, 123456)) if HAS_GPU else ((600, 12000, 0), (200, 3000, 123456))
    for nq, nt, off in sizes:
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        t[nt // 2:nt // 2 + 50] = t[:50]                 # equal rows in different segments: index tie-break
        q = t[rng.integers(0, nt, nq)].copy()
        q[::2] ^= rng.integers(0, 256, (len(q[::2]), 32), dtype=np.uint8) & 0x41
        dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
        st0, i0, d0 = _run(L.orbx_knn2_device, dq, dt, off, torch, dev)
        st1, i1, d1 = _run(L.orbx_knn2_device_noalloc, dq, dt, off, torch, dev)
        assert st0 == ORBX_OK and st1 == ORBX_OK
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
        assert (i1[:, 0] >= off).all()


def test_noalloc_rejects_what_the_array_cannot_hold():
    """More than 4 194 304 train rows need two segments at least; with enough queries their
    partial results exceed the array: a clear status, nothing launched.  (Only sizes are passed;
    the pointers are never dereferenced because the call returns before any launch.)"""
    import torch
    from orb_slam3_amd._lib import lib, ORBX_ERR_SCRATCH
    from tests.conftest import TORCH_DEVICE as dev
    L = lib()
    cap = int(L.orbx_knn2_noalloc_scratch_bytes())
    assert cap == 64 << 20
    nq = cap // (2 * 12) + 4096
    buf = torch.zeros(256, dtype=torch.uint8, device=dev)
    p = C.c_void_p(buf.data_ptr())
    assert L.orbx_knn2_device_noalloc(p, nq, p, (1 << 22) + 1, 0, p, p, None) == ORBX_ERR_SCRATCH
