import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_real_device: cannot run under the kernel host simulation (sizes, or a binary linked to the real library)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()

# ORBX_HOSTSIM=1: run the GPU-marked tests against tests/hostsim (the library's kernels
# compiled for the host and executed thread by thread with guarded memory).  This checks
# kernel addressing and logic where there is no GPU; it is never what the package loads by
# itself and it is not a parity result for the GPU.
HOSTSIM = os.environ.get("ORBX_HOSTSIM") == "1"


def _load_hostsim():
    import ctypes
    sim_dir = os.path.join(ROOT, "tests", "hostsim")
    # ORBX_HOSTSIM_LIB=ubsan: the build with undefined-behaviour checks compiled in as traps
    variant = os.environ.get("ORBX_HOSTSIM_LIB", "")
    subprocess.check_call(["make", "-C", sim_dir, "-j8", variant or "all"], stdout=subprocess.DEVNULL)
    from orb_slam3_amd import _lib
    handle = ctypes.CDLL(os.path.join(sim_dir, "_build", variant, "liborbx_hostsim.so"), mode=ctypes.RTLD_LOCAL)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    _lib._lib = handle


if HOSTSIM and not HAS_GPU:
    _load_hostsim()

# where tests that pass device pointers put their tensors: under the host simulation device
# memory is host memory
TORCH_DEVICE = "cuda" if HAS_GPU else "cpu"


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    if HOSTSIM:
        skip = pytest.mark.skip(reason="needs a real device")
        for item in items:
            if "gpu" in item.keywords and "needs_real_device" in item.keywords:
                item.add_marker(skip)
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liborb_oracle.so); built on demand where g++ exists."""
    so = os.path.join(ROOT, "oracle", "liborb_oracle.so")
    src = os.path.join(ROOT, "oracle", "orb_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    from tests import oracle_lib
    return oracle_lib
