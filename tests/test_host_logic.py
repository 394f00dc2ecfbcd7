"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol
declared in include/orbx.h, the std::sort restatement used by the quadtree kernel matches
libstdc++, status strings, synthetic frames, shard arithmetic."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "orbx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from orb_slam3_amd import _lib
    L = _lib.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "liborbx_hip.so does not export %s" % n
    assert set(_lib.SIGNATURES) == set(names), set(_lib.SIGNATURES) ^ set(names)


def test_status_strings_and_host_helpers():
    import orb_slam3_amd as orbx
    from orb_slam3_amd import _lib
    L = _lib.lib()
    for st in range(-8, 1):
        assert L.orbx_strerror(st) not in (None, b"unknown status")
    assert L.orbx_strerror(-99) == b"unknown status"
    a = np.arange(32, dtype=np.uint8)
    assert orbx.descriptor_distance(a, a) == 0
    assert orbx.descriptor_distance(np.zeros(32, np.uint8), np.full(32, 0x0F, np.uint8)) == 128


def test_no_gpu_means_loud_failure():
    import torch
    import orb_slam3_amd as orbx
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(orbx.OrbxError) as e:
        orbx.ORBextractor(1000, 1.2, 8, 20, 7)
    assert e.value.status == -5
    with pytest.raises(orbx.OrbxError):
        orbx.knn_match2(np.zeros((4, 32), np.uint8), np.zeros((4, 32), np.uint8))


def test_synth_frames_are_deterministic():
    import orb_slam3_amd as orbx
    a = orbx.textured_frame(5, 320, 240)
    assert np.array_equal(a, orbx.textured_frame(5, 320, 240))
    assert not np.array_equal(a, orbx.textured_frame(6, 320, 240))
    assert 40 < a.std() < 90


def test_std_sort_restatement_matches_libstdcxx(tmp_path):
    exe = str(tmp_path / "test_stdsort")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "test_stdsort.cpp")])
    out = subprocess.check_output([exe]).decode()
    assert out.startswith("OK")


def test_cv_adaptor_header_compiles():
    """include/orbx_cv.hpp against the test-only OpenCV stand-in (syntax and types only)."""
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror",
                           "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tests", "cpp", "cvstub"),
                           os.path.join(ROOT, "tests", "cpp", "test_adaptor.cpp")])


def test_shard_bounds_cover_everything_once():
    import orb_slam3_amd as orbx
    for n in (0, 1, 7, 1000, 10_000_001):
        for world in (1, 2, 4, 8):
            b = orbx.shard_bounds(n, world)
            assert b[0] == 0 and b[-1] == n and len(b) == world + 1
            assert all(b[i] <= b[i + 1] for i in range(world))
            assert all(x % 8 == 0 for x in b[:-1])


def test_launch_block_sizes_match_launch_bounds():
    """A launch with more threads than the kernel's __launch_bounds__ fails on the device and
    nowhere else (the host simulation does not know the bounds): every hipLaunchKernelGGL in the
    library names the block size of the kernel's declaration, or a smaller literal one (a kernel
    that also runs as narrower workgroups: the bounds are a maximum)."""
    import glob
    import re
    src = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "orb_slam3_amd", "csrc", "*.hip"))))
    bounds = {m.group(2): m.group(1).strip()
              for m in re.finditer(r"__global__\s+__launch_bounds__\(([^)]*)\)\s+void\s+(\w+)", src)}
    assert not re.findall(r"__global__\s+void\s+\w+", src), "every kernel declares its launch bounds"
    # launches whose block size is a variable: checked where it is computed
    # (k_octree_team<kOctTeamWaves> is launched with dim3(64, kOctTeamWaves) against bounds of 64 * NW:
    # the same constant is the template argument and the block's second dimension)
    # (k_debug_sort<NW> likewise: dim3(64, kOctTeamWaves) or dim3(64, 1) against 64 * NW)
    by_variable = {"k_pyr_level0": "256", "kern": None, "k_octree_team": None, "k_debug_sort": None}
    assert re.search(r"k_debug_sort<kOctTeamWaves>, dim3\(1\), dim3\(64, kOctTeamWaves\)", src)
    assert re.search(r"k_debug_sort<1>, dim3\(1\), dim3\(64, 1\)", src)
    assert re.search(r"k_octree_team<kOctTeamWaves>, grid, dim3\(64, kOctTeamWaves\)", src)
    seen = 0
    for m in re.finditer(r"hipLaunchKernelGGL\(\s*(?:orbx::)?(\w+)(?:<[^>]*>)?\s*,\s*(?:dim3\([^;]*?\)|[^,]+),"
                         r"\s*(?:dim3\(([^)]*)\)|([^,]+)),", src, re.S):
        kernel, block = m.group(1), (m.group(2) or m.group(3)).strip()
        seen += 1
        if kernel in by_variable:
            continue
        assert kernel in bounds, kernel
        if block.isdigit() and bounds[kernel].isdigit():
            assert int(block) % 64 == 0 and int(block) <= int(bounds[kernel]), (kernel, block, bounds[kernel])
        else:
            assert block == bounds[kernel], (kernel, block, bounds[kernel])
    assert seen >= 30


def test_header_is_plain_c():
    """The boundary is a C ABI: include/orbx.h compiles as C99 with warnings as errors."""
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "orbx.h")])


def test_no_kernel_of_the_built_library_uses_scratch_memory(tmp_path):
    """Every kernel of liborbx_hip.so keeps its state in registers and LDS: the code objects'
    metadata must give a private segment of 0 bytes per lane for each of them (a run-time index
    into a local array, or a register spill, would show up here as scratch memory)."""
    import re
    import shutil
    llvm = "/opt/rocm/lib/llvm/bin"
    so = os.path.join(ROOT, "orb_slam3_amd", "liborbx_hip.so")
    if not (os.path.exists(so) and os.path.exists(os.path.join(llvm, "llvm-objdump"))):
        pytest.skip("built library or LLVM tools not present")
    shutil.copy(so, tmp_path / "lib.so")
    subprocess.check_call([os.path.join(llvm, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp_path,
                          stdout=subprocess.DEVNULL)
    objs = sorted(p for p in os.listdir(tmp_path) if "amdgcn" in p)
    assert objs, "no device code object found in the library"
    kernels = {}
    for o in objs:
        assert o.endswith("gfx950"), o
        notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", o], cwd=tmp_path, text=True)
        name = None
        for line in notes.splitlines():
            m = re.search(r"\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name:
                kernels[name] = int(m.group(1))
    assert len(kernels) >= 40, sorted(kernels)
    assert any("k_knn2_fp4" in k for k in kernels) and any("k_fast_cells" in k for k in kernels)
    with_scratch = {k: v for k, v in kernels.items() if v != 0}
    assert not with_scratch, with_scratch
