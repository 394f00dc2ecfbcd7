#!/bin/bash
# Runs "knock-out" variants of k_orient_blur_describe (the kind of timing experiment that removes a
# kernel's loads, its arithmetic or its work-list read to see what bounds it) under the kernel
# host simulation with guard pages, so that a variant that reads or writes outside its buffers is
# caught on the CPU and never reaches a GPU.  Nothing here touches the repository's sources: every
# variant is a patched copy under /tmp.  Usage: tools/knockout_check.sh  (prints one line per variant;
# exit status 1 when a patch no longer applies to the sources or a variant does not build)
set -u
FAILED=0
R=$(cd "$(dirname "$0")/.." && pwd)
CXX=/opt/rocm/lib/llvm/bin/clang++
SIM=$R/tests/hostsim
OUT=${KO_OUT:-/tmp/orbx_knockout}
mkdir -p "$OUT"
FLAGS="-std=c++17 -O2 -g -fPIC -Wno-unused-command-line-argument -ffp-contract=off -fno-strict-aliasing -I$SIM -I$R/include -w"
NAMES="orbx_extract orbx_api orbx_match orbx_dist orbx_stereo orbx_grid orbx_guided orbx_bow"

variant() {   # name, python expression list of (old, new) replacements on orbx_extract.hip
  local name=$1 edits=$2 d=$OUT/$1
  rm -rf "$d" && mkdir -p "$d/src" "$d/_build"
  cp "$R"/orb_slam3_amd/csrc/*.hip "$R"/orb_slam3_amd/csrc/*.h "$R"/orb_slam3_amd/csrc/*.inc "$d/src/"
  python3 - "$d/src/orbx_extract.hip" "$edits" <<'PY' || { echo "$name: patch did not apply"; FAILED=1; return; }
import sys
p, edits = sys.argv[1], eval(sys.argv[2])
s = open(p).read()
for old, new in edits:
    if old not in s:
        sys.exit("missing: " + old[:60])
    s = s.replace(old, new)
open(p, "w").write(s)
PY
  local objs=""
  for n in $NAMES; do
    sed -E 's/extern __shared__ (__align__\(16\) )?([a-z0-9_]+) ([A-Za-z]+)\[\];/\2* \3 = (\2*)hostsim::dyn_lds();/' "$d/src/$n.hip" > "$d/_build/$n.cpp"
    $CXX $FLAGS -I"$d/src" -c "$d/_build/$n.cpp" -o "$d/_build/$n.o" || { echo "$name: build failed"; FAILED=1; return; }
    objs="$objs $d/_build/$n.o"
  done
  $CXX $FLAGS -c "$SIM/hostsim.cpp" -o "$d/_build/hostsim.o"
  $CXX -shared -fPIC -Wl,-Bsymbolic -o "$d/_build/liborbx_hostsim.so" $objs "$d/_build/hostsim.o" -lpthread -ldl
  for guard in end start; do
    HOSTSIM_GUARD=$guard HOSTSIM_FILL=0xFF python3 - "$d/_build/liborbx_hostsim.so" > "$d/run_$guard.log" 2>&1 <<'PY'
import ctypes, sys, numpy as np
sys.path.insert(0, __import__("os").environ.get("ORBX_ROOT", "."))
from orb_slam3_amd import _lib
h = ctypes.CDLL(sys.argv[1], mode=ctypes.RTLD_LOCAL)
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(h, name); fn.restype = res; fn.argtypes = args
_lib._lib = h
import orb_slam3_amd as orbx
img = orbx.textured_frame(5, 376, 240)
ex = orbx.ORBextractor(300, 1.2, 4, 20, 7)
mono, k, d = ex(img, None, (0, 1000))
print("completed: %d keypoints" % len(k))
PY
    local rc=$?
    local what
    what=$(grep -m1 -A2 "hostsim: invalid memory access" "$d/run_$guard.log" | tr -s ' \n' ' ' || true)
    if [ $rc -eq 0 ]; then echo "$name [guard=$guard]: ran to completion, no access outside any buffer ($(tail -1 "$d/run_$guard.log"))"
    else echo "$name [guard=$guard]: STOPPED (exit $rc) ${what:-see $d/run_$guard.log}"; fi
  done
}

export ORBX_ROOT=$R
# 0. the committed kernel (control)
variant control "[]"
# 1. the staging loads of the keypoint's window removed (the window keeps whatever LDS held)
variant no_window_loads "[('      const U4A4 t = *(const U4A4*)(pbase + (size_t)r * ps + 16 * c3);\n      st[u] = make_uint4(t.x, t.y, t.z, t.w);', '      st[u] = make_uint4(0, 0, 0, 0); (void)pbase; (void)r; (void)c3;'), ('      if (idx < FW_CHUNKS) *(uint4*)&raw[4 * idx] = st[u];', '      (void)idx;')]"
# 2. the trigonometry removed: the rotation coefficients come from whatever the window holds
#    instead of cos / sin of the angle
variant no_trig "[('  sincos_small((double)rad, sind, cosd);', '  sind = (double)__uint_as_float(raw[j32]); cosd = (double)__uint_as_float(raw[j32 + 32]); (void)rad;')]"
# 3. the work-list read removed: every keypoint's window starts at byte 0 of the pyramid buffer
#    with row stride 0 (the record carries the window's address, so nothing leaves the buffer)
variant no_work_item "[('  const WorkItem wi = work[(size_t)f * plan->kp_cap + (valid ? k : n - 1)];', '  WorkItem wi; wi.level = 0; wi.xy = 0; wi.dst = valid ? k : n - 1; wi.pyr_win = 0; wi.blur_win = 0; wi.strides = 0; (void)work;')]"
# 4. the blur pass removed: the descriptor samples whatever the blurred window's LDS held
variant no_blur "[('  if (j32 < 30) {\n    const int band = j32 >= 20 ? 2 : (j32 >= 10 ? 1 : 0);', '  if (false) {\n    const int band = j32 >= 20 ? 2 : (j32 >= 10 ? 1 : 0);')]"
exit $FAILED
