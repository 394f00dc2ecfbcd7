#!/usr/bin/env python3
"""Instruction-class mix of the extraction kernels, from the gfx950 assembly hipcc emits.

The vector ALU of gfx950 issues two classes of instructions at different rates (measured with
tools/probe/valu_rate_probe.hip at 8 waves per SIMD, profiles/r3_valu_rate_probe.txt):
  fast, ~1.07 ns per wave-instruction per SIMD: two-operand integer add / sub, and / or / xor /
        not, shifts, v_mov, and the f32 add / mul / fma family;
  slow, ~1.75 ns: everything else - three-operand integer forms, min / max, dot products,
        v_perm / v_alignbyte / v_bfe, 24- and 32-bit multiplies, packed 16-bit and packed f32,
        conversions, compares and v_cndmask, population counts, f64, SDWA and DPP forms.
A kernel that is bound by vector issue can therefore at best run at
  1024 SIMDs / (f_fast * 1.07 ns + f_slow * 1.75 ns) wave-instructions per second,
with f the shares of its instructions in each class.  This script counts the shares per kernel
in the assembly (every instruction once, as written: loops and rarely taken paths are not
weighted by how often they run) and writes them as JSON for bench.py.

  python tools/valu_class_mix.py [out.json]
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST_NS, SLOW_NS = 1.07, 1.75
N_SIMD = 256 * 4

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
        "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
        "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_mad_f32", "v_xnor_b32", "v_nop",
        "v_accvgpr_read_b32", "v_accvgpr_write_b32"}

KERNELS = {"k_fast_cells<48>": "k_fast_cellsILi48", "k_orient_blur_describe": "k_orient_blur_describe",
           "k_orient_describe": "k_orient_describeE", "k_pyr_level0": "k_pyr_level0", "k_pyr_resize<4>": "k_pyr_resizeILi4",
           "k_pyr_resize<2>": "k_pyr_resizeILi2", "k_pyr_resize<1>": "k_pyr_resizeILi1", "k_octree<false>": "k_octreeILb0",
           "k_blur": "k_blurE", "k_assemble": "k_assemble"}


def base_op(tok):
    op = tok
    variant = ""
    for suf in ("_sdwa", "_dpp"):
        if op.endswith(suf):
            op, variant = op[:-len(suf)], suf
    for suf in ("_e32", "_e64"):
        if op.endswith(suf):
            op = op[:-len(suf)]
    return op, variant


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r3_valu_class_mix.json")
    src = os.path.join(ROOT, "orb_slam3_amd", "csrc", "orbx_extract.hip")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "extract.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-ffp-contract=off", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-S",
                               "--cuda-device-only", src, "-o", asm], stderr=subprocess.DEVNULL)
        text = open(asm).read().splitlines()
    res = {"_meta": {"source": "hipcc -S --cuda-device-only orb_slam3_amd/csrc/orbx_extract.hip (static counts)",
                     "fast_ns": FAST_NS, "slow_ns": SLOW_NS, "rates_from": "profiles/r3_valu_rate_probe.txt (8 waves per SIMD)"}}
    cur = None
    counts = {}
    for line in text:
        m = re.match(r"^(_ZN4orbx\S+):", line)
        if m:
            cur = None
            for name, key in KERNELS.items():
                if key in m.group(1):
                    cur = name
                    counts[cur] = {"fast": 0, "slow": 0, "ops": {}}
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None:
            continue
        t = line.strip().split()
        if not t or not t[0].startswith("v_"):
            continue
        op, variant = base_op(t[0])
        cls = "fast" if (op in FAST and not variant) else "slow"
        counts[cur][cls] += 1
        counts[cur]["ops"][op + variant] = counts[cur]["ops"].get(op + variant, 0) + 1
    for name, c in counts.items():
        n = c["fast"] + c["slow"]
        if n == 0:
            continue
        ns = (c["fast"] * FAST_NS + c["slow"] * SLOW_NS) / n
        top = sorted(c["ops"].items(), key=lambda kv: -kv[1])[:12]
        res[name] = {"valu_static": n, "fast": c["fast"], "slow": c["slow"], "slow_share": c["slow"] / n,
                     "mean_ns_per_wave_inst_per_simd": ns, "peak_G_wave_insts_per_s": N_SIMD / ns,
                     "most_frequent": dict(top)}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    for k, v in res.items():
        if k != "_meta":
            print("%-26s %5d VALU  slow share %.2f  mean %.2f ns  peak %.0f G wave-insts/s" %
                  (k, v["valu_static"], v["slow_share"], v["mean_ns_per_wave_inst_per_simd"], v["peak_G_wave_insts_per_s"]))


if __name__ == "__main__":
    main()
