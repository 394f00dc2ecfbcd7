#!/usr/bin/env python3
"""Throughput bench of the ORB front-end hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one batched extraction (pyramid, FAST cells, quadtree, orientation,
blur, descriptors) of `--batch` synthetic 752x480 frames that are already
resident in HBM; outputs stay in HBM.  One process per GPU: with --gpus N > 1
and no launcher around it, this script starts `torch.distributed.run` with N
ranks as a child process and relays its output; under a launcher (RANK /
WORLD_SIZE set) it is one of the ranks.  Frames are independent, so extraction
has no collective (weak scaling; RCCL carries only the timing barrier); the
sharded brute-force 2-NN leg is the one step with an exchange (one all-gather
of top-2 records).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 752, 480
PARAMS = (1000, 1.2, 8, 20, 7)   # nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST
LAP = (0, 1000)                  # monocular call, core/src/Frame.cc:360
HBM_PEAK_GBS = 8000.0
VALU_PEAK_LANE_OPS = 256 * 64 * 2.4e9          # lane-ops/s: 256 CUs x 64 lanes x 2.4 GHz
# wave instructions/s of the vector ALU: 4 SIMDs per CU, one wave64 instruction per 2 cycles for the
# fast class (the guide's figure); the measured per-class rates (tools/probe/valu_rate_probe.hip) give
# each kernel a peak of its own from its instruction mix (tools/valu_class_mix.py)
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 2
MFMA_I8_PEAK_OPS = 5.0e15                      # dense int8 matrix-core peak: twice the 2.5 PFLOP/s of bf16
MFMA_FP4_PEAK_OPS = 10.0e15                    # dense block-scaled FP4 peak (the figure with 2:1 sparsity is twice that)


def level_sizes(w, h, nlevels, scale_factor):
    import numpy as np
    sizes, s = [], np.float32(1.0)
    for _ in range(nlevels):
        inv = np.float32(1.0) / s
        sizes.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        s = np.float32(np.float64(s) * np.float64(np.float32(scale_factor)))
    return sizes


def algorithmic_bytes(w, h, nlevels, scale_factor, nkp):
    """Per-frame algorithmic HBM bytes of each bandwidth-bound stage (unfused minimum):
    pyramid = input + padded planes written + resize sources read; fast = padded planes
    read; blur = level images read + written; whole frame adds 60 B per keypoint."""
    sz = level_sizes(w, h, nlevels, scale_factor)
    A = [a * b for a, b in sz]
    P = [(a + 38) * (b + 38) for a, b in sz]
    stage = {"pyramid": w * h + sum(P) + sum(A[:-1]), "fast": sum(P), "blur": 2 * sum(A)}
    total = w * h + 2 * sum(P) + 2 * sum(A) + sum(A[:-1]) + nkp * 60
    return stage, total


KERNEL_OF_STAGE = {"fast": "k_fast_cells", "blur": "k_blur", "describe": "k_orient_blur_describe"}


def valu_class_peak(kernel_prefix):
    """Peak wave-instructions/s of the vector ALU for the kernel's mix of fast- and slow-class
    instructions (static counts from the assembly, profiles/r3_valu_class_mix.json)."""
    path = os.path.join(ROOT, "profiles", "r3_valu_class_mix.json")
    try:
        with open(path) as fh:
            mix = json.load(fh)
    except OSError:
        return None
    for name, r in mix.items():
        if name.startswith(kernel_prefix):
            return {"peak": r["peak_G_wave_insts_per_s"] * 1e9, "slow_share": r["slow_share"],
                    "fast_ns": mix["_meta"]["fast_ns"], "slow_ns": mix["_meta"]["slow_ns"],
                    "source": "profiles/r3_valu_class_mix.json"}
    return None


def pmc_counters(stage, batch, two_pass):
    """Per-launch counters of the stage's kernel from the newest committed rocprofv3 counter
    summary of this same command whose recorded code state matches the running one (FETCH_SIZE,
    WRITE_SIZE and the SQ counters are collected in separate runs and cannot be read from inside
    the process).  FETCH_SIZE is in KiB and, on gfx950, tallies the 128-byte requests of wide
    coalesced reads at 64 bytes, hence the factor 2; WRITE_SIZE KiB is exact.  Returns
    (record or None, path or reason)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_b%d_pmc_summary.json" % batch)),
                   key=os.path.getmtime, reverse=True)
    cands.sort(key=lambda p: os.path.basename(p), reverse=True)   # r2b > r2a > r1...
    for path in cands:
        with open(path) as fh:
            summ = json.load(fh)
        meta = summ.get("_meta", {})
        if meta.get("fast_two_pass", None) != int(two_pass):
            continue
        for name, r in summ.items():
            if name.startswith(KERNEL_OF_STAGE[stage]) and "FETCH_SIZE" in r and "WRITE_SIZE" in r:
                return r, os.path.relpath(path, ROOT)
    return None, "no counter summary under profiles/ matches this configuration"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(frames, seconds_budget=15.0):
    """The CPU oracle (a restatement of the reference extractor; the reference itself needs
    OpenCV and cannot be built here) on this host's cores, one extractor per thread,
    independent frames - the reference's thread-per-camera model (core/src/Frame.cc:151-154).
    Threads: the CPU share of this job (OMP_NUM_THREADS where the box sets it - a GPU box shows
    the whole machine's CPUs to every job - else the affinity mask)."""
    from tests import oracle_lib as O
    so = os.path.join(ROOT, "oracle", "liborb_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    ncores = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    ncores = max(1, min(ncores, len(os.sched_getaffinity(0))))
    # calibrate with one frame, then size the sample for ~seconds_budget of wall time
    ex = O.Extractor(*PARAMS)
    ex.extract(frames[0], LAP)
    t0 = time.perf_counter()
    ex.extract(frames[0], LAP)
    one = time.perf_counter() - t0
    per_thread = max(2, min(int(seconds_budget / max(one, 1e-4)), 400))
    done = [0] * ncores

    def work(tid):
        e = O.Extractor(*PARAMS)
        for i in range(per_thread):
            e.extract(frames[(tid * per_thread + i) % len(frames)], LAP)
            done[tid] += 1

    threads = [threading.Thread(target=work, args=(i,)) for i in range(ncores)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    total = sum(done)
    return {"value": total / dt, "unit": "frames/s", "cores": ncores, "kind": "port",
            "cpu_model": cpu_model(),
            "sample": "%d synthetic 752x480 frames on %d threads (%.1f s), CPU oracle "
                      "oracle/orb_oracle.cpp (-O3 -march=x86-64-v3 -ffp-contract=off), one extractor per "
                      "thread" % (total, ncores, dt),
            "note": "a port of the reference's algorithm with scalar FAST; the reference's own OpenCV "
                    "(SIMD FAST, resize, blur) would be faster, and is not installable here",
            "single_thread_ms_per_frame": one * 1e3}


def hamming_bench(torch, orbx, dev, world, rank, nq=1 << 20, nt_per_gpu=1_250_000, reps=2):
    """Brute-force Hamming 2-NN pairs/s at the largest stated configuration's per-GPU share:
    1 M queries against 1.25 M train rows per GPU (10 M rows on 8 GPUs), uniform random 256-bit
    rows, everything resident in HBM.  On one rank the plain device entry point; on N ranks the
    sharded one (queries replicated, train rows sharded, one RCCL all-gather of the top-2
    records, merge).  The scan runs on the matrix cores (a pair = a 256-deep dot product of 0 / 1 by
    +1 / -1 values, 512 operations) in the block-scaled FP4 form of the instruction, so its roofline
    is the dense FP4 MFMA peak; the int8 form (ORBX_KNN2_I8=1) is priced against the int8 peak, and
    the vector-ALU roof the xor + popcount form (ORBX_KNN2_VALU=1) is bound by (16 lane-ops per
    pair) is quoted beside them."""
    import ctypes as C
    from orb_slam3_amd._lib import lib, check
    g = torch.Generator(device=dev).manual_seed(1234)
    q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev, generator=g)
    g.manual_seed(99 + rank)
    t = torch.randint(0, 256, (nt_per_gpu, 32), dtype=torch.uint8, device=dev, generator=g)
    offset = rank * nt_per_gpu
    stream = torch.cuda.current_stream(dev).cuda_stream
    sharded = world > 1 or os.environ.get("ORBX_BENCH_FORCE_SHARDED") == "1"   # the knob: one-rank communicator
    if world > 1:
        import torch.distributed as dist
    if sharded:
        def run():
            return orbx.knn_match2_sharded(q, t, offset)
    else:
        idx = torch.empty((nq, 2), dtype=torch.int32, device=dev)
        dst = torch.empty((nq, 2), dtype=torch.int16, device=dev)

        def run():
            check(lib().orbx_knn2_device(C.c_void_p(q.data_ptr()), nq, C.c_void_p(t.data_ptr()), nt_per_gpu,
                                         offset, C.c_void_p(idx.data_ptr()), C.c_void_p(dst.data_ptr()),
                                         C.c_void_p(stream)))
    run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / reps
    if world > 1:
        tt = torch.tensor([sec], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sec = float(tt.item())
    pairs = nq * nt_per_gpu * world / sec
    out = {"metric": "Hamming pairs/s (brute-force 2-NN, train rows sharded over the GPUs)",
           "value": pairs, "unit": "pairs/s", "queries": nq, "train_rows_total": nt_per_gpu * world,
           "train_rows_per_gpu": nt_per_gpu, "ms": sec * 1e3, "n_gpus": world,
           "roofline": {"bound": "mfma", "achieved": pairs * 512 / 1e12, "peak": MFMA_FP4_PEAK_OPS * world / 1e12,
                        "unit": "TOP/s (FP4)", "frac": pairs * 512 / (MFMA_FP4_PEAK_OPS * world),
                        "kernel": "k_knn2_fp4 (v_mfma_scale_f32_32x32x64_f8f6f4, both operands e2m1, unit scales)",
                        "frac_of_measured_instruction_rate": pairs / (17.3e12 * world),
                        "measured_instruction_rate": "17.3 T pairs/s: back-to-back instructions, two waves per SIMD "
                                                     "(tools/probe/fp4_hamming_probe.hip, profiles/r3_fp4_hamming_probe.txt)",
                        "rate_of_the_kernels_instruction_pattern": "11.5 T pairs/s in the same probe: two accumulators per wave "
                                                                   "restarted every four instructions, rotating random operands, a "
                                                                   "16-value min3 chain per accumulator, no memory, four waves per SIMD",
                        "times_the_int8_peak": pairs * 512 / (MFMA_I8_PEAK_OPS * world),
                        "times_the_vector_alu_roof": pairs * 16 / (VALU_PEAK_LANE_OPS * world)}}
    if os.environ.get("ORBX_KNN2_I8") == "1":
        out["roofline"] = {"bound": "mfma", "achieved": pairs * 512 / 1e12, "peak": MFMA_I8_PEAK_OPS * world / 1e12,
                           "unit": "TOP/s (int8)", "frac": pairs * 512 / (MFMA_I8_PEAK_OPS * world),
                           "kernel": "k_knn2_mfma (v_mfma_i32_32x32x32_i8)"}
    if os.environ.get("ORBX_KNN2_VALU") == "1":
        out["roofline"] = {"bound": "valu", "achieved": pairs * 16 / 1e12, "peak": VALU_PEAK_LANE_OPS * world / 1e12,
                           "unit": "T lane-ops/s", "frac": pairs * 16 / (VALU_PEAK_LANE_OPS * world),
                           "kernel": "k_knn2 (xor + popcount)"}
    if sharded:
        out["exchange"] = {"entry_point": "orbx_knn2_sharded", "collective": "ncclAllGather (in place)",
                           "bytes_per_rank_sent": nq * 12, "bytes_per_rank_received": nq * 12 * world}
    return out


def with_deadline(fn, seconds, dev, torch):
    """fn() on a worker thread with a deadline.  A collective that never completes (a rank that
    died, a fabric problem) must not take the headline line with it: after the deadline the caller
    reports the leg as timed out and the process later leaves through os._exit."""
    box = {}

    def work():
        try:
            torch.cuda.set_device(dev)          # the current device is per thread
            box["result"] = fn()
        except Exception as e:                  # reported in the line, not raised
            box["result"] = {"error": "%s: %s" % (type(e).__name__, e)}

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return {"error": "no result after %d s" % seconds}, True
    return box["result"], False


def host_path_bench(torch, orbx, dev, local_rank, frames, n_frames=2048):
    """The interval the reference instruments as "ORB ext" (core/src/Frame.cc:147-161): images in
    host memory -> keypoints and descriptors in host memory, PCIe both ways included.  Batches
    through orbx_extract_batch with ordinary (pageable) and with page-locked caller memory, and the
    blocking single-frame call orbx_extract.  Never the headline value."""
    import ctypes as C
    import numpy as np
    from orb_slam3_amd._lib import lib, check, KEYPOINT_DTYPE
    ex = orbx.ORBextractor(*PARAMS, device=local_rank, max_batch=n_frames)
    cap = ex.max_keypoints()
    i32p = C.POINTER(C.c_int32)
    n = np.zeros(n_frames, np.int32)
    mono = np.zeros(n_frames, np.int32)
    reps = (n_frames + len(frames) - 1) // len(frames)
    src = np.tile(frames, (reps, 1, 1))[:n_frames]
    out = {"frames_per_call": n_frames,
           "interval": "host images in -> host keypoints + descriptors out (PCIe included), core/src/Frame.cc:147-161"}

    def timed(imgs, kps, desc, label):
        ptrs = (C.c_void_p * n_frames)(*[imgs[i].ctypes.data for i in range(n_frames)])

        def call():
            check(lib().orbx_extract_batch(ex._h, n_frames, ptrs, W, H, W, LAP[0], LAP[1],
                                           kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap,
                                           n.ctypes.data_as(i32p), mono.ctypes.data_as(i32p)), "orbx_extract_batch")
        call()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            call()
            best = min(best, time.perf_counter() - t0)
        out[label] = {"frames_per_s": n_frames / best, "ms_per_call": best * 1e3,
                      "keypoints_per_frame": float(n.mean())}

    # ordinary memory both ways
    kps = np.zeros((n_frames, cap), KEYPOINT_DTYPE)
    desc = np.zeros((n_frames, cap, 32), np.uint8)
    timed(np.ascontiguousarray(src), kps, desc, "pageable")
    # page-locked memory both ways
    pin_i = torch.empty((n_frames, H, W), dtype=torch.uint8).pin_memory()
    pin_k = torch.empty((n_frames, cap * 28), dtype=torch.uint8).pin_memory()
    pin_d = torch.empty((n_frames, cap, 32), dtype=torch.uint8).pin_memory()
    pi = pin_i.numpy()
    pi[:] = src
    timed(pi, pin_k.numpy(), pin_d.numpy(), "pinned")
    # single-frame blocking call (the drop-in SLAM caller)
    ex.close()
    ex1 = orbx.ORBextractor(*PARAMS, device=local_rank)
    k1 = np.zeros(cap, KEYPOINT_DTYPE)
    d1 = np.zeros((cap, 32), np.uint8)
    n1, m1 = C.c_int32(), C.c_int32()
    img = np.ascontiguousarray(frames[0])

    def one():
        check(lib().orbx_extract(ex1._h, C.c_void_p(img.ctypes.data), W, H, W, LAP[0], LAP[1],
                                 k1.ctypes.data_as(C.c_void_p), d1.ctypes.data_as(C.c_void_p), cap,
                                 C.byref(n1), C.byref(m1)), "orbx_extract")
    for _ in range(30):
        one()
    per_call = []
    for _ in range(300):
        t0 = time.perf_counter()
        one()
        per_call.append(time.perf_counter() - t0)
    # the median: a loop of blocking calls from Python now and then contains one stall of tens of
    # milliseconds between two calls (seen as a whole loop's mean doubling while the time spent inside
    # the library stays the same); the mean is kept beside it
    out["single_frame_ms"] = float(np.median(per_call)) * 1e3
    out["single_frame_ms_mean"] = float(np.mean(per_call)) * 1e3
    ex1.close()
    return out


def stereo_bench(torch, orbx, dev, local_rank, batch=128, reps=5):
    """Secondary metric: rectified stereo pairs/s (config: 752x480, 1200 kp per camera,
    vLappingArea (0,0)): both extractions plus the device-resident ComputeStereoMatches
    (band search, L1 refinement, depth, median rule), images resident in HBM."""
    import numpy as np
    params = (1200, 1.2, 8, 20, 7)
    distinct = 8
    lefts = [orbx.textured_frame(500 + i, W, H) for i in range(distinct)]
    rights = [orbx.right_view(l, 500 + i) for i, l in enumerate(lefts)]
    reps_t = (batch + distinct - 1) // distinct
    dl = torch.from_numpy(np.tile(np.stack(lefts), (reps_t, 1, 1))[:batch].copy()).to(dev)
    dr = torch.from_numpy(np.tile(np.stack(rights), (reps_t, 1, 1))[:batch].copy()).to(dev)
    exl = orbx.ORBextractor(*params, device=local_rank, max_batch=batch)
    exr = orbx.ORBextractor(*params, device=local_rank, max_batch=batch)
    mbf, mb = 435.2 * 0.11, 0.11

    def run():
        exl.extract_batch_device(dl.data_ptr(), batch, W, H, W, (0, 0))
        exr.extract_batch_device(dr.data_ptr(), batch, W, H, W, (0, 0))
        orbx.compute_stereo_matches_batch(exl, exr, batch, mbf, mb)
    run()
    exl.sync()
    exr.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    exl.sync()
    exr.sync()
    sec = (time.perf_counter() - t0) / reps
    # stereo stage alone (extractions already done)
    exl.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        orbx.compute_stereo_matches_batch(exl, exr, batch, mbf, mb)
    exl.sync()
    sec_match = (time.perf_counter() - t0) / reps
    nmatch = float(np.mean([(orbx.stereo_download(exl, f)[1] > 0).sum() for f in range(min(batch, 8))]))
    res = {"metric": "stereo pairs/s @752x480, 2x1200kp, extraction + ComputeStereoMatches",
           "value": batch / sec, "unit": "pairs/s", "pairs_per_step": batch, "ms_per_step": sec * 1e3,
           "match_stage_ms_per_step": sec_match * 1e3, "matches_per_pair": nmatch}
    exl.close()
    exr.close()
    return res


def tracking_calls_bench(orbx, local_rank, reps=200):
    """Blocking latency (median per call, ms) of the calls a tracking thread makes per frame, on
    752x480 frames with 1000 keypoints: a stereo pair through orbx_extract_batch, the association of
    Frame::ComputeStereoMatches, the frame grid, SearchByProjection(CurrentFrame, LastFrame) and
    SearchByProjection(F, map points) (core/lib/ORB/matcher.cc:1591-1808, 41-142)."""
    import ctypes as C
    import numpy as np
    from orb_slam3_amd import grid, guided, matcher
    from orb_slam3_amd._lib import lib, check
    shift = 6
    wide = orbx.textured_frame(41, W + shift, H)
    a = np.ascontiguousarray(wide[:, shift:])
    b = np.ascontiguousarray(wide[:, :W])
    exa = orbx.ORBextractor(*PARAMS, device=local_rank)
    exb = orbx.ORBextractor(*PARAMS, device=local_rank)
    _, ka, da = exa(a, None, (0, 0))
    _, kb, db = exb(b, None, (0, 0))
    bounds = grid.image_bounds(None, W, H)
    rng = np.random.default_rng(5)
    nl = len(ka)
    uv = np.stack([ka["x"] + shift + rng.uniform(-2, 2, nl), ka["y"] + rng.uniform(-2, 2, nl)], 1).astype(np.float32)
    invz = rng.uniform(0.05, 1.0, nl).astype(np.float32)
    ones = np.ones(nl, np.uint8)
    state = np.zeros(len(kb), np.uint8)

    def med(fn):
        for _ in range(20):
            fn()
        per = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            per.append(time.perf_counter() - t0)
        return float(np.median(per)) * 1e3

    out = {"unit": "ms per blocking call, median of %d" % reps, "keypoints": int(len(kb))}
    pair = orbx.ORBextractor(*PARAMS, device=local_rank, max_batch=2)
    cap = pair.max_keypoints()
    two = np.stack([a, b])
    ptrs = (C.c_void_p * 2)(two[0].ctypes.data, two[1].ctypes.data)
    k2 = np.zeros((2, cap * 28), np.uint8)
    d2 = np.zeros((2, cap * 32), np.uint8)
    n2 = np.zeros(2, np.int32)
    m2 = np.zeros(2, np.int32)
    i32p = C.POINTER(C.c_int32)
    out["extract_pair"] = med(lambda: check(lib().orbx_extract_batch(
        pair._h, 2, ptrs, W, H, W, 0, 0, k2.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p), cap,
        n2.ctypes.data_as(i32p), m2.ctypes.data_as(i32p)), "orbx_extract_batch"))
    out["stereo_match"] = med(lambda: matcher.compute_stereo_matches(exb, exa, 47.9, 0.11))
    out["frame_grid"] = med(lambda: (grid.frame_grid_batch(exb, 1, None, bounds), exb.sync()))
    grid.frame_grid_batch(exb, 1, None, bounds)
    sbp = lambda: guided.search_by_projection_last(exb, 0, ones, uv, invz, ka["octave"], ka["angle"], da, ones, None,
                                                   state, 15.0, 0, 0, 40.0, True)
    out["search_by_projection_last"] = med(sbp)
    out["search_by_projection_last_matches"] = int((sbp()[1] >= 0).sum())
    npts = 2 * nl
    proj = np.concatenate([uv, uv + 0.7]).astype(np.float32)
    lvl = np.concatenate([ka["octave"], ka["octave"]])
    dd = np.concatenate([da, da])
    pts = lambda: guided.search_by_projection_points(exb, 0, np.ones(npts, np.uint8), proj, np.full(npts, -1, np.float32),
                                                     lvl, np.full(npts, 0.999, np.float32), dd, np.ones(npts, np.uint8),
                                                     None, state, 1.0, 0.8)
    out["search_by_projection_points_%d" % npts] = med(pts)
    for e in (exa, exb, pair):
        e.close()
    return out


def fisheye_bench(torch, orbx, dev, local_rank, batch=128, reps=5):
    """Secondary metric: fisheye stereo pairs/s (config: 512x512 pair, 1500 kp per camera, lapping
    areas): both extractions for the whole batch, then per pair the brute-force 2-NN of the left
    lapping-area descriptors against the right ones (core/src/Frame.cc:1159-1177), indices and
    distances back on the host."""
    import numpy as np
    params = (1500, 1.2, 8, 20, 7)
    S, distinct = 512, 8
    lefts = [orbx.textured_frame(800 + i, S, S) for i in range(distinct)]
    rights = [np.roll(l, 9, axis=1) for l in lefts]
    reps_t = (batch + distinct - 1) // distinct
    dl = torch.from_numpy(np.tile(np.stack(lefts), (reps_t, 1, 1))[:batch].copy()).to(dev)
    dr = torch.from_numpy(np.tile(np.stack(rights), (reps_t, 1, 1))[:batch].copy()).to(dev)
    exl = orbx.ORBextractor(*params, device=local_rank, max_batch=batch)
    exr = orbx.ORBextractor(*params, device=local_rank, max_batch=batch)
    lap_l, lap_r = (100, 511), (0, 400)

    def extract():
        exl.extract_batch_device(dl.data_ptr(), batch, S, S, S, lap_l)
        exr.extract_batch_device(dr.data_ptr(), batch, S, S, S, lap_r)
    extract()
    exl.sync()
    exr.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        extract()
    exl.sync()
    exr.sync()
    sec_ext = (time.perf_counter() - t0) / reps
    nm = min(batch, 32)
    orbx.lapping_knn_match2(exl, exr, 0)   # the first call on a device sets up the stream-ordered pool
    t0 = time.perf_counter()
    nq = 0
    for f in range(nm):
        idx, dist, ml, mr = orbx.lapping_knn_match2(exl, exr, f)
        nq += len(idx)
    sec_knn = (time.perf_counter() - t0) / nm
    # the whole batch's lapping-area 2-NN in one launch, results resident on the device
    orbx.lapping_knn_match2_batch(exl, exr, batch)
    exl.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        orbx.lapping_knn_match2_batch(exl, exr, batch)
    exl.sync()
    sec_knn_batch = (time.perf_counter() - t0) / reps
    res = {"metric": "fisheye stereo pairs/s @512x512, 2x1500kp, lapping areas: extraction + lapping-area 2-NN, "
                     "both batched, results resident on the device",
           "value": batch / (sec_ext + sec_knn_batch), "unit": "pairs/s",
           "extraction_ms_per_pair": sec_ext / batch * 1e3,
           "lapping_knn2_batched_ms_per_pair": sec_knn_batch / batch * 1e3,
           "lapping_knn2_blocking_call_ms_per_pair": sec_knn * 1e3,
           "pairs_per_s_with_blocking_per_pair_calls": 1.0 / (sec_ext / batch + sec_knn),
           "lapping_queries_per_pair": nq / nm}
    exl.close()
    exr.close()
    return res


def kitti_bench(torch, orbx, dev, local_rank, batch=256, reps=5):
    """Secondary metric: frames/s at 1241x376, 2000 kp, 8 levels (the batched stereo configuration's
    per-camera extraction), images resident in HBM."""
    import numpy as np
    w, h, distinct = 1241, 376, 8
    params = (2000, 1.2, 8, 20, 7)
    imgs = np.stack([orbx.textured_frame(300 + i, w, h) for i in range(distinct)])
    d = torch.from_numpy(np.tile(imgs, ((batch + distinct - 1) // distinct, 1, 1))[:batch].copy()).to(dev)
    ex = orbx.ORBextractor(*params, device=local_rank, max_batch=batch)
    ex.set_pipelined(True)
    ex.extract_batch_device(d.data_ptr(), batch, w, h, w, (0, 0))
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ex.extract_batch_device(d.data_ptr(), batch, w, h, w, (0, 0))
    ex.sync()
    sec = (time.perf_counter() - t0) / reps
    nkp = float(np.mean([len(ex.download(f)[1]) for f in range(4)]))
    _, frame_bytes = algorithmic_bytes(w, h, params[2], params[1], int(round(nkp)))
    res = {"metric": "ORB frames/s @1241x376,2000kp,8lvl", "value": batch / sec, "unit": "frames/s",
           "frames_per_step": batch, "ms_per_step": sec * 1e3, "keypoints_per_frame": nkp,
           "whole_frame_roofline_frac": batch / sec * frame_bytes / (HBM_PEAK_GBS * 1e9)}
    ex.close()
    return res


def kitti_stereo_bench(torch, orbx, dev, local_rank, batch=128, reps=5):
    """Secondary metric: 1241x376 stereo pairs/s, 2000 kp per camera: what Frame::Frame(imLeft,
    imRight, ...) does per pair (core/src/Frame.cc:120-181) - both extractions and
    ComputeStereoMatches (band search, L1 refinement, depth, median rule) - with the images resident
    in HBM and everything staying on the device."""
    import numpy as np
    w, h, distinct = 1241, 376, 8
    params = (2000, 1.2, 8, 20, 7)
    lefts = [orbx.textured_frame(600 + i, w, h) for i in range(distinct)]
    rights = [orbx.right_view(l, 600 + i) for i, l in enumerate(lefts)]
    reps_t = (batch + distinct - 1) // distinct
    dl = torch.from_numpy(np.tile(np.stack(lefts), (reps_t, 1, 1))[:batch].copy()).to(dev)
    dr = torch.from_numpy(np.tile(np.stack(rights), (reps_t, 1, 1))[:batch].copy()).to(dev)
    exl = orbx.ORBextractor(*params, device=local_rank, max_batch=batch)
    exr = orbx.ORBextractor(*params, device=local_rank, max_batch=batch)
    mbf, mb = 718.856 * 0.54, 0.54     # KITTI-like focal length and baseline

    def run():
        exl.extract_batch_device(dl.data_ptr(), batch, w, h, w, (0, 0))
        exr.extract_batch_device(dr.data_ptr(), batch, w, h, w, (0, 0))
        orbx.compute_stereo_matches_batch(exl, exr, batch, mbf, mb)
    run()
    exl.sync()
    exr.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    exl.sync()
    exr.sync()
    sec = (time.perf_counter() - t0) / reps
    nmatch = float(np.mean([(orbx.stereo_download(exl, f)[1] > 0).sum() for f in range(min(batch, 8))]))
    nkp = float(np.mean([len(exl.download(f)[1]) for f in range(4)]))
    _, frame_bytes = algorithmic_bytes(w, h, params[2], params[1], int(round(nkp)))
    res = {"metric": "stereo pairs/s @1241x376, 2x2000kp, extraction x2 + ComputeStereoMatches",
           "value": batch / sec, "unit": "pairs/s", "pairs_per_step": batch, "ms_per_step": sec * 1e3,
           "matches_per_pair": nmatch, "keypoints_per_frame": nkp,
           "whole_frame_roofline_frac_extraction_only": 2 * batch / sec * frame_bytes / (HBM_PEAK_GBS * 1e9)}
    exl.close()
    exr.close()
    return res


def uhd_bench(torch, orbx, dev, local_rank, batch=128, reps=3):
    """Secondary metric: frames/s at 3840x2160, 10000 kp, 8 levels (the largest configuration's
    extraction half), batches resident in HBM.  Whole-frame fraction against the algorithmic bytes
    of that size (138.8 MB per frame).  The quadtree of a 4K level is one long chain per frame
    (about a millisecond whatever the batch), so the batch is large enough to hide it: 32 frames
    per step give 13.7 k frames/s, 64 17.6 k, 128 21.2 k."""
    import numpy as np
    w, h, distinct = 3840, 2160, 4
    params = (10000, 1.2, 8, 20, 7)
    imgs = np.stack([orbx.textured_frame(7000 + i, w, h) for i in range(distinct)])
    d = torch.from_numpy(np.tile(imgs, ((batch + distinct - 1) // distinct, 1, 1))[:batch].copy()).to(dev)
    ex = orbx.ORBextractor(*params, device=local_rank, max_batch=batch)
    ex.set_pipelined(True)
    ex.extract_batch_device(d.data_ptr(), batch, w, h, w, (0, 0))
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ex.extract_batch_device(d.data_ptr(), batch, w, h, w, (0, 0))
    ex.sync()
    sec = (time.perf_counter() - t0) / reps
    nkp = float(np.mean([len(ex.download(f)[1]) for f in range(2)]))
    _, frame_bytes = algorithmic_bytes(w, h, params[2], params[1], int(round(nkp)))
    res = {"metric": "ORB frames/s @3840x2160,10000kp,8lvl", "value": batch / sec, "unit": "frames/s",
           "frames_per_step": batch, "ms_per_step": sec * 1e3, "keypoints_per_frame": nkp,
           "algorithmic_bytes_per_frame": frame_bytes,
           "whole_frame_roofline_frac": batch / sec * frame_bytes / (HBM_PEAK_GBS * 1e9)}
    ex.close()
    return res


def _bench_vocabulary(orbx, local_rank, k=10, depth=5):
    """Synthetic vocabulary of k^depth words (children = bit-flipped copies of their parent)."""
    from tests import voc_util
    arrays = voc_util.make_vocabulary(k, depth, seed=1)
    return orbx.bow.ORBVocabulary.from_arrays(k, depth, *arrays, device=local_rank), arrays


def bow_bench(torch, orbx, dev, local_rank, batch=256, reps=5, k=10, depth=5):
    """Secondary metric (opt-in): frames/s of Frame::ComputeBoW on device-resident descriptors."""
    import numpy as np
    voc, _ = _bench_vocabulary(orbx, local_rank, k, depth)
    distinct = 8
    imgs = np.stack([orbx.textured_frame(700 + i, W, H) for i in range(distinct)])
    d = torch.from_numpy(np.tile(imgs, ((batch + distinct - 1) // distinct, 1, 1))[:batch].copy()).to(dev)
    ex = orbx.ORBextractor(*PARAMS, device=local_rank, max_batch=batch)
    ex.extract_batch_device(d.data_ptr(), batch, W, H, W, (0, 1000))
    orbx.bow.transform_batch(ex, voc, batch, 4)
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        orbx.bow.transform_batch(ex, voc, batch, 4)
    ex.sync()
    sec = (time.perf_counter() - t0) / reps
    words = float(np.mean([len(orbx.bow.download(ex, f)[0]) for f in range(min(batch, 4))]))
    return {"metric": "ComputeBoW frames/s (transform only, descriptors resident)", "value": batch / sec,
            "unit": "frames/s", "frames_per_step": batch, "ms_per_step": sec * 1e3,
            "vocabulary": {"k": k, "L": depth, "nodes": voc.info()[2], "words": voc.info()[3]},
            "words_per_frame": words}


def guided_bench(torch, orbx, dev, local_rank, reps=20):
    """Secondary metric (opt-in): latency of one SearchByProjection(CurrentFrame, LastFrame) call
    (host arrays in, assignments out) on a 752x480 frame with 1000 features, last frame = the same
    scene shifted by 6 pixels."""
    import numpy as np
    shift = 6
    wide = orbx.textured_frame(41, W + shift, H)
    a, b = np.ascontiguousarray(wide[:, shift:]), np.ascontiguousarray(wide[:, :W])
    exa = orbx.ORBextractor(*PARAMS, device=local_rank)
    _, ka, da = exa(a, None, (0, 0))
    exb = orbx.ORBextractor(*PARAMS, device=local_rank)
    exb(b, None, (0, 0))
    orbx.grid.frame_grid_batch(exb, 1, None, orbx.grid.image_bounds(None, W, H))
    rng = np.random.default_rng(0)
    n_last = len(ka)
    uv = np.stack([ka["x"] + shift + rng.uniform(-2, 2, n_last), ka["y"] + rng.uniform(-2, 2, n_last)], 1)
    args = (exb, 0, np.ones(n_last, np.uint8), uv.astype(np.float32), np.full(n_last, 0.5, np.float32),
            ka["octave"], ka["angle"], da, np.ones(n_last, np.uint8))
    nm, _, rounds = orbx.guided.search_by_projection_last(*args, th=15.0)
    t0 = time.perf_counter()
    for _ in range(reps):
        orbx.guided.search_by_projection_last(*args, th=15.0)
    sec = (time.perf_counter() - t0) / reps
    # the local-map matcher on three map points per keypoint, and SearchByBoW against frame a
    kp3, dp3 = np.concatenate([ka] * 3), np.concatenate([da] * 3)
    n3 = len(kp3)
    proj = np.stack([kp3["x"] + shift + rng.uniform(-3, 3, n3), kp3["y"] + rng.uniform(-3, 3, n3)], 1).astype(np.float32)
    pargs = (exb, 0, np.ones(n3, np.uint8), proj, proj[:, 0] - 5, kp3["octave"], np.full(n3, 0.9995, np.float32), dp3,
             np.ones(n3, np.uint8))
    nm3, _, rounds3 = orbx.guided.search_by_projection_points(*pargs, th=3.0)
    t0 = time.perf_counter()
    for _ in range(reps):
        orbx.guided.search_by_projection_points(*pargs, th=3.0)
    sec3 = (time.perf_counter() - t0) / reps
    voc, _ = _bench_vocabulary(orbx, local_rank)
    orbx.bow.transform_batch(exa, voc, 1, 4)
    _, _, an, as_, af = orbx.bow.download(exa, 0)
    orbx.bow.transform_batch(exb, voc, 1, 4)
    bargs = (exb, 0, np.ones(len(ka), np.uint8), ka["angle"], da, an, as_, af)
    nmb, _ = orbx.guided.search_by_bow(*bargs)
    t0 = time.perf_counter()
    for _ in range(reps):
        orbx.guided.search_by_bow(*bargs)
    secb = (time.perf_counter() - t0) / reps
    return {"metric": "SearchByProjection(Frame, LastFrame) ms per call (host arrays in and out)",
            "value": sec * 1e3, "unit": "ms", "higher_is_better": False, "queries": n_last, "matches": nm,
            "rounds": rounds,
            "local_map": {"ms": sec3 * 1e3, "map_points": n3, "matches": nm3, "rounds": rounds3},
            "by_bow": {"ms": secb * 1e3, "keyframe_features": len(ka), "nodes": len(an), "matches": nmb}}


def spawn_ranks(n):
    """Not under a launcher: start one rank per GPU with torch.distributed.run as a child process
    (this process has not touched torch or HIP) and exit with its status."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dry_run_cpu(args):
    """Launcher plumbing without a GPU (tests): rendezvous over gloo, the barrier and the
    max-over-ranks reduction of the timed loop, the shard layout of the 2-NN leg, and rank 0's
    JSON line.  No part of the hot path runs, so the line carries no value."""
    import torch
    import torch.distributed as dist
    from orb_slam3_amd.matcher import shard_bounds
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    b = shard_bounds(10_000_000, world)
    if rank == 0:
        print(json.dumps({"metric": "ORB frames/s @752x480,1000kp,8lvl", "value": None, "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "max_over_ranks_s": float(t.item()), "shard_bounds": b}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per step per GPU")
    ap.add_argument("--no-match", action="store_true", help="skip the Hamming 2-NN leg")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the secondary legs (host path, stereo, fisheye, 1241x376)")
    ap.add_argument("--stereo", action="store_true", help="kept for older command lines: the stereo leg is on by default")
    ap.add_argument("--bow", action="store_true", help="add the bag-of-words leg (frames/s of ComputeBoW)")
    ap.add_argument("--guided", action="store_true", help="add the guided-matching leg (ms per SearchByProjection call)")
    ap.add_argument("--distinct", type=int, default=32, help="distinct synthetic frames (tiled to the batch)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--stage-times", action="store_true",
                    help="kept for older command lines: the serialised stage timing is on by default")
    ap.add_argument("--no-stage-times", action="store_true",
                    help="skip timing every stage serialised on one stream (whole batch per launch)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="launcher plumbing only, over gloo, without a GPU (used by the tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if args.dry_run_cpu:
        return dry_run_cpu(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    import orb_slam3_amd as orbx

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    B = args.batch
    frames = np.stack([orbx.textured_frame(1000 * rank + i, W, H) for i in range(args.distinct)])
    reps = (B + args.distinct - 1) // args.distinct
    d_imgs = torch.from_numpy(np.tile(frames, (reps, 1, 1))[:B].copy()).to(dev)
    torch.cuda.synchronize()

    ex = orbx.ORBextractor(*PARAMS, device=local_rank, max_batch=B)
    # back-to-back steps as two independent half-batch pipelines (nothing is ordered behind the
    # handle's stream between the steps here; the default joins the halves after every call)
    ex.set_pipelined(True)

    def step():
        ex.extract_batch_device(d_imgs.data_ptr(), B, W, H, W, LAP)

    def barrier():
        if world > 1:
            dist.barrier()
        ex.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ex.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    barrier()

    # the dominant kernel (per-cell FAST detection) timed live with HIP events on the streams its
    # launches run on: the launches of the timed loop itself (the library keeps the events of the
    # last 32 batched calls, so reading them here puts no wait between the steps)
    launches = ex.detect_launch_times(last_calls=min(args.steps, 32))
    n_launches = len(launches)
    det_ms_per_launch = sum(ms for ms, _ in launches) / n_launches
    det_frames_per_launch = sum(fr for _, fr in launches) / n_launches

    # every stage serialised on one stream, whole batch per launch (each kernel alone on the chip)
    acc = {}
    if not args.no_stage_times:
        ex.set_profiling(True)
        nprof = max(3, min(args.steps, 10))
        for _ in range(nprof):
            step()
            ex.sync()
            for k, v in ex.stage_times_ms().items():
                acc[k] = acc.get(k, 0.0) + v / nprof
        ex.set_profiling(False)

    # keypoints per frame (for the algorithmic byte count and as a sanity figure)
    step()
    ex.sync()
    nkp = float(np.mean([len(ex.download(f)[1]) for f in range(min(B, 8))]))

    out = None
    if rank == 0:
        stage_bytes, frame_bytes = algorithmic_bytes(W, H, PARAMS[2], PARAMS[1], int(round(nkp)))
        fps = world * B * args.steps / dt
        dom = "fast"
        bytes_per_launch = stage_bytes[dom] * det_frames_per_launch
        achieved = bytes_per_launch / (det_ms_per_launch * 1e-3) / 1e9
        two_pass = os.environ.get("ORBX_FAST_SINGLE_PASS", "0") != "1"
        pmc, pmc_src = pmc_counters(dom, B, two_pass)
        traffic = pmc["FETCH_SIZE"] * 2048.0 + pmc["WRITE_SIZE"] * 1024.0 if pmc else None
        alone = None
        if acc.get(dom):
            a_bytes = stage_bytes[dom] * B
            a_rate = a_bytes / (acc[dom] * 1e-3) / 1e9
            alone = {"launch_ms": acc[dom], "frames_per_launch": B, "achieved": a_rate, "unit": "GB/s",
                     "frac": a_rate / HBM_PEAK_GBS}
        valu = None
        if pmc and "SQ_INSTS_VALU" in pmc:
            rate = pmc["SQ_INSTS_VALU"] / (det_ms_per_launch * 1e-3)
            cls = valu_class_peak(KERNEL_OF_STAGE[dom])
            peak = cls["peak"] if cls else VALU_PEAK_WAVE_INSTS
            valu = {"wave_insts_per_launch": pmc["SQ_INSTS_VALU"], "insts_per_cell_wave": pmc.get("valu_per_wave"),
                    "achieved": rate / 1e9, "peak": peak / 1e9, "unit": "G wave-insts/s",
                    "frac": rate / peak,
                    "peak_is": "class-weighted: 1024 SIMDs / (share_fast * fast_ns + share_slow * slow_ns)" if cls
                               else "one wave64 instruction per 2 cycles per SIMD",
                    "class_mix": cls,
                    "frac_of_2_cycle_peak": rate / VALU_PEAK_WAVE_INSTS}
            if alone:   # counters are per 512-frame launch; the serialised launch covers the whole batch
                r_alone = pmc["SQ_INSTS_VALU"] * (B / det_frames_per_launch) / (alone["launch_ms"] * 1e-3)
                valu["frac_alone"] = r_alone / peak
        out = {
            "metric": "ORB frames/s @752x480,1000kp,8lvl",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "EuRoC-shaped monocular 752x480, ORBextractor 1000 kp / 8 levels / "
                                   "scale 1.2 / FAST 20,7, vLappingArea (0,1000); inputs and outputs "
                                   "resident in HBM", "frames_per_step_per_gpu": B,
                       "keypoints_per_frame": nkp, "parallelism": "frames sharded, no collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": pmc_src,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "algorithmic_bytes_per_frame": stage_bytes[dom],
                         "launch_ms": det_ms_per_launch, "frames_per_launch": det_frames_per_launch,
                         "launches_timed": n_launches,
                         "schedule": "the launches of the timed steps themselves (HIP events kept per call, read "
                                     "after the loop): a batch runs as two parts on two streams, so a detection "
                                     "launch shares the chip with the other part's descriptor / pyramid / quadtree "
                                     "kernels and its duration includes that sharing; 'alone' = the same kernel "
                                     "with the chip to itself (serialised stage timing, whole batch per launch)",
                         "alone": alone,
                         "true_bound": "valu: the kernel's limiter is VALU issue, not HBM (see valu)",
                         "valu": valu},
            "stage_ms_per_step_serialised": acc or None,
            "whole_frame_roofline_frac": fps / world * frame_bytes / (HBM_PEAK_GBS * 1e9),
        }
    ex.close()
    del d_imgs
    # the 2-NN leg runs on every rank (sharded when there are several)
    stuck = False
    if not args.no_match:
        # the headline line must not be lost to a failure (or a hang) of a secondary leg
        ham, stuck = with_deadline(lambda: hamming_bench(torch, orbx, dev, world, rank), 240, dev, torch)
        if out is not None:
            out["hamming"] = ham
    if rank == 0:
        extra = not args.no_extra and world == 1
        legs = []
        if extra:
            legs += [("host_path", lambda: host_path_bench(torch, orbx, dev, local_rank, frames)),
                     ("stereo", lambda: stereo_bench(torch, orbx, dev, local_rank)),
                     ("fisheye", lambda: fisheye_bench(torch, orbx, dev, local_rank)),
                     ("kitti", lambda: kitti_bench(torch, orbx, dev, local_rank)),
                     ("kitti_stereo", lambda: kitti_stereo_bench(torch, orbx, dev, local_rank)),
                     ("uhd", lambda: uhd_bench(torch, orbx, dev, local_rank)),
                     ("tracking_calls", lambda: tracking_calls_bench(orbx, local_rank))]
        elif args.stereo:
            legs.append(("stereo", lambda: stereo_bench(torch, orbx, dev, local_rank)))
        if args.bow:
            legs.append(("bow", lambda: bow_bench(torch, orbx, dev, local_rank)))
        if args.guided:
            legs.append(("guided", lambda: guided_bench(torch, orbx, dev, local_rank)))
        for name, fn in legs:
            try:
                out[name] = fn()
            except Exception as e:
                out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        # the interval the reference instruments (host images in -> host keypoints and descriptors
        # out, PCIe both ways included), beside `value`, which is the device-resident rate
        hp = out.get("host_path")
        out["value_host_interval"] = hp["pinned"]["frames_per_s"] if isinstance(hp, dict) and "pinned" in hp else None
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if stuck:
        # a collective is still pending on a worker thread: no orderly shutdown is possible, and the
        # run must not look like a success to whoever started it (the line above says which leg)
        sys.stdout.flush()
        os._exit(3)
    if world > 1:
        dist.barrier()   # the other ranks wait here while rank 0 finishes its single-rank legs
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
