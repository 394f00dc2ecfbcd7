#!/usr/bin/env python3
"""Throughput bench of the ORB extractor hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one batched extraction (pyramid, FAST cells, quadtree, orientation,
blur, descriptors) of `--batch` synthetic 752x480 frames that are already
resident in HBM; outputs stay in HBM.  One process per GPU; for N > 1 launch
through torch.distributed.run (RCCL is only used for the timing barrier: frames
are independent, so the data path has no collective and scaling is weak).
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 752, 480
PARAMS = (1000, 1.2, 8, 20, 7)   # nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST
LAP = (0, 1000)                  # monocular call, core/src/Frame.cc:360
HBM_PEAK_GBS = 8000.0


def level_sizes(w, h, nlevels, scale_factor):
    sizes, s = [], np.float32(1.0)
    for l in range(nlevels):
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


KERNEL_OF_STAGE = {"fast": "k_fast_cells", "blur": "k_blur"}


def pmc_traffic(stage, batch):
    """HBM-side bytes per launch of the stage's kernel(s) from the committed rocprofv3 counter
    passes of this same command (FETCH_SIZE and WRITE_SIZE are collected in separate runs and
    cannot be read from inside the process).  FETCH_SIZE is in KiB and, on gfx950, tallies the
    128-byte requests of wide coalesced reads at 64 bytes, hence the factor 2; WRITE_SIZE KiB is
    exact.  Returns None when no summary for this batch size is present."""
    path = os.path.join(ROOT, "profiles", "r1_final_bench_b%d_pmc_summary.json" % batch)
    if not os.path.exists(path):
        return None, None
    with open(path) as fh:
        summ = json.load(fh)
    for name, r in summ.items():
        if name.startswith(KERNEL_OF_STAGE[stage]) and "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            return r["FETCH_SIZE"] * 2048.0 + r["WRITE_SIZE"] * 1024.0, os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(frames, seconds_budget=15.0):
    """The CPU oracle (a restatement of the reference extractor; the reference itself needs
    OpenCV and cannot be built here) on this host's cores, one extractor per thread,
    independent frames - the reference's thread-per-camera model (core/src/Frame.cc:151-154)."""
    from tests import oracle_lib as O
    import subprocess
    so = os.path.join(ROOT, "oracle", "liborb_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    ncores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    ncores = max(1, min(ncores, len(os.sched_getaffinity(0))))
    # calibrate with one frame, then size the sample for ~seconds_budget of wall time
    ex = O.Extractor(*PARAMS)
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
            "sample": "%d synthetic 752x480 frames on %d threads (%.1f s), CPU oracle "
                      "oracle/orb_oracle.cpp -O2, one extractor per thread" % (total, ncores, dt),
            "single_thread_ms_per_frame": one * 1e3}


def hamming_bench(torch, orbx, dev, nq=65536, nt=262144, reps=3):
    """Secondary metric: brute-force Hamming 2-NN pairs/s on device-resident descriptors
    (uniform random 256-bit rows).  Roofline = VALU: 16 lane-ops per pair (8 xor + 8 bcnt)
    against 256 CUs x 64 lanes x 2.4 GHz."""
    import ctypes as C
    from orb_slam3_amd._lib import lib, check
    q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev)
    t = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device=dev)
    idx = torch.empty((nq, 2), dtype=torch.int32, device=dev)
    dst = torch.empty((nq, 2), dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run():
        check(lib().orbx_knn2_device(C.c_void_p(q.data_ptr()), nq, C.c_void_p(t.data_ptr()), nt, 0,
                                     C.c_void_p(idx.data_ptr()), C.c_void_p(dst.data_ptr()),
                                     C.c_void_p(stream)))
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    pairs = nq * nt / sec
    peak_ops = 256 * 64 * 2.4e9
    return {"metric": "Hamming pairs/s (brute-force 2-NN)", "value": pairs, "unit": "pairs/s",
            "queries": nq, "train": nt, "ms": sec * 1e3,
            "roofline": {"bound": "valu", "achieved": pairs * 16 / 1e12, "peak": peak_ops / 1e12,
                         "unit": "T lane-ops/s", "frac": pairs * 16 / peak_ops}}


def stereo_bench(torch, orbx, dev, local_rank, batch=128, reps=5):
    """Secondary metric: rectified stereo pairs/s (config: 752x480, 1200 kp per camera,
    vLappingArea (0,0)): both extractions plus the device-resident ComputeStereoMatches
    (band search, L1 refinement, depth, median rule), images resident in HBM."""
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
    return {"metric": "stereo pairs/s @752x480, 2x1200kp, extraction + ComputeStereoMatches",
            "value": batch / sec, "unit": "pairs/s", "pairs_per_step": batch, "ms_per_step": sec * 1e3,
            "match_stage_ms_per_step": sec_match * 1e3, "matches_per_pair": nmatch}


def _bench_vocabulary(orbx, local_rank, k=10, depth=5):
    """Synthetic vocabulary of k^depth words (children = bit-flipped copies of their parent)."""
    from tests import voc_util
    arrays = voc_util.make_vocabulary(k, depth, seed=1)
    return orbx.bow.ORBVocabulary.from_arrays(k, depth, *arrays, device=local_rank), arrays


def bow_bench(torch, orbx, dev, local_rank, batch=256, reps=5, k=10, depth=5):
    """Secondary metric (opt-in): frames/s of Frame::ComputeBoW on device-resident descriptors."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per step per GPU")
    ap.add_argument("--no-match", action="store_true", help="skip the Hamming 2-NN leg")
    ap.add_argument("--stereo", action="store_true",
                    help="add the stereo leg (752x480 pairs, 2x1200 kp: both extractions + ComputeStereoMatches)")
    ap.add_argument("--bow", action="store_true", help="add the bag-of-words leg (frames/s of ComputeBoW)")
    ap.add_argument("--guided", action="store_true", help="add the guided-matching leg (ms per SearchByProjection call)")
    ap.add_argument("--distinct", type=int, default=32, help="distinct synthetic frames (tiled to the batch)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--stage-times", action="store_true",
                    help="also time every stage serialised on one stream (whole batch per launch)")
    args = ap.parse_args()

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
    # launches run on, in the same schedule as the timed loop (a batch runs as two parts)
    det_ms, det_frames, nsample = 0.0, 0, max(3, min(args.steps, 10))
    for _ in range(nsample):
        step()
        for ms, fr in ex.detect_launch_times():
            det_ms += ms
            det_frames += fr
    n_launches = nsample * len(ex.detect_launch_times())
    det_ms_per_launch = det_ms / n_launches
    det_frames_per_launch = det_frames / n_launches

    # optional: every stage serialised on one stream, whole batch per launch
    acc = {}
    if args.stage_times:
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

    if rank == 0:
        stage_bytes, frame_bytes = algorithmic_bytes(W, H, PARAMS[2], PARAMS[1], int(round(nkp)))
        fps = world * B * args.steps / dt
        dom = "fast"
        bytes_per_launch = stage_bytes[dom] * det_frames_per_launch
        achieved = bytes_per_launch / (det_ms_per_launch * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dom, B)
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
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "algorithmic_bytes_per_frame": stage_bytes[dom],
                         "launch_ms": det_ms_per_launch, "frames_per_launch": det_frames_per_launch},
            "stage_ms_per_step_serialised": acc or None,
            "whole_frame_roofline_frac": fps / world * frame_bytes / (HBM_PEAK_GBS * 1e9),
        }
        if not args.no_match:
            out["hamming"] = hamming_bench(torch, orbx, dev)
        if args.stereo:
            out["stereo"] = stereo_bench(torch, orbx, dev, local_rank)
        if args.bow:
            out["bow"] = bow_bench(torch, orbx, dev, local_rank)
        if args.guided:
            out["guided"] = guided_bench(torch, orbx, dev, local_rank)
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames)
            out["gpu_over_cpu"] = fps / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()   # the other ranks wait here while rank 0 finishes its single-rank legs
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
