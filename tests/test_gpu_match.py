"""GPU parity of the Hamming matchers against the CPU oracle (bit-exact, ties included)."""
import numpy as np
import pytest

import orb_slam3_amd as orbx

pytestmark = pytest.mark.gpu


def rand_desc(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def planted(rng, nq, nt, frac=0.3):
    """Uniform random descriptors (distances ~128, many exact ties) with a fraction of the
    queries planted as a train row with 0-40 flipped bits."""
    t = rand_desc(rng, nt)
    q = rand_desc(rng, nq)
    for i in range(nq):
        if rng.random() < frac and nt:
            row = t[rng.integers(0, nt)].copy()
            bits = np.unpackbits(row)
            flips = rng.choice(256, rng.integers(0, 41), replace=False)
            bits[flips] ^= 1
            q[i] = np.packbits(bits)
    return q, t


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (3, 0), (64, 64), (513, 1000), (1500, 1500),
                                   (700, 20000), (5, 300000)])
def test_knn2_matches_oracle(oracle, nq, nt):
    from tests.conftest import HAS_GPU
    if nt > 100000 and not HAS_GPU:
        nt = 60000    # the host simulation emulates every matrix instruction lane by lane
    rng = np.random.default_rng(nq * 131 + nt)
    q, t = planted(rng, nq, nt)
    idx, dist = orbx.knn_match2(q, t)
    ridx, rdist = oracle.knn2(q, t)
    assert np.array_equal(dist, rdist)
    assert np.array_equal(idx, ridx)


@pytest.mark.parametrize("nq,nt", [(31, 31), (33, 33), (255, 63), (257, 65), (256, 64), (300, 127), (32, 129),
                                   (65, 4097)])
def test_knn2_tile_and_workgroup_boundaries(oracle, nq, nt):
    """Sizes around the matrix-core kernel's units: 32 train rows per tile, 64 per stage, 32 queries
    per tile, 256 per workgroup."""
    rng = np.random.default_rng(nq * 7 + nt)
    q, t = planted(rng, nq, nt)
    idx, dist = orbx.knn_match2(q, t)
    ridx, rdist = oracle.knn2(q, t)
    assert np.array_equal(dist, rdist) and np.array_equal(idx, ridx)


def test_knn2_extreme_descriptors(oracle):
    """All-zero and all-one rows (distances 0 and 256, the ends of the product's range), and rows
    that differ in a single bit at every position of a 32-bit word."""
    rng = np.random.default_rng(11)
    t = rand_desc(rng, 200)
    t[0] = 0
    t[1] = 255
    for b in range(32):
        bits = np.zeros(256, np.uint8)
        bits[b] = 1
        t[2 + b] = np.packbits(bits, bitorder="little")
    q = np.concatenate([t[:40], 255 - t[:40], rand_desc(rng, 20)])
    idx, dist = orbx.knn_match2(q, t)
    ridx, rdist = oracle.knn2(q, t)
    assert np.array_equal(dist, rdist) and np.array_equal(idx, ridx)
    assert dist[0, 0] == 0 and dist[1, 0] == 0 and dist.max() <= 256


@pytest.mark.parametrize("knob", ["ORBX_KNN2_I8", "ORBX_KNN2_VALU"])
def test_knn2_forms_agree(oracle, knob, monkeypatch):
    """The brute-force scan exists in three forms: on the block-scaled FP4 matrix instruction (the
    default), on the int8 matrix instruction (ORBX_KNN2_I8=1) and as xor + popcount on the vector
    ALU (ORBX_KNN2_VALU=1).  The knobs are read at every call; all forms must equal the oracle,
    planted near-duplicates, ties and segment boundaries included."""
    from tests.conftest import HAS_GPU
    rng = np.random.default_rng(21)
    q, t = planted(rng, 700, 5000) if HAS_GPU else planted(rng, 200, 1500)
    ridx, rdist = oracle.knn2(q, t)
    idx, dist = orbx.knn_match2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    monkeypatch.setenv(knob, "1")
    idx2, dist2 = orbx.knn_match2(q, t)
    assert np.array_equal(idx2, ridx) and np.array_equal(dist2, rdist)


def test_knn2_duplicate_train_rows_tie_break(oracle):
    rng = np.random.default_rng(3)
    t = np.repeat(rand_desc(rng, 50), 40, axis=0)  # every row 40 times -> massive ties
    rng.shuffle(t, axis=0)
    q = t[rng.integers(0, len(t), 300)]
    idx, dist = orbx.knn_match2(q, t)
    ridx, rdist = oracle.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert (dist[:, 0] == 0).all() and (idx[:, 0] < idx[:, 1]).all()


def test_knn2_device_sharded_merge(oracle):
    """Shard the train set 1, 2, 4 and 8 ways in one process; merging the per-shard
    results must reproduce the unsharded answer exactly."""
    import ctypes as C
    import torch
    from orb_slam3_amd._lib import lib, check
    from tests.conftest import TORCH_DEVICE
    rng = np.random.default_rng(11)
    from tests.conftest import HAS_GPU
    q, t = planted(rng, 777, 40000) if HAS_GPU else planted(rng, 300, 8000)   # host simulation: smaller
    ridx, rdist = oracle.knn2(q, t)
    dq = torch.from_numpy(q).to(TORCH_DEVICE)
    dt = torch.from_numpy(t).to(TORCH_DEVICE)
    for shards in (1, 2, 4, 8):
        bounds = np.linspace(0, len(t), shards + 1).astype(int)
        bounds = (bounds // 8) * 8
        bounds[-1] = len(t)
        pidx = torch.empty((shards, len(q), 2), dtype=torch.int32, device=TORCH_DEVICE)
        pdst = torch.empty((shards, len(q), 2), dtype=torch.int16, device=TORCH_DEVICE)
        for s in range(shards):
            sl = dt[bounds[s]:bounds[s + 1]]
            check(lib().orbx_knn2_device(C.c_void_p(dq.data_ptr()), len(q), C.c_void_p(sl.data_ptr()),
                                         sl.shape[0], int(bounds[s]), C.c_void_p(pidx[s].data_ptr()),
                                         C.c_void_p(pdst[s].data_ptr()), None))
        oi = torch.empty((len(q), 2), dtype=torch.int32, device=TORCH_DEVICE)
        od = torch.empty((len(q), 2), dtype=torch.int16, device=TORCH_DEVICE)
        check(lib().orbx_top2_merge_device(C.c_void_p(pidx.data_ptr()), C.c_void_p(pdst.data_ptr()), shards,
                                           len(q), C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), None))
        if TORCH_DEVICE == "cuda":
            torch.cuda.synchronize()
        assert np.array_equal(oi.cpu().numpy().view(np.uint32), ridx), shards
        assert np.array_equal(od.cpu().numpy().view(np.uint16), rdist), shards


def test_hamming_is_popcount(oracle):
    rng = np.random.default_rng(5)
    a, b = rand_desc(rng, 200), rand_desc(rng, 200)
    for i in range(200):
        want = int(np.unpackbits(a[i] ^ b[i]).sum())
        assert orbx.descriptor_distance(a[i], b[i]) == want == oracle.hamming(a[i], b[i])


def test_stereo_pair_extract_and_band_match(oracle):
    """Config 2 shape: a 752x480 stereo pair (right = left shifted), 1200 kp per camera,
    extraction followed by the row-band descriptor search."""
    left = orbx.textured_frame(21, 752 + 40, 480)
    right = np.ascontiguousarray(left[:, 25:25 + 752])
    left = np.ascontiguousarray(left[:, :752])
    exl = orbx.ORBextractor(1200, 1.2, 8, 20, 7)
    exr = orbx.ORBextractor(1200, 1.2, 8, 20, 7)
    _, kl, dl = exl(left, None, (0, 0))
    _, kr, dr = exr(right, None, (0, 0))
    sf = exl.GetScaleFactors()
    bi, bd = orbx.stereo_band_match(kl, dl, kr, dr, sf, 480, 0.0, 435.0)
    ri, rd = oracle.stereo_band_match(kl, dl, kr, dr, sf, 480, 0.0, 435.0)
    assert np.array_equal(bi, ri) and np.array_equal(bd, rd)
    assert (bd < 75).sum() > 100  # plenty of accepted matches on a shifted copy


def test_stereo_band_random(oracle):
    rng = np.random.default_rng(9)
    n = 900
    def kps():
        k = np.zeros(n, orbx.KEYPOINT_DTYPE)
        k["x"] = rng.uniform(20, 700, n).astype(np.float32)
        k["y"] = rng.uniform(20, 460, n).astype(np.float32)
        k["octave"] = rng.integers(0, 8, n)
        return k
    kl, kr = kps(), kps()
    base = rand_desc(rng, 30)
    dl = base[rng.integers(0, 30, n)]
    dr = base[rng.integers(0, 30, n)]  # few distinct descriptors -> distance ties
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    bi, bd = orbx.stereo_band_match(kl, dl, kr, dr, sf, 480, 0.0, 300.0)
    ri, rd = oracle.stereo_band_match(kl, dl, kr, dr, sf, 480, 0.0, 300.0)
    assert np.array_equal(bi, ri) and np.array_equal(bd, rd)


def test_fisheye_config_knn_on_lapping_subsets(oracle):
    """Config 3 shape: 512x512 pair, 1500 kp, lapping area -> 2-NN between the lapping subsets
    (rows [mono:], core/src/Frame.cc:1159-1177) + ratio test."""
    a = orbx.textured_frame(31, 512, 512)
    b = np.roll(a, 9, axis=1)
    ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7)
    ml, kl, dl = ex(a, None, (100, 511))
    mr, kr, dr = ex(b, None, (0, 400))
    idx, dist = orbx.knn_match2(dl[ml:], dr[mr:])
    ridx, rdist = oracle.knn2(dl[ml:], dr[mr:])
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert orbx.ratio_test(idx, dist).sum() > 50


@pytest.mark.needs_real_device
def test_knn2_large_properties():
    """200k queries x 2M train rows (4e11 pairs; the oracle cannot run this): size-independent
    properties - sharded == unsharded, planted rows found at distance 0 at their lowest index,
    ordered neighbours, and a random sample checked against numpy brute force."""
    import ctypes as C
    import torch
    from orb_slam3_amd._lib import lib, check
    g = torch.Generator(device="cuda").manual_seed(5)
    nq, nt = 200_000, 2_000_000
    t = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device="cuda", generator=g)
    q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device="cuda", generator=g)
    t[nt - 1000:] = t[:1000]                    # duplicates far apart: index tie-break across shards
    src = torch.randint(0, nt, (nq // 10,), device="cuda", generator=g)
    q[: nq // 10] = t[src]

    def run(tt, off, oi, od):
        check(lib().orbx_knn2_device(C.c_void_p(q.data_ptr()), nq, C.c_void_p(tt.data_ptr()), tt.shape[0], off,
                                     C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), None))
    idx = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    dst = torch.empty((nq, 2), dtype=torch.int16, device="cuda")
    run(t, 0, idx, dst)
    shards = 8
    b = orbx.shard_bounds(nt, shards)
    pi = torch.empty((shards, nq, 2), dtype=torch.int32, device="cuda")
    pd = torch.empty((shards, nq, 2), dtype=torch.int16, device="cuda")
    for s in range(shards):
        run(t[b[s]:b[s + 1]], b[s], pi[s], pd[s])
    mi, md = torch.empty_like(idx), torch.empty_like(dst)
    check(lib().orbx_top2_merge_device(C.c_void_p(pi.data_ptr()), C.c_void_p(pd.data_ptr()), shards, nq,
                                       C.c_void_p(mi.data_ptr()), C.c_void_p(md.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.equal(mi, idx) and torch.equal(md, dst)
    i_np = idx.cpu().numpy().view(np.uint32)
    d_np = dst.cpu().numpy().view(np.uint16)
    assert (d_np[:, 0] <= d_np[:, 1]).all() and (i_np[:, 0] != i_np[:, 1]).all()
    assert (d_np[: nq // 10, 0] == 0).all()
    src_np = src.cpu().numpy()
    dup = src_np >= nt - 1000                   # planted from the duplicated tail -> earliest copy wins
    want = np.where(dup, src_np - (nt - 1000), src_np)
    hit = i_np[: nq // 10, 0]
    # the planted row (or an identical earlier row) must be reported
    assert (hit <= want).all()
    assert (hit[~dup & (src_np >= 1000)] == want[~dup & (src_np >= 1000)]).mean() > 0.999
    t_np = t.cpu().numpy()
    q_np = q.cpu().numpy()
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint16)
    for qi in np.random.default_rng(1).integers(0, nq, 24):
        d = lut[np.bitwise_xor(t_np, q_np[qi])].sum(axis=1)
        order = np.lexsort((np.arange(nt), d))[:2]
        assert i_np[qi].tolist() == order.tolist() and d_np[qi].tolist() == d[order].tolist()


@pytest.mark.needs_real_device
def test_knn2_sharded_entry_point_single_rank(oracle):
    """orbx_knn2_sharded through a one-rank RCCL communicator of the library's own making: the
    per-shard search, the all-gather call and the merge run as they do on N ranks, and the result
    equals the unsharded search (workspace given by the caller, then allocated by the call)."""
    import ctypes as C
    import torch
    from orb_slam3_amd._lib import lib, check
    from orb_slam3_amd.matcher import RcclComm
    rng = np.random.default_rng(21)
    q, t = planted(rng, 900, 30000)
    ridx, rdist = oracle.knn2(q, t)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    comm = RcclComm(0)
    assert comm.world == 1
    i, d = orbx.knn_match2_sharded(dq, dt, 0, comm=comm)
    torch.cuda.synchronize()
    assert np.array_equal(i.cpu().numpy().view(np.uint32), ridx)
    assert np.array_equal(d.cpu().numpy().view(np.uint16), rdist)
    # a shard that starts at a global offset, workspace left to the call
    oi = torch.empty((len(q), 2), dtype=torch.int32, device="cuda")
    od = torch.empty((len(q), 2), dtype=torch.int16, device="cuda")
    check(lib().orbx_knn2_sharded(comm._h, C.c_void_p(dq.data_ptr()), len(q), C.c_void_p(dt.data_ptr()), len(t),
                                  5000, C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), None, 0, None))
    torch.cuda.synchronize()
    assert np.array_equal(oi.cpu().numpy().view(np.uint32), ridx + np.uint32(5000))
    assert np.array_equal(od.cpu().numpy().view(np.uint16), rdist)
    comm.close()


def _two_gpu_worker(rank, world, port, q, t, out):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    import orb_slam3_amd as ox
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # only hands the RCCL id over
    b = ox.shard_bounds(len(t), world)
    dq = torch.from_numpy(q).cuda()
    shard = torch.from_numpy(t[b[rank]:b[rank + 1]].copy()).cuda()
    i, d = ox.knn_match2_sharded(dq, shard, b[rank])
    torch.cuda.synchronize()
    out[rank] = (i.cpu().numpy().view(np.uint32).copy(), d.cpu().numpy().view(np.uint16).copy())
    dist.destroy_process_group()


@pytest.mark.needs_real_device
def test_knn2_sharded_over_rccl_equals_unsharded(oracle):
    """The sharded search on every GPU of the box (2 at least, else skipped): HIP kernels + one
    RCCL all-gather + merge, one process per GPU, against the oracle's unsharded answer."""
    import socket
    import torch
    import torch.multiprocessing as mp
    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip("needs 2 GPUs")
    rng = np.random.default_rng(world)
    t = rand_desc(rng, 2001)
    t = np.repeat(t[:400], 6, axis=0)[:2001]          # duplicates across shard boundaries
    q = t[rng.integers(0, len(t), 150)].copy()
    q[::3] ^= rng.integers(0, 256, (50, 32), dtype=np.uint8) & 0x11
    ridx, rdist = oracle.knn2(q, t)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_two_gpu_worker, args=(r, world, port, q, t, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for r in range(world):
        assert np.array_equal(out[r][0], ridx) and np.array_equal(out[r][1], rdist)


@pytest.mark.needs_real_device
def test_knn2_full_database_size_properties():
    """1 M queries x 10 M train rows (1e13 pairs, the largest stated configuration; one pass
    unsharded, one pass as 8 shards + merge): sharded == unsharded, planted rows found at
    distance 0 at the lowest index that holds them, ordered neighbours, and a sample of queries
    against numpy brute force."""
    import ctypes as C
    import torch
    from orb_slam3_amd._lib import lib, check
    g = torch.Generator(device="cuda").manual_seed(9)
    nq, nt = 1_000_000, 10_000_000
    t = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device="cuda", generator=g)
    q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device="cuda", generator=g)
    t[nt - 5000:] = t[:5000]                    # duplicates in the first and the last shard
    npl = nq // 10
    src = torch.randint(0, nt, (npl,), device="cuda", generator=g)
    q[:npl] = t[src]
    flips = torch.randint(0, 256, (npl, 32), dtype=torch.uint8, device="cuda", generator=g) & 0x10
    q[npl // 2:npl] ^= flips[npl // 2:]         # second half of the planted rows: up to 32 flipped bits

    def run(tt, off, oi, od):
        check(lib().orbx_knn2_device(C.c_void_p(q.data_ptr()), nq, C.c_void_p(tt.data_ptr()), tt.shape[0], off,
                                     C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), None))
    idx = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    dst = torch.empty((nq, 2), dtype=torch.int16, device="cuda")
    run(t, 0, idx, dst)
    torch.cuda.synchronize()
    shards = 8
    b = orbx.shard_bounds(nt, shards)
    pi = torch.empty((shards, nq, 2), dtype=torch.int32, device="cuda")
    pd = torch.empty((shards, nq, 2), dtype=torch.int16, device="cuda")
    for s in range(shards):
        run(t[b[s]:b[s + 1]], b[s], pi[s], pd[s])
    mi, md = torch.empty_like(idx), torch.empty_like(dst)
    check(lib().orbx_top2_merge_device(C.c_void_p(pi.data_ptr()), C.c_void_p(pd.data_ptr()), shards, nq,
                                       C.c_void_p(mi.data_ptr()), C.c_void_p(md.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.equal(mi, idx) and torch.equal(md, dst)
    i_np = idx.cpu().numpy().view(np.uint32)
    d_np = dst.cpu().numpy().view(np.uint16)
    assert (d_np[:, 0] <= d_np[:, 1]).all() and (i_np[:, 0] != i_np[:, 1]).all()
    assert (i_np < nt).all()
    assert (d_np[: npl // 2, 0] == 0).all() and (d_np[npl // 2:npl, 0] <= 32).all()
    src_np = src.cpu().numpy()
    dup = src_np >= nt - 5000
    want = np.where(dup, src_np - (nt - 5000), src_np)
    hit = i_np[: npl // 2, 0]
    assert (hit <= want[: npl // 2]).all()      # the planted row or an identical earlier one
    t_np = t.cpu().numpy()
    q_np = q.cpu().numpy()
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint16)
    for qi in np.random.default_rng(1).integers(0, nq, 6):
        d = lut[np.bitwise_xor(t_np, q_np[qi])].sum(axis=1, dtype=np.uint16)
        best = np.flatnonzero(d <= np.partition(d, 1)[1])
        order = best[np.lexsort((best, d[best]))][:2]
        assert i_np[qi].tolist() == order.tolist() and d_np[qi].tolist() == d[order].tolist()
