"""GPU parity of the extractor against the CPU oracle, stage by stage, through the C ABI."""
import numpy as np
import pytest

import orb_slam3_amd as orbx

pytestmark = pytest.mark.gpu

ANGLE_TOL_DEG = float(np.degrees(1e-4))  # angles must agree within 1e-4 rad


def adversarial(kind, w, h):
    rng = np.random.default_rng(7)
    if kind == "zeros":
        return np.zeros((h, w), np.uint8)
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "checker":
        yy, xx = np.mgrid[0:h, 0:w]
        return (((yy // 8 + xx // 8) & 1) * 255).astype(np.uint8)
    if kind == "ramp":
        return np.tile(np.linspace(0, 255, w).astype(np.uint8), (h, 1))
    if kind == "dots":
        img = np.full((h, w), 40, np.uint8)
        img[19::36, 19::36] = 255
        img[20::38, 18::36] = 250
        return img
    raise ValueError(kind)


def compare_full(oracle, img, params, lap, check_stages=True):
    nf, sf, nl, ini, mn = params
    ref = oracle.Extractor(nf, sf, nl, ini, mn)
    st, rmono, rk, rd = ref.extract(img, lap)
    assert st == 0
    ex = orbx.ORBextractor(nf, sf, nl, ini, mn)
    mono, k, d = ex(img, None, lap)
    if check_stages:
        for l in range(nl):
            w, h, s = ref.level_dims(l)
            assert ex.level_dims(l)[:2] == (w, h)
            got = ex.debug_padded(0, l)
            assert np.array_equal(got, ref.padded(l)[:, :w + 38]), "pyramid level %d" % l
            assert np.array_equal(ex.mvImagePyramid[l], ref.padded(l)[19:19 + h, 19:19 + w])
            rc = ref.candidates(l)
            gc = ex.debug_candidates(0, l)
            assert np.array_equal(gc, rc), "candidates level %d (%d vs %d)" % (l, len(gc), len(rc))
            rkeys, _ = ref.keys(l)
            gs = ex.debug_selected(0, l)
            assert len(gs) == len(rkeys), "selected count level %d" % l
            assert np.array_equal(gs[:, 0] + 16, rkeys["x"].astype(np.int32))
            assert np.array_equal(gs[:, 1] + 16, rkeys["y"].astype(np.int32))
            assert np.array_equal(gs[:, 2], rkeys["response"].astype(np.int32))
            rb = ref.blurred(l)
            if rb is not None:
                assert np.array_equal(ex.debug_blurred(0, l), rb), "blur level %d" % l
    assert mono == rmono
    assert len(k) == len(rk)
    if len(rk) == 0:
        assert d is None
        return 0
    for fld in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(k[fld], rk[fld]), fld
    dang = np.abs(k["angle"] - rk["angle"])
    dang = np.minimum(dang, 360.0 - dang)
    assert dang.max() <= ANGLE_TOL_DEG, dang.max()
    bad = np.unpackbits(d ^ rd, axis=1).sum(axis=1) > 0
    # descriptor bit differences are only tolerated on <= 0.1 % of the keypoints
    assert bad.mean() <= 1e-3, "descriptor mismatches on %d of %d keypoints" % (bad.sum(), len(bad))
    return int(bad.sum())


def test_textured_752x480_mono(oracle):
    img = orbx.textured_frame(0, 752, 480)
    assert compare_full(oracle, img, (1000, 1.2, 8, 20, 7), (0, 1000)) == 0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_textured_seeds(oracle, seed):
    img = orbx.textured_frame(seed, 752, 480)
    compare_full(oracle, img, (1200, 1.2, 8, 20, 7), (0, 0))


@pytest.mark.parametrize("kind", ["zeros", "noise", "checker", "ramp", "dots"])
def test_adversarial_320x240(oracle, kind):
    img = adversarial(kind, 320, 240)
    compare_full(oracle, img, (500, 1.2, 4, 20, 7), (100, 200))


def test_noise_752x480_phase_b(oracle):
    compare_full(oracle, adversarial("noise", 752, 480), (1000, 1.2, 8, 20, 7), (0, 0))


@pytest.mark.parametrize("w,h,nf,lap", [(512, 512, 1500, (0, 511)), (1241, 376, 2000, (0, 0)),
                                         (640, 480, 5000, (0, 1000))])
def test_other_configs(oracle, w, h, nf, lap):
    compare_full(oracle, orbx.textured_frame(11, w, h), (nf, 1.2, 8, 20, 7), lap)


@pytest.mark.parametrize("nf,sf,nl", [(50, 1.2, 1), (1000, 1.5, 4), (300, 1.2, 8), (1000, 1.1, 8)])
def test_param_sweep(oracle, nf, sf, nl):
    compare_full(oracle, orbx.textured_frame(5, 480, 360), (nf, sf, nl, 20, 7), (100, 400))


def test_batch_equals_single(oracle):
    imgs = np.stack([orbx.textured_frame(s, 376, 240) for s in range(5)])
    ex = orbx.ORBextractor(400, 1.2, 6, 20, 7, max_batch=8)
    res = ex.extract_batch(imgs, (0, 0))
    ref = oracle.Extractor(400, 1.2, 6, 20, 7)
    for f in range(5):
        st, rmono, rk, rd = ref.extract(imgs[f], (0, 0))
        mono, k, d = res[f]
        assert mono == rmono and len(k) == len(rk)
        for fld in ("x", "y", "size", "response", "octave"):
            assert np.array_equal(k[fld], rk[fld])
        assert np.array_equal(d, rd)


def test_errors():
    ex = orbx.ORBextractor(100, 1.2, 8, 20, 7)
    mono, k, d = ex(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(k) == 0 and d is None
    with pytest.raises(orbx.OrbxError) as e:
        ex(np.zeros((100, 100), np.uint8))
    assert e.value.status == -2


@pytest.mark.parametrize("entry", ["orbx_debug_sort_records", "orbx_debug_sort_records_team"])
def test_device_sort_matches_std_sort(oracle, entry):
    """The quadtree kernels' sorts (one wave; a team of waves taking the partitions breadth first)
    must give std::sort's permutation, ties included."""
    import ctypes as C
    from orb_slam3_amd._lib import lib, check, ptr
    rng = np.random.default_rng(17)
    cases = []
    for n in [1, 2, 15, 16, 17, 33, 64, 65, 100, 128, 200, 233, 500, 1000, 2200, 3000]:
        for nkeys in (1, 2, 5, 40, 100000):
            for mode in range(4):
                size = rng.integers(2, 2 + nkeys, n)
                ulx = rng.integers(0, 1 + nkeys // 3, n)
                if mode == 1:
                    size = np.arange(n) % nkeys + 2
                if mode == 2:
                    size = (n - np.arange(n)) % nkeys + 2
                if mode == 3:
                    size = np.where(np.arange(n) < n // 2, np.arange(n) % nkeys, (n - np.arange(n)) % nkeys) + 2
                cases.append((size.astype(np.int32), (ulx % 4096).astype(np.int32)))
    for size, ulx in cases:
        n = len(size)
        perm_ref = oracle.sort_sized_nodes(size, ulx)
        rec = ((size.astype(np.uint64) << np.uint64(12) | ulx.astype(np.uint64)) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        rec = np.ascontiguousarray(rec)
        check(getattr(lib(), entry)(0, ptr(rec), n))
        perm = (rec & np.uint64(0xFFFFFFFF)).astype(np.int32)
        assert np.array_equal(perm, perm_ref), (n, size[:8], ulx[:8])


def test_4k_frame_10000_features(oracle):
    """Config 5 shape: one synthetic 3840x2160 frame, 10000 features (the quadtree's node
    arrays exceed LDS here and live in the global workspace)."""
    img = orbx.textured_frame(77, 3840, 2160)
    compare_full(oracle, img, (10000, 1.2, 8, 20, 7), (0, 0), check_stages=False)


def test_many_features_small_image(oracle):
    compare_full(oracle, adversarial("noise", 640, 480), (20000, 1.2, 6, 20, 7), (0, 0), check_stages=False)


def test_batch_position_independence_and_idempotence():
    """A frame's result does not depend on where it sits in a batch, on the batch size, or
    on what the handle processed before (re-planning for another image size included)."""
    ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=64)
    a = orbx.textured_frame(1, 752, 480)
    b = orbx.textured_frame(2, 752, 480)
    batch = np.stack([a, b] * 8 + [b, a] * 8)          # 32 frames
    res = ex.extract_batch(batch, (0, 1000))
    ex.extract_batch(np.stack([orbx.textured_frame(3, 512, 512)] * 3), (0, 0))   # other size in between
    res2 = ex.extract_batch(batch[::-1].copy(), (0, 1000))[::-1]
    single = ex(a, None, (0, 1000))
    for r in (res, res2):
        for f in range(32):
            ref = res[0] if np.array_equal(batch[f], a) else res[1]
            assert r[f][0] == ref[0]
            assert np.array_equal(r[f][1], ref[1]) and np.array_equal(r[f][2], ref[2])
    assert single[0] == res[0][0] and np.array_equal(single[1], res[0][1]) and np.array_equal(single[2], res[0][2])


def test_device_resident_batch_matches_host_batch():
    import torch
    from tests.conftest import TORCH_DEVICE
    frames = np.stack([orbx.textured_frame(40 + i, 752, 480) for i in range(24)])
    ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=24)
    host = ex.extract_batch(frames, (0, 1000))
    d = torch.from_numpy(frames).to(TORCH_DEVICE)
    ex.extract_batch_device(d.data_ptr(), 24, 752, 480, 752, (0, 1000))
    ex.sync()
    for f in range(24):
        mono, k, desc = ex.download(f)
        assert mono == host[f][0] and np.array_equal(k, host[f][1]) and np.array_equal(desc, host[f][2])


@pytest.mark.parametrize("env", [{"ORBX_PYR_XCD": "0"}, {"ORBX_PYR_WAVES": "2"}, {"ORBX_PYR_WAVES": "4"},
                                 {"ORBX_BLUR_PLANES": "1"}, {"ORBX_BLUR_PLANES": "1", "ORBX_BLUR_WAVES": "4"},
                                 {"ORBX_BLUR_PLANES": "1", "ORBX_SIDE_BLUR": "0"},
                                 {"ORBX_BLUR_PLANES": "1", "ORBX_SIDE_BLUR": "1"},
                                 {"ORBX_BLUR_PLANES": "1", "ORBX_SIDE_BLUR": "2"},
                                 {"ORBX_FAST_SINGLE_PASS": "1"}, {"ORBX_OCT_KEYS_LDS": "0"}, {"ORBX_OCT_TEAM": "0"},
                                 {"ORBX_OCT_TEAM": "0", "ORBX_OCT_KEYS_LDS": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_timing_knobs_do_not_change_results(oracle, env, monkeypatch):
    """The environment knobs that exist for timing experiments (workgroup shapes and numbering, the
    full-plane blur schedule with its separate descriptor kernel and the blur's place in that
    chain, one detection pass instead of two, quadtree keys outside LDS) select other launch
    geometries and code paths; every stage must still equal the oracle."""
    for knob, value in env.items():
        monkeypatch.setenv(knob, value)
    compare_full(oracle, orbx.textured_frame(11, 640, 400), (600, 1.2, 6, 20, 7), (0, 1000))


@pytest.mark.parametrize("team", ["0", "64"])
@pytest.mark.parametrize("case", ["textured", "noise", "wide", "checker"])
def test_quadtree_by_one_wave_and_by_a_team_of_waves(oracle, case, team, monkeypatch):
    """Small batches build their quadtrees with a workgroup of four waves per (frame, level), large
    ones with one wave; ORBX_OCT_TEAM=0 sends a single frame through the one-wave kernel.  Both
    must equal the oracle: on a textured frame, on noise (more candidates on the big levels than the
    team's LDS buffers hold: those levels are built by the team kernel's one-wave branch), on a
    strip with seven quadtree roots (more than the team's four packed counters) and on a checkerboard
    (every corner has the same score, so the result hangs on the order of the keys)."""
    monkeypatch.setenv("ORBX_OCT_TEAM", team)
    if case == "textured":
        img, params = orbx.textured_frame(21, 752, 480), (1000, 1.2, 8, 20, 7)
    elif case == "noise":
        img, params = adversarial("noise", 752, 480), (1500, 1.2, 8, 20, 7)
    elif case == "wide":
        img, params = orbx.textured_frame(22, 1280, 220), (800, 1.2, 5, 20, 7)
    else:
        img, params = adversarial("checker", 640, 480), (700, 1.2, 8, 20, 7)
    compare_full(oracle, img, params, (0, 1000))


def test_detect_launch_history():
    """The library keeps the detection-launch events of the last 32 batched calls, so the launches of
    a run of back-to-back calls can be read after the run, oldest first, without a wait between
    the calls (one launch per call for batches this small)."""
    import torch
    from tests.conftest import TORCH_DEVICE
    frames = np.stack([orbx.textured_frame(60 + i, 376, 240) for i in range(4)])
    ex = orbx.ORBextractor(400, 1.2, 6, 20, 7, max_batch=4)
    d = torch.from_numpy(frames).to(TORCH_DEVICE)
    for n in (4, 3, 2):
        ex.extract_batch_device(d.data_ptr(), n, 376, 240, 376, (0, 0))
    hist = ex.detect_launch_times(last_calls=3)
    assert [fr for _, fr in hist] == [4, 3, 2]
    assert all(ms >= 0.0 for ms, _ in hist)
    assert [fr for _, fr in ex.detect_launch_times()] == [2]
    assert [fr for _, fr in ex.detect_launch_times(last_calls=100)] == [4, 3, 2]
    for _ in range(40):
        ex.extract_batch_device(d.data_ptr(), 1, 376, 240, 376, (0, 0))
    assert [fr for _, fr in ex.detect_launch_times(last_calls=100)] == [1] * 32


def test_split_batch_matches_single_frames():
    """Batches of 256 frames and more are processed as two parts on two streams; every frame
    must still equal its single-frame result (checked at the part boundary and the ends)."""
    params = (1000, 1.2, 8, 20, 7)
    distinct = [orbx.textured_frame(900 + i, 752, 480) for i in range(5)]
    B = 261
    frames = [distinct[i % 5] for i in range(B)]
    ex = orbx.ORBextractor(*params, max_batch=B)
    res = ex.extract_batch(frames, (0, 1000))
    single = orbx.ORBextractor(*params)
    ref = [single(d, None, (0, 1000)) for d in distinct]
    for f in (0, 1, 129, 130, 131, 259, 260):
        mono, k, d = res[f]
        rmono, rk, rd = ref[f % 5]
        assert mono == rmono and np.array_equal(k, rk) and np.array_equal(d, rd), f


@pytest.mark.needs_real_device
def test_large_image_batch_splits_by_work():
    """A device-resident batch is cut into two parts when each part can fill the chip: 128 frames of
    752 x 480, fewer of a larger image in proportion (23 at 1920 x 1080).  Frames on both sides of
    the cut equal their single-frame results."""
    import torch
    from tests.conftest import TORCH_DEVICE
    params = (3000, 1.2, 8, 20, 7)
    w, h, B = 1920, 1080, 48
    distinct = [orbx.textured_frame(950 + i, w, h) for i in range(3)]
    frames = np.stack([distinct[i % 3] for i in range(B)])
    ex = orbx.ORBextractor(*params, max_batch=B)
    d = torch.from_numpy(frames).to(TORCH_DEVICE)
    ex.extract_batch_device(d.data_ptr(), B, w, h, w, (0, 0))
    assert [fr for _, fr in ex.detect_launch_times()] == [24, 24]     # one detection launch per part
    single = orbx.ORBextractor(*params)
    ref = [single(f, None, (0, 0)) for f in distinct]
    for f in (0, 1, 23, 24, 25, 46, 47):
        mono, k, desc = ex.download(f)
        rmono, rk, rd = ref[f % 3]
        assert mono == rmono and np.array_equal(k, rk) and np.array_equal(desc, rd), f


def _pinned_array(shape, dtype):
    """numpy array over page-locked host memory from the library (orbx_host_alloc)."""
    import ctypes as C
    from orb_slam3_amd._lib import lib, check
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    check(lib().orbx_host_alloc(nbytes, C.byref(p)), "orbx_host_alloc")
    buf = (C.c_uint8 * nbytes).from_address(p.value)
    return np.frombuffer(buf, dtype).reshape(shape), p


@pytest.mark.parametrize("pinned", [False, True])
def test_host_batch_pipeline_groups(oracle, pinned, monkeypatch):
    """orbx_extract_batch over several groups of frames (transfers and kernels of neighbouring
    groups overlap; group size forced small so that 11 frames make 4 groups, the last one short):
    pageable caller memory (staged by host threads) and page-locked caller memory (copied
    directly), contiguous and strided rows - every frame equals the oracle."""
    import ctypes as C
    from orb_slam3_amd._lib import lib, check, KEYPOINT_DTYPE
    monkeypatch.setenv("ORBX_HOST_GROUP", "3")
    W, H, B, stride = 376, 240, 11, 384
    frames = [orbx.textured_frame(900 + i, W, H) for i in range(B)]
    ex = orbx.ORBextractor(400, 1.2, 6, 20, 7, max_batch=B)
    cap = ex.max_keypoints()
    holds = []
    if pinned:
        imgs, h0 = _pinned_array((B, H, stride), np.uint8)
        kps, h1 = _pinned_array((B, cap), KEYPOINT_DTYPE)
        desc, h2 = _pinned_array((B, cap, 32), np.uint8)
        holds = [h0, h1, h2]
    else:
        imgs = np.zeros((B, H, stride), np.uint8)
        kps = np.zeros((B, cap), KEYPOINT_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
    imgs[:] = 0x5A
    for f in range(B):
        imgs[f, :, :W] = frames[f]
    n = np.zeros(B, np.int32)
    mono = np.zeros(B, np.int32)
    i32p = C.POINTER(C.c_int32)
    for use_stride in (stride, W):
        if use_stride == W:
            src = imgs.reshape(-1)[:B * H * W].reshape(B, H, W)     # same memory, rows packed
            for f in range(B):
                src[f] = frames[f]
        else:
            src = imgs
        ptrs = (C.c_void_p * B)(*[src[i].ctypes.data for i in range(B)])
        kps[:] = np.zeros((), KEYPOINT_DTYPE)
        desc[:] = 0
        check(lib().orbx_extract_batch(ex._h, B, ptrs, W, H, use_stride, 0, 1000, kps.ctypes.data_as(C.c_void_p),
                                       desc.ctypes.data_as(C.c_void_p), cap, n.ctypes.data_as(i32p),
                                       mono.ctypes.data_as(i32p)), "orbx_extract_batch")
        for f in range(B):
            st, rmono, rk, rd = oracle.Extractor(400, 1.2, 6, 20, 7).extract(frames[f], (0, 1000))
            assert st == 0 and n[f] == len(rk) and mono[f] == rmono, (f, use_stride)
            k = kps[f, :n[f]]
            for fld in ("x", "y", "size", "response", "octave", "class_id"):
                assert np.array_equal(k[fld], rk[fld]), (f, fld)
            assert np.array_equal(desc[f, :n[f]], rd), f
    ex.close()
    from orb_slam3_amd._lib import lib as _l
    for h in holds:
        _l().orbx_host_free(h)


@pytest.mark.parametrize("pinned", [False, True])
def test_single_frame_host_call(oracle, pinned):
    """orbx_extract, the call Frame::ExtractORB makes: one frame from page-locked or ordinary
    memory, packed or strided rows, results equal to the oracle; an output array that is too small
    gives the count and ORBX_ERR_CAPACITY."""
    import ctypes as C
    from orb_slam3_amd._lib import lib, check, KEYPOINT_DTYPE, ORBX_ERR_CAPACITY
    W, H, stride = 376, 240, 400
    frame = orbx.textured_frame(77, W, H)
    st, rmono, rk, rd = oracle.Extractor(400, 1.2, 6, 20, 7).extract(frame, (100, 250))
    assert st == 0
    ex = orbx.ORBextractor(400, 1.2, 6, 20, 7)
    cap = ex.max_keypoints()
    holds = []
    if pinned:
        img, h0 = _pinned_array((H, stride), np.uint8)
        holds.append(h0)
    else:
        img = np.zeros((H, stride), np.uint8)
    img[:] = 0xA5
    img[:, :W] = frame
    packed = np.ascontiguousarray(frame)
    n, mono = C.c_int32(), C.c_int32()
    for src, use_stride in ((img, stride), (packed, W)):
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        check(lib().orbx_extract(ex._h, C.c_void_p(src.ctypes.data), W, H, use_stride, 100, 250,
                                 kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, C.byref(n),
                                 C.byref(mono)), "orbx_extract")
        assert n.value == len(rk) and mono.value == rmono, use_stride
        for fld in ("x", "y", "size", "response", "octave", "class_id"):
            assert np.array_equal(kps[fld][:n.value], rk[fld]), (use_stride, fld)
        assert np.array_equal(desc[:n.value], rd), use_stride
    small = max(1, len(rk) // 2)
    kps = np.zeros(small, KEYPOINT_DTYPE)
    desc = np.zeros((small, 32), np.uint8)
    status = lib().orbx_extract(ex._h, C.c_void_p(packed.ctypes.data), W, H, W, 100, 250, kps.ctypes.data_as(C.c_void_p),
                                desc.ctypes.data_as(C.c_void_p), small, C.byref(n), C.byref(mono))
    assert status == ORBX_ERR_CAPACITY and n.value == len(rk)
    ex.close()
    for h in holds:
        lib().orbx_host_free(h)


@pytest.mark.parametrize("B", [2, 3, 5, 8, 9])
def test_host_call_with_a_few_frames(oracle, B):
    """orbx_extract_batch with a handful of frames (a stereo pair, a rig) takes the path without
    copy engines up to eight frames and the grouped path beyond; rows wider than the image, a
    different image per frame, then a capacity that one frame exceeds: that frame reports its
    count without data, the call returns ORBX_ERR_CAPACITY and the other frames are complete."""
    import ctypes as C
    from orb_slam3_amd._lib import lib, check, KEYPOINT_DTYPE, ORBX_ERR_CAPACITY
    W, H, stride = 376, 240, 384
    params = (300, 1.2, 5, 20, 7)
    frames = [orbx.textured_frame(500 + i, W, H) for i in range(B)]
    frames[1] = adversarial("noise", W, H)      # far more corners than the others
    want, single = [], []
    one = orbx.ORBextractor(*params)
    for img in frames:
        st, rmono, rk, rd = oracle.Extractor(*params).extract(img, (50, 300))
        assert st == 0
        want.append((rmono, rk, rd))
        single.append(one(img, None, (50, 300)))     # angles and descriptor bits: the library's own single-frame result
    wide = np.full((B, H, stride), 0x5A, np.uint8)
    for f in range(B):
        wide[f, :, :W] = frames[f]
    ptrs = (C.c_void_p * B)(*[wide[f].ctypes.data for f in range(B)])
    ex = orbx.ORBextractor(*params, max_batch=B)
    i32p = C.POINTER(C.c_int32)
    lens = [len(w[1]) for w in want]
    for cap in (ex.max_keypoints(), max(lens) - 1):
        kps = np.zeros((B, cap), KEYPOINT_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        mono = np.zeros(B, np.int32)
        status = lib().orbx_extract_batch(ex._h, B, ptrs, W, H, stride, 50, 300, kps.ctypes.data_as(C.c_void_p),
                                          desc.ctypes.data_as(C.c_void_p), cap, n.ctypes.data_as(i32p),
                                          mono.ctypes.data_as(i32p))
        assert status == (ORBX_ERR_CAPACITY if cap < max(lens) else 0)
        assert list(n) == lens and list(mono) == [w[0] for w in want]
        for f in range(B):
            if lens[f] > cap:
                continue
            for fld in ("x", "y", "size", "response", "octave", "class_id"):
                assert np.array_equal(kps[fld][f, :lens[f]], want[f][1][fld]), (cap, f, fld)
            assert np.array_equal(kps[f, :lens[f]], single[f][1]), (cap, f)
            assert np.array_equal(desc[f, :lens[f]], single[f][2]), (cap, f)
    ex.close()


def test_two_part_batches_back_to_back(oracle):
    """Batches large enough to run as two parts on two streams, enqueued back to back without a
    synchronisation in between (the second part of one batch may still run while the first part of
    the next one starts), with different images per batch, a differently split batch in between,
    and downloads right after the last call: every downloaded frame equals the oracle's result
    for the image that the LAST call put at that position."""
    import torch
    from tests.conftest import TORCH_DEVICE
    W, H, B = 376, 240, 256
    params = (300, 1.2, 5, 20, 7)
    base = [orbx.textured_frame(4000 + i, W, H) for i in range(6)]
    want = {}
    for i, img in enumerate(base):
        st, rmono, rk, rd = oracle.Extractor(*params).extract(img, (0, 1000))
        assert st == 0
        want[i] = (rmono, rk, rd)

    def batch(shift, n):
        idx = [(f + shift) % len(base) for f in range(n)]
        return idx, torch.from_numpy(np.stack([base[i] for i in idx])).to(TORCH_DEVICE)

    ex = orbx.ORBextractor(*params, max_batch=B)
    ia, da = batch(0, B)
    ib, db = batch(1, B)
    ic, dc = batch(2, 160)        # split differently (2 x 80): must wait for the previous second part
    idd, dd = batch(3, B)
    for d, n in ((da, B), (db, B), (dc, 160), (dd, B), (da, B), (db, B)):
        ex.extract_batch_device(d.data_ptr(), n, W, H, W, (0, 1000))
    for f in (0, 1, 127, 128, 129, 200, 255):
        mono, k, dsc = ex.download(f)
        rmono, rk, rd = want[ib[f]]
        assert mono == rmono and len(k) == len(rk), f
        for fld in ("x", "y", "size", "response", "octave"):
            assert np.array_equal(k[fld], rk[fld]), (f, fld)
        assert np.array_equal(dsc, rd), f
    ex.close()
