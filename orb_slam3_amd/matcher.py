"""Hamming matching: ORBmatcher::DescriptorDistance and the two descriptor
matching loops of Frame (core/lib/ORB/matcher.cc:1961-1975,
core/src/Frame.cc:840-908 and 1177-1185)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KEYPOINT_DTYPE, check, lib, ptr

TH_HIGH = 100  # core/lib/ORB/matcher.cc:34
TH_LOW = 50    # core/lib/ORB/matcher.cc:35


def descriptor_distance(a, b):
    """ORBmatcher::DescriptorDistance: Hamming distance of two 32-byte descriptors."""
    a = np.ascontiguousarray(a, np.uint8).reshape(32)
    b = np.ascontiguousarray(b, np.uint8).reshape(32)
    return int(lib().orbx_hamming(ptr(a), ptr(b)))


def knn_match2(query, train, device=0):
    """BFMatcher(NORM_HAMMING).knnMatch(query, train, k=2) -> (idx[nq,2] uint32, dist[nq,2] uint16).

    Neighbours are ordered by (distance, train index); missing ones (fewer than
    two train rows) are idx 0xFFFFFFFF / dist 0xFFFF."""
    q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
    idx = np.zeros((q.shape[0], 2), np.uint32)
    dist = np.zeros((q.shape[0], 2), np.uint16)
    check(lib().orbx_knn2(int(device), ptr(q), q.shape[0], ptr(t), t.shape[0], ptr(idx), ptr(dist)),
          "orbx_knn2")
    return idx, dist


def ratio_test(idx, dist, ratio=0.7):
    """Lowe ratio test of core/src/Frame.cc:1185: both neighbours present and
    d0 < d1 * 0.7 (double comparison).  Returns a boolean mask."""
    have2 = idx[:, 1] != 0xFFFFFFFF
    return have2 & (dist[:, 0].astype(np.float64) < dist[:, 1].astype(np.float32).astype(np.float64) * ratio)


def stereo_band_match(kps_l, desc_l, kps_r, desc_r, scale_factors, n_rows, min_d, max_d, device=0):
    """Descriptor search of Frame::ComputeStereoMatches (Frame.cc:840-908).
    Returns (best_idx int32[nL], best_dist int32[nL]); -1 / 100 when nothing qualified."""
    kl = np.ascontiguousarray(kps_l, KEYPOINT_DTYPE)
    kr = np.ascontiguousarray(kps_r, KEYPOINT_DTYPE)
    dl = np.ascontiguousarray(desc_l, np.uint8).reshape(-1, 32)
    dr = np.ascontiguousarray(desc_r, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    bi = np.zeros(kl.shape[0], np.int32)
    bd = np.zeros(kl.shape[0], np.int32)
    check(lib().orbx_stereo_band_match(int(device), ptr(kl), ptr(dl), kl.shape[0], ptr(kr), ptr(dr),
                                       kr.shape[0], sf.ctypes.data_as(C.POINTER(C.c_float)), sf.shape[0],
                                       int(n_rows), float(min_d), float(max_d), ptr(bi), ptr(bd)),
          "orbx_stereo_band_match")
    return bi, bd


def compute_stereo_matches_batch(ex_left, ex_right, n_frames, mbf, mb):
    """Frame::ComputeStereoMatches (core/src/Frame.cc:831-996) for frames [0, n_frames) of the
    last batched extractions on the two ORBextractor handles; everything stays on the device.
    Asynchronous; fetch results with stereo_download()."""
    check(lib().orbx_stereo_match_batch_device(ex_left._h, ex_right._h, int(n_frames), float(mbf),
                                               float(mb)), "orbx_stereo_match_batch_device")


def lapping_knn_match2(ex_left, ex_right, frame=0):
    """The knnMatch of Frame::ComputeStereoFishEyeMatches (core/src/Frame.cc:1159-1177) on two
    device-resident extractions: 2-NN of the left lapping-area descriptors against the right ones.
    -> (idx[nq, 2], dist[nq, 2], mono_left, mono_right); idx are trainIdx (relative to mono_right)."""
    cap = ex_left.max_keypoints()
    idx = np.empty((cap, 2), np.uint32)
    dist = np.empty((cap, 2), np.uint16)
    nq, ml, nt, mr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    check(lib().orbx_lapping_knn2(ex_left._h, ex_right._h, int(frame), ptr(idx), ptr(dist), cap, C.byref(nq),
                                  C.byref(ml), C.byref(nt), C.byref(mr)), "orbx_lapping_knn2")
    return idx[:nq.value].copy(), dist[:nq.value].copy(), ml.value, mr.value


def lapping_knn_match2_batch(ex_left, ex_right, n_frames):
    """lapping_knn_match2 for frames [0, n_frames) of the two last batched extractions in one
    launch; results stay on the device.  Asynchronous; fetch with lapping_knn_download()."""
    check(lib().orbx_lapping_knn2_batch_device(ex_left._h, ex_right._h, int(n_frames)),
          "orbx_lapping_knn2_batch_device")


def lapping_knn_download(ex_left, frame):
    """One frame's results of lapping_knn_match2_batch, as lapping_knn_match2 returns them."""
    cap = ex_left.max_keypoints()
    idx = np.empty((cap, 2), np.uint32)
    dist = np.empty((cap, 2), np.uint16)
    nq, ml, nt, mr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    check(lib().orbx_lapping_knn2_download(ex_left._h, int(frame), ptr(idx), ptr(dist), cap, C.byref(nq),
                                           C.byref(ml), C.byref(nt), C.byref(mr)), "orbx_lapping_knn2_download")
    return idx[:nq.value].copy(), dist[:nq.value].copy(), ml.value, mr.value


def lapping_ratio_test_batch(ex_left, ratio=0.7):
    """The ratio test of core/src/Frame.cc:1185 on the records lapping_knn_match2_batch left on the
    device, for every frame of that batch in one launch.  Asynchronous; fetch with
    lapping_matches_download()."""
    check(lib().orbx_lapping_ratio_test_batch_device(ex_left._h, float(ratio)),
          "orbx_lapping_ratio_test_batch_device")


def lapping_matches_download(ex_left, frame):
    """One frame's verdicts of lapping_ratio_test_batch: int32[nq], the nearest neighbour's
    trainIdx where the query passed the ratio test, else -1."""
    cap = ex_left.max_keypoints()
    match = np.empty(cap, np.int32)
    nq, ng = C.c_int32(), C.c_int32()
    check(lib().orbx_lapping_matches_download(ex_left._h, int(frame), ptr(match), cap, C.byref(nq), C.byref(ng)),
          "orbx_lapping_matches_download")
    assert ng.value == int((match[:nq.value] >= 0).sum())
    return match[:nq.value].copy()


def ratio_test_device(idx_t, dist_t, ratio=0.7):
    """The same test on device-resident records (torch CUDA tensors in the layout knn_match2_sharded
    returns): int32[nq] tensor of nearest train indices, -1 where the test fails."""
    import torch
    nq = idx_t.shape[0]
    out = torch.empty(nq, dtype=torch.int32, device=idx_t.device)
    stream = torch.cuda.current_stream(idx_t.device).cuda_stream if idx_t.is_cuda else 0
    check(lib().orbx_ratio_test_device(C.c_void_p(idx_t.data_ptr()), C.c_void_p(dist_t.data_ptr()), nq, float(ratio),
                                       C.c_void_p(out.data_ptr()), C.c_void_p(stream)), "orbx_ratio_test_device")
    return out


def compute_stereo_from_rgbd_batch(ex, n_frames, d_depth_ptr, stride_floats, mbf):
    """Frame::ComputeStereoFromRGBD (core/src/Frame.cc:998-1016) for frames [0, n_frames) of the
    last extraction + grid.frame_grid_batch on ex; d_depth_ptr = device address of the float depth
    images.  Asynchronous; fetch results with stereo_download()."""
    check(lib().orbx_stereo_from_rgbd_batch_device(ex._h, int(n_frames), C.c_void_p(int(d_depth_ptr)),
                                                   int(stride_floats), float(mbf)),
          "orbx_stereo_from_rgbd_batch_device")


def stereo_download(ex_left, frame, with_l1=False):
    """(mvuRight, mvDepth[, L1 distance]) of one frame of the last stereo call."""
    cap = ex_left.max_keypoints()
    ur = np.empty(cap, np.float32)
    dp = np.empty(cap, np.float32)
    l1 = np.empty(cap, np.int32) if with_l1 else None
    n = C.c_int32()
    check(lib().orbx_stereo_download(ex_left._h, int(frame), ptr(ur), ptr(dp), ptr(l1), cap, C.byref(n)),
          "orbx_stereo_download")
    out = (ur[:n.value].copy(), dp[:n.value].copy())
    return out + (l1[:n.value].copy(),) if with_l1 else out


def compute_stereo_matches(ex_left, ex_right, mbf, mb):
    """Single stereo frame: call after ex_left(image_left) and ex_right(image_right).
    Returns (mvuRight, mvDepth) as float32 arrays, -1 where there is no match."""
    cap = ex_left.max_keypoints()
    ur = np.empty(cap, np.float32)
    dp = np.empty(cap, np.float32)
    n = C.c_int32()
    check(lib().orbx_stereo_match(ex_left._h, ex_right._h, float(mbf), float(mb), ptr(ur), ptr(dp), cap,
                                  C.byref(n)), "orbx_stereo_match")
    return ur[:n.value].copy(), dp[:n.value].copy()


def shard_bounds(n_rows, world):
    """Contiguous train-row ranges [b[r], b[r+1]) per rank, starts aligned to 8 rows.
    Contiguous ranges keep global index order = (rank, local index) order, so merging
    per-shard results in rank order reproduces the single-GPU tie-break."""
    b = [(n_rows * r // world) // 8 * 8 for r in range(world)] + [n_rows]
    return b


class RcclComm:
    """An RCCL communicator of this process's own for the sharded matcher, one rank per GPU.
    The 128-byte unique id is made by rank 0 through the library and handed to the other ranks
    over the torch.distributed group (any backend: the hand-over is host data)."""

    def __init__(self, device_index, group=None):
        import torch
        import torch.distributed as dist
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        uid = np.zeros(128, np.uint8)
        st = lib().orbx_rccl_unique_id(ptr(uid)) if self.rank == 0 else 0
        if self.world > 1:
            # the status travels with the id, so that a failure on rank 0 is raised on every rank
            # instead of leaving the others waiting for an id that never comes
            box = [(int(st), uid.tobytes())]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                       group=group)
            st, raw = box[0]
            uid = np.frombuffer(raw, np.uint8).copy()
        check(st, "orbx_rccl_unique_id")
        h = C.c_void_p()
        check(lib().orbx_rccl_comm_create(ptr(uid), self.world, self.rank, int(device_index), C.byref(h)),
              "orbx_rccl_comm_create")
        self._h = h
        self._ws = None   # workspace tensor, grown on demand

    def workspace(self, nq, device):
        import torch
        need = int(lib().orbx_knn2_sharded_workspace_bytes(self.world, int(nq)))
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
        return self._ws, need

    def close(self):
        if getattr(self, "_h", None):
            lib().orbx_rccl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


_comms = {}


def _comm_for(group, device):
    key = (id(group) if group is not None else None, device.index)
    if key not in _comms:
        _comms[key] = RcclComm(device.index, group)
    return _comms[key]


def knn_match2_sharded(query_t, train_shard_t, shard_offset, group=None, comm=None, _hooks=None):
    """Brute-force 2-NN against a train set sharded over the ranks of a
    torch.distributed group (one process per GPU, RCCL over xGMI).

    query_t: (nq, 32) uint8 CUDA tensor replicated on every rank; train_shard_t: this
    rank's contiguous slice of the train rows, whose first row has global index
    shard_offset.  Every rank returns the global (idx int32[nq,2], dist int16[nq,2])
    tensors (bit patterns of uint32 / uint16).  CUDA tensors go through the C entry point
    orbx_knn2_sharded (per-shard search, one RCCL all-gather of the per-shard top-2 records,
    merge kernel) on an RcclComm that is created on first use per (group, device) or passed
    in as `comm`.

    _hooks = (local_search, merge) replaces the two HIP calls and moves the records with the
    group's own all-gather; it exists so the world_size-2 gloo test can drive the sharding,
    offsets and merge order on CPU tensors.  Without hooks CPU tensors are rejected: there is
    no CPU implementation in this package."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nq = query_t.shape[0]
    if _hooks is None:
        if not query_t.is_cuda or not train_shard_t.is_cuda:
            raise RuntimeError("knn_match2_sharded needs CUDA tensors (no CPU fallback)")
        dev = query_t.device
        q, t = query_t.contiguous(), train_shard_t.contiguous()
        c = comm if comm is not None else _comm_for(group, dev)
        ws, need = c.workspace(nq, dev)
        idx = torch.empty((nq, 2), dtype=torch.int32, device=dev)
        dst = torch.empty((nq, 2), dtype=torch.int16, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        check(lib().orbx_knn2_sharded(c._h, C.c_void_p(q.data_ptr()), nq, C.c_void_p(t.data_ptr()), t.shape[0],
                                      int(shard_offset), C.c_void_p(idx.data_ptr()), C.c_void_p(dst.data_ptr()),
                                      C.c_void_p(ws.data_ptr()), need, C.c_void_p(stream)), "orbx_knn2_sharded")
        return idx, dst
    local_search, merge = _hooks
    # Same protocol as the C entry point: a rank whose search fails still enters the exchange,
    # with "no neighbour" records and its status beside them, and every rank raises the worst one.
    status = 0
    try:
        idx, dst = local_search(query_t.contiguous(), train_shard_t.contiguous(), shard_offset)
    except _lib.OrbxError as e:
        status = int(e.status)
    except Exception:
        status = _lib.ORBX_ERR_HIP
    if status != 0:
        idx = torch.full((nq, 2), -1, dtype=torch.int32)
        dst = torch.full((nq, 2), -1, dtype=torch.int16)
    if world == 1:
        check(status, "knn_match2_sharded")
        return idx, dst
    idx_all = torch.empty((world * nq, 2), dtype=idx.dtype, device=idx.device)
    # 16-bit integers are not a collective dtype of gloo: ship the bytes
    dst_all = torch.empty((world * nq, 4), dtype=torch.uint8, device=dst.device)
    st_all = torch.empty(world, dtype=torch.int32, device=idx.device)
    dist.all_gather_into_tensor(idx_all, idx.contiguous(), group=group)
    dist.all_gather_into_tensor(dst_all, dst.contiguous().view(torch.uint8), group=group)
    dist.all_gather_into_tensor(st_all, torch.tensor([status], dtype=torch.int32, device=idx.device), group=group)
    check(int(st_all.min().item()), "knn_match2_sharded (worst status over the ranks)")
    return merge(idx_all.view(world, nq, 2), dst_all.view(torch.int16).view(world, nq, 2))
