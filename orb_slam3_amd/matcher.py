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


def knn_match2_sharded(query_t, train_shard_t, shard_offset, group=None):
    """Brute-force 2-NN against a train set sharded over the ranks of a
    torch.distributed group.  query_t: (nq, 32) uint8 tensor replicated on every
    rank; train_shard_t: this rank's contiguous slice of the train rows, whose
    first row has global index shard_offset.  Every rank returns the global
    (idx, dist) tensors.  One all-gather of the per-shard top-2 records is the
    only exchange; on GPUs it runs over RCCL, and the merge is a HIP kernel.
    On CPU tensors (gloo) the same exchange is followed by a numpy merge of the
    records - that path exists to test the sharding logic without a GPU and
    still needs the HIP library for the per-shard search."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nq = query_t.shape[0]
    on_gpu = query_t.is_cuda
    if on_gpu:
        dev = query_t.device
        idx = torch.empty((nq, 2), dtype=torch.int32, device=dev)
        dst = torch.empty((nq, 2), dtype=torch.int16, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        check(lib().orbx_knn2_device(C.c_void_p(query_t.data_ptr()), nq,
                                     C.c_void_p(train_shard_t.data_ptr()), train_shard_t.shape[0],
                                     int(shard_offset), C.c_void_p(idx.data_ptr()),
                                     C.c_void_p(dst.data_ptr()), C.c_void_p(stream)), "orbx_knn2_device")
    else:
        i_np, d_np = knn_match2(query_t.numpy(), train_shard_t.numpy())
        have = i_np != 0xFFFFFFFF
        i_np = np.where(have, i_np + np.uint32(shard_offset), i_np)
        idx = torch.from_numpy(i_np.view(np.int32))
        dst = torch.from_numpy(d_np.view(np.int16))
    if world == 1:
        return idx, dst
    idx_all = torch.empty((world, nq, 2), dtype=idx.dtype, device=idx.device)
    dst_all = torch.empty((world, nq, 2), dtype=dst.dtype, device=dst.device)
    dist.all_gather_into_tensor(idx_all, idx, group=group)
    dist.all_gather_into_tensor(dst_all, dst, group=group)
    if on_gpu:
        out_i = torch.empty_like(idx)
        out_d = torch.empty_like(dst)
        stream = torch.cuda.current_stream(idx.device).cuda_stream
        check(lib().orbx_top2_merge_device(C.c_void_p(idx_all.data_ptr()), C.c_void_p(dst_all.data_ptr()),
                                           world, nq, C.c_void_p(out_i.data_ptr()),
                                           C.c_void_p(out_d.data_ptr()), C.c_void_p(stream)),
              "orbx_top2_merge_device")
        return out_i, out_d
    out_i, out_d = merge_top2_host(idx_all.numpy().view(np.uint32), dst_all.numpy().view(np.uint16))
    return torch.from_numpy(out_i.view(np.int32)), torch.from_numpy(out_d.view(np.int16))


def merge_top2_host(idx_parts, dist_parts):
    """Host restatement of the top-2 merge: parts [P, nq, 2] -> global [nq, 2]
    ordered by (distance, index)."""
    P, nq, _ = idx_parts.shape
    key = (dist_parts.astype(np.uint64) << np.uint64(32)) | idx_parts.astype(np.uint64)
    key = np.transpose(key, (1, 0, 2)).reshape(nq, 2 * P)
    key.sort(axis=1)
    best = key[:, :2]
    return (best & np.uint64(0xFFFFFFFF)).astype(np.uint32), (best >> np.uint64(32)).astype(np.uint16)
