// Brute-force Hamming 2-NN against a train set sharded over the GPUs of a node: the one step
// of the path that exchanges data between GPUs (core/src/Frame.cc:1177 scaled up to a keyframe
// descriptor database).  Every rank searches its own contiguous range of train rows
// (orbx_knn2_device with the range's global offset), one all-gather over RCCL moves the
// per-rank top-2 records (12 bytes per query) to every rank, and k_top2_merge reduces them in
// rank order - which is global index order, so the (distance, index) tie-break of the unsharded
// search is reproduced exactly.
//
// RCCL is resolved at run time from the copy already loaded in the process (PyTorch ships one)
// or from librccl.so.1 on the loader path; the library itself carries no link-time dependency
// on it, and single-GPU users never load it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>

#include <cstring>
#include <mutex>

#include "../../include/orbx.h"

namespace orbx {
void launch_top2_merge_strided(const uint8_t* idx_parts, size_t idx_stride, const uint8_t* dist_parts,
                               size_t dist_stride, int nparts, int nq, uint32_t* idx, uint16_t* dist,
                               hipStream_t s);
}

namespace {

// the part of RCCL's C interface (rccl.h, identical to nccl.h) used here
struct UniqueId { char internal[128]; };
typedef void* Comm;
const int kNcclSuccess = 0;
const int kNcclUint8 = 1;   // ncclUint8 in ncclDataType_t

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*CommCount)(Comm, int*) = nullptr;
  int (*CommUserRank)(Comm, int*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  bool ok = false;
};

std::once_flag g_once;
Rccl g_rccl;

const Rccl& rccl() {
  std::call_once(g_once, [] {
    Rccl& r = g_rccl;
    // symbols of a copy that is already in the process win (one RCCL per process)
    void* h = dlopen(nullptr, RTLD_NOW);
    if (!h || !dlsym(h, "ncclAllGather")) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h) return;
    r.handle = h;
    r.GetUniqueId = (int (*)(UniqueId*))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(Comm*, int, UniqueId, int))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (int (*)(Comm))dlsym(h, "ncclCommDestroy");
    r.CommCount = (int (*)(Comm, int*))dlsym(h, "ncclCommCount");
    r.CommUserRank = (int (*)(Comm, int*))dlsym(h, "ncclCommUserRank");
    r.AllGather = (int (*)(const void*, void*, size_t, int, Comm, hipStream_t))dlsym(h, "ncclAllGather");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.CommUserRank && r.AllGather;
  });
  return g_rccl;
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }
// One rank's block of the exchange: nq*2 indices, then nq*2 distances, then a header whose first
// word is the rank's status (0 = its records are valid, else the negated orbx_status of its
// local search in every byte).  The status travels with the records so that a rank whose search failed still
// takes part in the collective and every rank learns of the failure.
size_t block_dist_off(int nq) { return align256((size_t)nq * 8); }
size_t block_status_off(int nq) { return block_dist_off(nq) + align256((size_t)nq * 4); }
size_t block_bytes(int nq) { return block_status_off(nq) + 256; }

}  // namespace

extern "C" {

int orbx_rccl_unique_id(uint8_t* id128) {
  if (!id128) return ORBX_ERR_BAD_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return ORBX_ERR_COMM;
  UniqueId u;
  if (r.GetUniqueId(&u) != kNcclSuccess) return ORBX_ERR_COMM;
  std::memcpy(id128, u.internal, sizeof(u.internal));
  return ORBX_OK;
}

int orbx_rccl_comm_create(const uint8_t* id128, int nranks, int rank, int device_id, void** comm) {
  if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return ORBX_ERR_BAD_ARG;
  *comm = nullptr;
  const Rccl& r = rccl();
  if (!r.ok) return ORBX_ERR_COMM;
  if (hipSetDevice(device_id) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  UniqueId u;
  std::memcpy(u.internal, id128, sizeof(u.internal));
  Comm c = nullptr;
  if (r.CommInitRank(&c, nranks, u, rank) != kNcclSuccess || !c) return ORBX_ERR_COMM;
  *comm = c;
  return ORBX_OK;
}

int orbx_rccl_comm_destroy(void* comm) {
  if (!comm) return ORBX_OK;
  const Rccl& r = rccl();
  if (!r.ok) return ORBX_ERR_COMM;
  return r.CommDestroy(comm) == kNcclSuccess ? ORBX_OK : ORBX_ERR_COMM;
}

size_t orbx_knn2_sharded_workspace_bytes(int nranks, int nq) {
  if (nranks < 1 || nq < 0) return 0;
  return (size_t)nranks * block_bytes(nq);
}

int orbx_knn2_sharded(void* nccl_comm, const uint8_t* d_q, int nq, const uint8_t* d_t_shard, int64_t nt_shard,
                      uint32_t shard_offset, uint32_t* d_idx, uint16_t* d_dist, void* d_workspace,
                      size_t workspace_bytes, void* stream) {
  // Failures up to here leave before any collective; they concern the communicator itself (a
  // rank without a usable one cannot take part in anything) or nq, which the contract makes
  // equal on every rank.
  if (!nccl_comm || nq < 0) return ORBX_ERR_BAD_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return ORBX_ERR_COMM;
  int nranks = 0, rank = -1;
  if (r.CommCount(nccl_comm, &nranks) != kNcclSuccess || r.CommUserRank(nccl_comm, &rank) != kNcclSuccess ||
      nranks < 1 || rank < 0 || rank >= nranks)
    return ORBX_ERR_COMM;
  if (nq == 0) return ORBX_OK;   // nothing to exchange either: every rank sees the same nq
  hipStream_t s = (hipStream_t)stream;
  const size_t blk = block_bytes(nq), need = (size_t)nranks * blk;
  const size_t oDist = block_dist_off(nq), oStatus = block_status_off(nq);
  // Everything that can differ from rank to rank is only recorded from here on: this rank enters
  // the all-gather whatever happened, with sentinel records and its status in the block header,
  // so that no peer is left waiting in the collective.
  int local = ORBX_OK;
  if (!d_q || !d_idx || !d_dist || nt_shard < 0 || (nt_shard > 0 && !d_t_shard)) local = ORBX_ERR_BAD_ARG;
  uint8_t* ws = (uint8_t*)d_workspace;
  bool own = false;
  if (ws && (workspace_bytes < need || ((uintptr_t)ws & 255))) {
    local = ORBX_ERR_BAD_ARG;   // unusable as given: exchange through memory of our own instead
    ws = nullptr;
  }
  if (!ws) {
    // without memory for the exchange this rank cannot take part: the one failure that leaves
    // the peers waiting (they need an external timeout or an abort of the communicator)
    if (hipMallocAsync((void**)&ws, need, s) != hipSuccess) { (void)hipGetLastError(); return ORBX_ERR_HIP; }
    own = true;
  }
  // this rank's block sits where the in-place all-gather expects it
  uint8_t* mine = ws + (size_t)rank * blk;
  if (local == ORBX_OK)
    local = orbx_knn2_device(d_q, nq, d_t_shard, nt_shard, shard_offset, (uint32_t*)mine, (uint16_t*)(mine + oDist),
                             stream);
  bool hipBad = false;
  if (local != ORBX_OK) {
    (void)hipGetLastError();
    // "no neighbour" records: index 0xFFFFFFFF, distance 0xFFFF
    hipBad |= hipMemsetAsync(mine, 0xFF, oStatus, s) != hipSuccess;
  }
  // (the status as a byte fill: every orbx_status fits a byte, and a plain memset needs no host source)
  hipBad |= hipMemsetAsync(mine + oStatus, -local, 4, s) != hipSuccess;
  int st = local;
  const bool gathered = r.AllGather(mine, ws, blk, kNcclUint8, nccl_comm, s) == kNcclSuccess;
  if (!gathered) st = ORBX_ERR_COMM;
  if (hipBad && st == ORBX_OK) st = ORBX_ERR_HIP;
  if (gathered) {
    // every rank's status word; the call waits here for the exchange (the merge below is
    // asynchronous again)
    int32_t codes[256];
    const int nread = nranks < 256 ? nranks : 256;
    if (hipMemcpy2DAsync(codes, 4, ws + oStatus, blk, 4, (size_t)nread, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      (void)hipGetLastError();
      if (st == ORBX_OK) st = ORBX_ERR_HIP;
    } else {
      for (int k = 0; k < nread; ++k) {
        const int code = codes[k] & 0xFF;
        if (code != 0 && -code < st) st = -code;   // the worst (most negative) status wins
      }
    }
  }
  if (st == ORBX_OK) {
    orbx::launch_top2_merge_strided(ws, blk, ws + oDist, blk, nranks, nq, d_idx, d_dist, s);
    if (hipGetLastError() != hipSuccess) st = ORBX_ERR_HIP;
  }
  if (own) (void)hipFreeAsync(ws, s);
  return st;
}

}  // extern "C"
