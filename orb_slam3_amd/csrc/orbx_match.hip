// Hamming matching on 256-bit ORB descriptors (gfx950).
//
//   orbx_hamming            ORBmatcher::DescriptorDistance      core/lib/ORB/matcher.cc:1961-1975
//   k_knn2 / k_top2_merge   BFMatcher(NORM_HAMMING).knnMatch(.., 2) as used at core/src/Frame.cc:1177
//   k_stereo_band           descriptor search of Frame::ComputeStereoMatches  core/src/Frame.cc:840-908
//
// The brute-force kernel is VALU-bound by design: per (query, train) pair it
// spends 8 v_xor_b32 + 8 accumulating v_bcnt_u32_b32 and 4 instructions of
// top-2 upkeep.  Every lane keeps QPL queries in registers; the train
// descriptor of the current step is the same for the whole wave, so it is
// fetched through wave-uniform (scalar) loads and costs no vector memory or
// LDS traffic at all.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/orbx.h"

namespace orbx {
void keep_pool_memory(int device);   // orbx_api.hip

#define KNN_QPL 2          // queries per lane
#define KNN_BLOCK 256
#define KNN_IDX_BITS 22    // train rows per scan segment <= 2^22
#define KNN_ROWS 4          // train rows per step of the scan
#ifndef KNN_WANT_WGS
#define KNN_WANT_WGS 8192  // workgroups a matrix-core scan is cut into at least (query blocks x train segments)
#endif

// Candidates are ranked by the 32-bit word (distance << 22 | local train
// index): its unsigned order is exactly (distance, index) ascending, the
// order in which cv::batchDistance keeps neighbours (strict comparisons, the
// earliest index wins a tie).
__global__ __launch_bounds__(KNN_BLOCK) void k_knn2(const uint32_t* __restrict__ q, int nq,
                                                   const uint32_t* __restrict__ t, long long nt,
                                                   int seg_len, uint32_t index_offset,
                                                   uint32_t* __restrict__ out_idx,
                                                   uint16_t* __restrict__ out_dist) {
  const int tid = threadIdx.x;
  const int seg = blockIdx.y;
  const long long t0 = (long long)seg * seg_len;
  const int len = (int)min((long long)seg_len, nt - t0);
  uint32_t qa[KNN_QPL][8];
  uint32_t k0[KNN_QPL], k1[KNN_QPL];
#pragma unroll
  for (int u = 0; u < KNN_QPL; ++u) {
    const int qi = (blockIdx.x * KNN_QPL + u) * KNN_BLOCK + tid;
    const uint4* qp = (const uint4*)(q + (size_t)min(qi, nq - 1) * 8);
    const uint4 a = qp[0], b = qp[1];
    qa[u][0] = a.x; qa[u][1] = a.y; qa[u][2] = a.z; qa[u][3] = a.w;
    qa[u][4] = b.x; qa[u][5] = b.y; qa[u][6] = b.z; qa[u][7] = b.w;
    k0[u] = 0xFFFFFFFFu;
    k1[u] = 0xFFFFFFFFu;
  }
  const uint32_t* tp = t + (size_t)t0 * 8;
  // Steps of KNN_ROWS train rows.  A row can only enter a query's pair when its distance is
  // below the distance of the current second neighbour (rows arrive in ascending index, so an
  // equal distance loses the tie): one compare per pair decides, and the key arithmetic runs
  // only in the steps where some lane of the wave has such a row - after the first few hundred
  // rows of a segment that is a small fraction of the steps.
  uint32_t th[KNN_QPL];
#pragma unroll
  for (int u = 0; u < KNN_QPL; ++u) th[u] = 0xFFFFFFFFu >> KNN_IDX_BITS;
  int j = 0;
  for (; j + KNN_ROWS <= len; j += KNN_ROWS) {
    uint32_t tv[KNN_ROWS][8];
#pragma unroll
    for (int r = 0; r < KNN_ROWS; ++r)
#pragma unroll
      for (int w = 0; w < 8; ++w) tv[r][w] = tp[(size_t)(j + r) * 8 + w];  // wave-uniform -> scalar loads
    uint32_t d[KNN_QPL][KNN_ROWS];
    bool any = false;
#pragma unroll
    for (int u = 0; u < KNN_QPL; ++u)
#pragma unroll
      for (int r = 0; r < KNN_ROWS; ++r) {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) c += __builtin_popcount(qa[u][w] ^ tv[r][w]);
        d[u][r] = c;
        any |= c < th[u];
      }
    if (__any(any)) {
#pragma unroll
      for (int u = 0; u < KNN_QPL; ++u) {
#pragma unroll
        for (int r = 0; r < KNN_ROWS; ++r) {
          const uint32_t key = (d[u][r] << KNN_IDX_BITS) | (uint32_t)(j + r);
          const uint32_t hi = max(k0[u], key);
          k0[u] = min(k0[u], key);
          k1[u] = min(k1[u], hi);
        }
        th[u] = k1[u] >> KNN_IDX_BITS;
      }
    }
  }
  for (; j < len; ++j) {
    uint32_t tv[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) tv[w] = tp[(size_t)j * 8 + w];
#pragma unroll
    for (int u = 0; u < KNN_QPL; ++u) {
      uint32_t c = 0;
#pragma unroll
      for (int w = 0; w < 8; ++w) c += __builtin_popcount(qa[u][w] ^ tv[w]);
      const uint32_t key = (c << KNN_IDX_BITS) | (uint32_t)j;
      const uint32_t hi = max(k0[u], key);
      k0[u] = min(k0[u], key);
      k1[u] = min(k1[u], hi);
    }
  }
#pragma unroll
  for (int u = 0; u < KNN_QPL; ++u) {
    const int qi = (blockIdx.x * KNN_QPL + u) * KNN_BLOCK + tid;
    if (qi < nq) {
      const size_t o = ((size_t)seg * nq + qi) * 2;
      const uint32_t base = index_offset + (uint32_t)t0;
      const uint32_t mask = (1u << KNN_IDX_BITS) - 1u;
      out_idx[o] = k0[u] == 0xFFFFFFFFu ? 0xFFFFFFFFu : base + (k0[u] & mask);
      out_idx[o + 1] = k1[u] == 0xFFFFFFFFu ? 0xFFFFFFFFu : base + (k1[u] & mask);
      out_dist[o] = k0[u] == 0xFFFFFFFFu ? (uint16_t)0xFFFF : (uint16_t)(k0[u] >> KNN_IDX_BITS);
      out_dist[o + 1] = k1[u] == 0xFFFFFFFFu ? (uint16_t)0xFFFF : (uint16_t)(k1[u] >> KNN_IDX_BITS);
    }
  }
}

// ---------------------------------------------------------------------------
// The same scan on the matrix cores.  With every train bit a written as the byte
// a (0 or 1) and every query bit b as the signed byte 1 - 2 b, the dot product
// over the 256 bits is (train bits set where the query's is clear) - (set where
// the query's is set), and adding the query's own popcount gives the Hamming
// distance: an all-pairs search is a 256-deep int8 matrix product, which
// v_mfma_i32_32x32x32_i8 computes at about four times the rate the vector ALU
// reaches with xor + popcount.
//
// A workgroup of four waves owns KM_QPB queries for one train segment.  A wave
// keeps KM_NQT tiles of 32 queries as the B operand in registers for the whole
// scan.  Train rows come in tiles of 32: the workgroup reads a tile's packed bits
// (1 KB), writes them to LDS as bytes in the order the A operand wants them
// (k-step, half, row: a wave's 16-byte reads are contiguous), double-buffered, and
// every wave multiplies the tile with its query tiles.  The 32 k values of an
// instruction may sit in any slot order as long as A and B use the same one (a
// dot product does not care): step s takes 32-bit word s of the descriptors,
// lane half h its bit positions j = 4 h + d (d = 0..3 = dword of the operand),
// byte e of that dword bit 8 e + j - which is what (word >> j) & 0x01010101
// yields, two instructions per four bits.
// In the 32 x 32 result a lane holds one query's products with 16 of the tile's rows
// (row = (i & 3) + 8 (i >> 2) + 4 (l >> 5) for register i), ascending, so the top-2
// upkeep is per lane: one compare per pair decides whether a row can enter (its
// distance is below the lane's current second-best; rows arrive in ascending
// index, so a tie loses), and the key arithmetic runs only in the steps where
// some lane has such a row.  The two lanes of a query merge their pairs at the end.
// ---------------------------------------------------------------------------
#define KM_NQT 2                              // query tiles of 32 per wave
#define KM_WAVES 4
#define KM_SUB 2                              // train tiles of 32 rows per stage
#define KM_QPB (KM_WAVES * KM_NQT * 32)       // queries per workgroup
typedef int km_v4i __attribute__((ext_vector_type(4)));
typedef int km_v16i __attribute__((ext_vector_type(16)));

// bits j0 .. j0 + 3 (+ 8 e for byte e) of a word as four dwords of 0 / 1 bytes
__device__ __forceinline__ km_v4i bits_to_bytes(uint32_t w, int j0) {
  km_v4i r;
  r.x = (int)((w >> j0) & 0x01010101u);
  r.y = (int)((w >> (j0 + 1)) & 0x01010101u);
  r.z = (int)((w >> (j0 + 2)) & 0x01010101u);
  r.w = (int)((w >> (j0 + 3)) & 0x01010101u);
  return r;
}
// 0 / 1 bytes -> +1 / -1 bytes
__device__ __forceinline__ int pm1(int t) { return (int)(((uint32_t)t * 0xFEu) | 0x01010101u); }

// The scan of one workgroup: queries [qblock * KM_QPB, + KM_QPB) of q (nq rows) against the len
// train rows at tp; results (train index = idx_base + row) for query i at out_idx / out_dist + 2 i.
__device__ __forceinline__ void knn2_mfma_scan(const uint32_t* __restrict__ q, int nq, int qblock,
                                               const uint32_t* __restrict__ tp, int len, uint32_t idx_base,
                                               uint32_t* __restrict__ out_idx, uint16_t* __restrict__ out_dist,
                                               uint8_t (*sA)[KM_SUB * 8192]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // ---- the wave's queries as B operands (+1 / -1), and their popcounts --------
  km_v4i bq[KM_NQT][8];
  int qpop[KM_NQT];
#pragma unroll
  for (int qt = 0; qt < KM_NQT; ++qt) {
    const int qi = qblock * KM_QPB + (wave * KM_NQT + qt) * 32 + r;
    const uint4* qp = (const uint4*)(q + (size_t)min(qi, nq - 1) * 8);
    const uint4 a = qp[0], b = qp[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    int pc = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const km_v4i e = bits_to_bytes(w[s], 4 * h);
      bq[qt][s].x = pm1(e.x); bq[qt][s].y = pm1(e.y); bq[qt][s].z = pm1(e.z); bq[qt][s].w = pm1(e.w);
      pc += __builtin_popcount(w[s]);
    }
    qpop[qt] = pc;
  }
  uint32_t k0[KM_NQT], k1[KM_NQT];
  int thd[KM_NQT];   // a row can enter when its product is below this (second-best distance - popcount)
#pragma unroll
  for (int qt = 0; qt < KM_NQT; ++qt) { k0[qt] = 0xFFFFFFFFu; k1[qt] = 0xFFFFFFFFu; thd[qt] = 1024; }
  km_v16i zero;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero[i] = 0;

  // ---- train tiles, KM_SUB of them per stage (one workgroup barrier per stage) ------
  // staging: this thread's row of a tile and word of the row (neighbouring lanes = neighbouring
  // rows, so that their 16-byte LDS writes are contiguous)
  const int srow = tid & 31, sw = tid >> 5;
  const int ntiles = (len + 31) >> 5;
  const int nstages = (ntiles + KM_SUB - 1) / KM_SUB;
  uint32_t pre[KM_SUB];
  auto fetch = [&](int stg) {
#pragma unroll
    for (int u = 0; u < KM_SUB; ++u) {
      const int row = (stg * KM_SUB + u) * 32 + srow;
      pre[u] = row < len ? tp[(size_t)row * 8 + sw] : 0u;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < KM_SUB; ++u) {
      *(km_v4i*)&sA[buf][u * 8192 + ((sw * 2 + 0) * 32 + srow) * 16] = bits_to_bytes(pre[u], 0);
      *(km_v4i*)&sA[buf][u * 8192 + ((sw * 2 + 1) * 32 + srow) * 16] = bits_to_bytes(pre[u], 4);
    }
  };
  if (nstages > 0) { fetch(0); stage(0); }
  __syncthreads();
  for (int stg = 0; stg < nstages; ++stg) {
    const int cur = stg & 1;
    if (stg + 1 < nstages) fetch(stg + 1);
#pragma unroll
    for (int sub = 0; sub < KM_SUB; ++sub) {
      const int tile = stg * KM_SUB + sub;
      const uint8_t* tileA = &sA[cur][sub * 8192];
      km_v16i acc[KM_NQT];
#pragma unroll
      for (int s4 = 0; s4 < 8; s4 += 4) {
        // four A fragments in flight at a time
        km_v4i a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *(const km_v4i*)&tileA[(((s4 + u) * 2 + h) * 32 + r) * 16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int qt = 0; qt < KM_NQT; ++qt)
            acc[qt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[u], bq[qt][s4 + u], s4 + u == 0 ? zero : acc[qt], 0, 0, 0);
      }
      // can any of the lane's 16 rows enter?  (smallest product against the threshold; rows past
      // the segment's end are zeros in LDS and are refused again where the keys are built)
      bool any = false;
#pragma unroll
      for (int qt = 0; qt < KM_NQT; ++qt) {
        int mn = acc[qt][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mn = min(mn, acc[qt][i]);
        any |= mn < thd[qt];
      }
      if (__any(any)) {
        const int rowBase = tile * 32 + 4 * h;
#pragma unroll
        for (int qt = 0; qt < KM_NQT; ++qt) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = rowBase + (i & 3) + 8 * (i >> 2);
            const uint32_t dist = (uint32_t)(qpop[qt] + acc[qt][i]);
            const uint32_t key = row < len ? ((dist << KNN_IDX_BITS) | (uint32_t)row) : 0xFFFFFFFFu;
            const uint32_t hi = max(k0[qt], key);
            k0[qt] = min(k0[qt], key);
            k1[qt] = min(k1[qt], hi);
          }
          thd[qt] = (int)(k1[qt] >> KNN_IDX_BITS) - qpop[qt];
        }
      }
    }
    if (stg + 1 < nstages) stage(cur ^ 1);
    __syncthreads();
  }
  // ---- the two lanes of a query merge their pairs ------------------------------
#pragma unroll
  for (int qt = 0; qt < KM_NQT; ++qt) {
    const uint32_t p0 = __shfl_xor(k0[qt], 32, 64), p1 = __shfl_xor(k1[qt], 32, 64);
    const uint32_t m0 = min(k0[qt], p0);
    const uint32_t m1 = min(max(k0[qt], p0), min(k1[qt], p1));
    const int qi = qblock * KM_QPB + (wave * KM_NQT + qt) * 32 + r;
    if (h == 0 && qi < nq) {
      const size_t o = (size_t)qi * 2;
      const uint32_t mask = (1u << KNN_IDX_BITS) - 1u;
      out_idx[o] = m0 == 0xFFFFFFFFu ? 0xFFFFFFFFu : idx_base + (m0 & mask);
      out_idx[o + 1] = m1 == 0xFFFFFFFFu ? 0xFFFFFFFFu : idx_base + (m1 & mask);
      out_dist[o] = m0 == 0xFFFFFFFFu ? (uint16_t)0xFFFF : (uint16_t)(m0 >> KNN_IDX_BITS);
      out_dist[o + 1] = m1 == 0xFFFFFFFFu ? (uint16_t)0xFFFF : (uint16_t)(m1 >> KNN_IDX_BITS);
    }
  }
}

// grid (query blocks, train segments): segment seg's results at out + seg * nq * 2
__global__ __launch_bounds__(256) void k_knn2_mfma(const uint32_t* __restrict__ q, int nq,
                                                   const uint32_t* __restrict__ t, long long nt,
                                                   int seg_len, uint32_t index_offset,
                                                   uint32_t* __restrict__ out_idx,
                                                   uint16_t* __restrict__ out_dist) {
  __shared__ __align__(16) uint8_t sA[2][KM_SUB * 8192];   // two stages of KM_SUB tiles of 32 rows
  const int seg = blockIdx.y;
  const long long t0 = (long long)seg * seg_len;
  const int len = (int)min((long long)seg_len, nt - t0);
  knn2_mfma_scan(q, nq, (int)blockIdx.x, t + (size_t)t0 * 8, len, index_offset + (uint32_t)t0,
                 out_idx + (size_t)seg * nq * 2, out_dist + (size_t)seg * nq * 2, sA);
}

// grid (query blocks, frames): per frame the rows [mono, n) of the left descriptors against the
// rows [mono, n) of the right ones (the lapping-area subsets of a fisheye pair,
// core/src/Frame.cc:1159-1165); results for left row mono + i at slot frame * capL + i, train
// indices relative to the right frame's mono
__global__ __launch_bounds__(256) void k_knn2_mfma_frames(const uint32_t* __restrict__ descL, int capL,
                                                          const int* __restrict__ nL, const int* __restrict__ monoL,
                                                          const uint32_t* __restrict__ descR, int capR,
                                                          const int* __restrict__ nR, const int* __restrict__ monoR,
                                                          uint32_t* __restrict__ out_idx,
                                                          uint16_t* __restrict__ out_dist) {
  __shared__ __align__(16) uint8_t sA[2][KM_SUB * 8192];
  const int f = blockIdx.y;
  const int ml = monoL[f], mr = monoR[f];
  const int nq = nL[f] - ml, nt = nR[f] - mr;
  if ((int)blockIdx.x * KM_QPB >= nq) return;   // the whole workgroup
  knn2_mfma_scan(descL + ((size_t)f * capL + ml) * 8, nq, (int)blockIdx.x, descR + ((size_t)f * capR + mr) * 8,
                 max(nt, 0), 0u, out_idx + (size_t)f * capL * 2, out_dist + (size_t)f * capL * 2, sA);
}

// ---------------------------------------------------------------------------
// The same scan on the block-scaled FP4 form of the matrix instruction
// (v_mfma_scale_f32_32x32x64_f8f6f4 with both operands e2m1 and both scales 2^0): 64 bits of the
// descriptors per instruction instead of 32, in about the same cycles, so twice the pairs per
// clock of the int8 form.  A train bit is the nibble 0x2 (1.0) or 0x0, a query bit the nibble
// 0x2 (+1.0, bit clear) or 0xA (-1.0, bit set); products and sums are small integers, exact in
// the f32 accumulators.  Step s of a pair's four instructions takes words 2s and 2s+1 of the
// descriptors, lane half h word 2s+h, whose 32 bits become the 32 nibbles of the lane's four
// operand dwords: dword d holds bits d, d+4, .. d+28 (the slot order inside an instruction is
// free as long as both operands use the same one), which is (word << 1 >> d) & 0x22222222 - one
// shift and one and per eight bits.  A train tile of 32 rows is 4 KB in LDS (step, half, row:
// a wave's 16-byte reads are contiguous); everything else - query tiles resident in registers,
// two train tiles per barrier, per-lane top-2 upkeep gated by one compare per tile - is the
// int8 kernel's.
// ---------------------------------------------------------------------------
#ifndef KF_NQT
#define KF_NQT 2                              // query tiles of 32 per wave
#endif
#ifndef KF_WAVES
#define KF_WAVES 4                            // waves per workgroup
#endif
#define KF_QPB (KF_WAVES * KF_NQT * 32)       // queries per workgroup
#ifndef KF_SUB
#define KF_SUB 2                              // train tiles of 32 rows per stage
#endif
#define KF_TILE 4096                          // bytes of one train tile in LDS
typedef int km_v8i __attribute__((ext_vector_type(8)));
typedef float km_v16f __attribute__((ext_vector_type(16)));
#define KF_SCALE_ONE 0x7f7f7f7f               // E8M0 scale bytes: 2^(127 - 127)

// the 32 bits of a word as 32 e2m1 nibbles, 1.0 where the bit is set
__device__ __forceinline__ km_v4i bits_to_fp4(uint32_t w) {
  km_v4i r;
  r.x = (int)((w << 1) & 0x22222222u);
  r.y = (int)(w & 0x22222222u);
  r.z = (int)((w >> 1) & 0x22222222u);
  r.w = (int)((w >> 2) & 0x22222222u);
  return r;
}
// 0 / 1.0 nibbles -> +1.0 / -1.0 nibbles
__device__ __forceinline__ int fp4_pm1(int t) { return (int)(0x22222222u | ((uint32_t)t << 2)); }
__device__ __forceinline__ km_v8i fp4_operand(km_v4i v) {   // the instruction reads the first four dwords
  km_v8i r = {v.x, v.y, v.z, v.w, 0, 0, 0, 0};
  return r;
}

__device__ __forceinline__ void knn2_fp4_scan(const uint32_t* __restrict__ q, int nq, int qblock,
                                              const uint32_t* __restrict__ tp, int len, uint32_t idx_base,
                                              uint32_t* __restrict__ out_idx, uint16_t* __restrict__ out_dist,
                                              uint8_t (*sA)[KF_SUB * KF_TILE]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // ---- the wave's queries as B operands (+1 / -1), and their popcounts --------
  km_v4i bq[KF_NQT][4];
  float qpop[KF_NQT];    // popcount: distance = result + popcount
#pragma unroll
  for (int qt = 0; qt < KF_NQT; ++qt) {
    const int qi = qblock * KF_QPB + (wave * KF_NQT + qt) * 32 + r;
    const uint4* qp = (const uint4*)(q + (size_t)min(qi, nq - 1) * 8);
    const uint4 a = qp[0], b = qp[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    int pc = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) pc += __builtin_popcount(w[s]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // (the half's word by mask: a select between two elements of w[] by a run-time condition
      // becomes an indexed read of the array in scratch memory)
      const uint32_t hm = 0u - (uint32_t)h;
      const km_v4i e = bits_to_fp4((w[2 * s + 1] & hm) | (w[2 * s] & ~hm));
      bq[qt][s].x = fp4_pm1(e.x); bq[qt][s].y = fp4_pm1(e.y); bq[qt][s].z = fp4_pm1(e.z); bq[qt][s].w = fp4_pm1(e.w);
    }
    qpop[qt] = (float)pc;
  }
  uint32_t k0[KF_NQT], k1[KF_NQT];
  // The accumulators start at zero (an inline operand of the instruction: no registers hold a start
  // vector), so a result is the product itself, a small integer in [-256, 256] held in a float; the
  // smallest of a lane's 16 results comes out of float min3 instructions.  A row can enter when its
  // result is below thd = second-best distance - popcount.
  float thd[KF_NQT];
#pragma unroll
  for (int qt = 0; qt < KF_NQT; ++qt) { k0[qt] = 0xFFFFFFFFu; k1[qt] = 0xFFFFFFFFu; thd[qt] = 1024.f; }
  km_v16f zero;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero[i] = 0.0f;

  // ---- train tiles, KF_SUB of them per stage (one workgroup barrier per stage) ------
  // staging: this thread's row of a tile and word of the row; word sw = 2 s + h goes to the
  // operand slot of step s, half h
  // (a stage is KF_SUB * 256 words; thread tid takes words tid, tid + threads, ...)
  constexpr int kThreads = 64 * KF_WAVES;
  constexpr int kPer = KF_SUB * 256 / kThreads;          // words per thread and stage
  static_assert(KF_SUB * 256 % kThreads == 0, "stage words must divide over the workgroup");
  const int ntiles = (len + 31) >> 5;
  const int nstages = (ntiles + KF_SUB - 1) / KF_SUB;
  uint32_t pre[kPer];
  auto fetch = [&](int stg) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int wi = tid + u * kThreads;                  // word of the stage: tile, word of the row, row
      const int row = (stg * KF_SUB + (wi >> 8)) * 32 + (wi & 31);
      pre[u] = row < len ? tp[(size_t)row * 8 + ((wi >> 5) & 7)] : 0u;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int wi = tid + u * kThreads;
      *(km_v4i*)&sA[buf][(wi >> 8) * KF_TILE + (((wi >> 5) & 7) * 32 + (wi & 31)) * 16] = bits_to_fp4(pre[u]);
    }
  };
  if (nstages > 0) { fetch(0); stage(0); }
  __syncthreads();
  for (int stg = 0; stg < nstages; ++stg) {
    const int cur = stg & 1;
    if (stg + 1 < nstages) fetch(stg + 1);
#pragma unroll
    for (int sub = 0; sub < KF_SUB; ++sub) {
      const int tile = stg * KF_SUB + sub;
      const uint8_t* tileA = &sA[cur][sub * KF_TILE];
      km_v16f acc[KF_NQT];
      km_v4i a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *(const km_v4i*)&tileA[((u * 2 + h) * 32 + r) * 16];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int qt = 0; qt < KF_NQT; ++qt)
          acc[qt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_operand(a[u]), fp4_operand(bq[qt][u]),
                                                                    u == 0 ? zero : acc[qt], 4, 4, 0, KF_SCALE_ONE, 0,
                                                                    KF_SCALE_ONE);
      // can any of the lane's 16 rows enter?  (smallest product against the threshold; rows past
      // the segment's end are zeros in LDS and are refused again where the keys are built)
      bool any[KF_NQT];
      bool anyAll = false;
#pragma unroll
      for (int qt = 0; qt < KF_NQT; ++qt) {
        float mn = acc[qt][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mn = fminf(mn, acc[qt][i]);
        any[qt] = mn < thd[qt];
        anyAll |= any[qt];
      }
      if (__any(anyAll)) {
        // (an empty asm statement cannot be executed speculatively: without it the compiler could
        // turn this rarely taken block into straight-line code with selects)
        asm volatile("" ::: "memory");
        const int rowBase = tile * 32 + 4 * h;
        const bool whole = tile * 32 + 32 <= len;     // every row of the tile exists (all but the last tile)
#pragma unroll
        for (int qt = 0; qt < KF_NQT; ++qt) {
          if (!__any(any[qt])) continue;               // wave-uniform: only the query tile that has a candidate
          asm volatile("" ::: "memory");
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = rowBase + (i & 3) + 8 * (i >> 2);
            const uint32_t dist = (uint32_t)(int)(qpop[qt] + acc[qt][i]);
            const uint32_t key = (whole || row < len) ? ((dist << KNN_IDX_BITS) | (uint32_t)row) : 0xFFFFFFFFu;
            const uint32_t hi = max(k0[qt], key);
            k0[qt] = min(k0[qt], key);
            k1[qt] = min(k1[qt], hi);
          }
          // (an empty second slot gives a distance field of 1023: above every result)
          thd[qt] = (float)(int)(k1[qt] >> KNN_IDX_BITS) - qpop[qt];
        }
      }
    }
    if (stg + 1 < nstages) stage(cur ^ 1);
    __syncthreads();
  }
  // ---- the two lanes of a query merge their pairs ------------------------------
#pragma unroll
  for (int qt = 0; qt < KF_NQT; ++qt) {
    const uint32_t p0 = __shfl_xor(k0[qt], 32, 64), p1 = __shfl_xor(k1[qt], 32, 64);
    const uint32_t m0 = min(k0[qt], p0);
    const uint32_t m1 = min(max(k0[qt], p0), min(k1[qt], p1));
    const int qi = qblock * KF_QPB + (wave * KF_NQT + qt) * 32 + r;
    if (h == 0 && qi < nq) {
      const size_t o = (size_t)qi * 2;
      const uint32_t mask = (1u << KNN_IDX_BITS) - 1u;
      out_idx[o] = m0 == 0xFFFFFFFFu ? 0xFFFFFFFFu : idx_base + (m0 & mask);
      out_idx[o + 1] = m1 == 0xFFFFFFFFu ? 0xFFFFFFFFu : idx_base + (m1 & mask);
      out_dist[o] = m0 == 0xFFFFFFFFu ? (uint16_t)0xFFFF : (uint16_t)(m0 >> KNN_IDX_BITS);
      out_dist[o + 1] = m1 == 0xFFFFFFFFu ? (uint16_t)0xFFFF : (uint16_t)(m1 >> KNN_IDX_BITS);
    }
  }
}

__global__ __launch_bounds__(64 * KF_WAVES) void k_knn2_fp4(const uint32_t* __restrict__ q, int nq,
                                                  const uint32_t* __restrict__ t, long long nt,
                                                  int seg_len, uint32_t index_offset,
                                                  uint32_t* __restrict__ out_idx,
                                                  uint16_t* __restrict__ out_dist) {
  __shared__ __align__(16) uint8_t sA[2][KF_SUB * KF_TILE];
  const int seg = blockIdx.y;
  const long long t0 = (long long)seg * seg_len;
  const int len = (int)min((long long)seg_len, nt - t0);
  knn2_fp4_scan(q, nq, (int)blockIdx.x, t + (size_t)t0 * 8, len, index_offset + (uint32_t)t0,
                out_idx + (size_t)seg * nq * 2, out_dist + (size_t)seg * nq * 2, sA);
}

__global__ __launch_bounds__(64 * KF_WAVES) void k_knn2_fp4_frames(const uint32_t* __restrict__ descL, int capL,
                                                         const int* __restrict__ nL, const int* __restrict__ monoL,
                                                         const uint32_t* __restrict__ descR, int capR,
                                                         const int* __restrict__ nR, const int* __restrict__ monoR,
                                                         uint32_t* __restrict__ out_idx,
                                                         uint16_t* __restrict__ out_dist) {
  __shared__ __align__(16) uint8_t sA[2][KF_SUB * KF_TILE];
  const int f = blockIdx.y;
  const int ml = monoL[f], mr = monoR[f];
  const int nq = nL[f] - ml, nt = nR[f] - mr;
  if ((int)blockIdx.x * KF_QPB >= nq) return;   // the whole workgroup
  knn2_fp4_scan(descL + ((size_t)f * capL + ml) * 8, nq, (int)blockIdx.x, descR + ((size_t)f * capR + mr) * 8,
                max(nt, 0), 0u, out_idx + (size_t)f * capL * 2, out_dist + (size_t)f * capL * 2, sA);
}

// Merge partial 2-NN lists by (distance, index); part p holds its nq*2 indices at idx_parts +
// p*idx_stride bytes and its nq*2 distances at dist_parts + p*dist_stride bytes.  Parts are
// visited in order and comparisons are strict, so equal (distance, index) pairs cannot reorder
// and the earliest index wins a distance tie.
__global__ __launch_bounds__(256) void k_top2_merge(const uint8_t* __restrict__ idx_parts, size_t idx_stride,
                                                    const uint8_t* __restrict__ dist_parts, size_t dist_stride,
                                                    int nparts, int nq, uint32_t* __restrict__ idx,
                                                    uint16_t* __restrict__ dist) {
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (qi >= nq) return;
  unsigned long long b0 = ~0ull, b1 = ~0ull;
  for (int p = 0; p < nparts; ++p) {
    const uint2 pi = ((const uint2*)(idx_parts + (size_t)p * idx_stride))[qi];
    const uint32_t pd = ((const uint32_t*)(dist_parts + (size_t)p * dist_stride))[qi];
    const unsigned long long key[2] = {((unsigned long long)(pd & 0xFFFFu) << 32) | pi.x,
                                       ((unsigned long long)(pd >> 16) << 32) | pi.y};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned long long hi = key[e] > b0 ? key[e] : b0;
      b0 = key[e] < b0 ? key[e] : b0;
      b1 = hi < b1 ? hi : b1;
    }
  }
  idx[2 * (size_t)qi] = (uint32_t)b0;
  idx[2 * (size_t)qi + 1] = (uint32_t)b1;
  dist[2 * (size_t)qi] = (uint16_t)(b0 >> 32);
  dist[2 * (size_t)qi + 1] = (uint16_t)(b1 >> 32);
}

// Lowe's ratio test as the caller of knnMatch applies it (core/src/Frame.cc:1185):
//   matches.size() >= 2 && matches[0].distance < matches[1].distance * 0.7
// DMatch::distance is a float holding the integer distance and the product is taken in double
// (float * double literal).  match[i] = the nearest train index when the test holds, else -1.
// grid (query blocks, frames): frame f's records start at f * cap * 2 and its nq[f] - mono[f]
// first queries are tested (nq / mono null: count queries of one frame).
__global__ __launch_bounds__(256) void k_ratio_test(const uint32_t* __restrict__ idx, const uint16_t* __restrict__ dist,
                                                    int cap, const int* __restrict__ nq, const int* __restrict__ mono,
                                                    int count, double ratio, int32_t* __restrict__ match) {
  const int f = blockIdx.y;
  const int n = nq ? nq[f] - mono[f] : count;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || i >= cap) return;
  const size_t o = (size_t)f * cap + i;
  const uint2 id = ((const uint2*)idx)[o];
  const uint32_t d = ((const uint32_t*)dist)[o];
  const bool two = id.y != 0xFFFFFFFFu;
  const bool good = two && (double)(float)(d & 0xFFFFu) < (double)(float)(d >> 16) * ratio;
  match[o] = good ? (int32_t)id.x : -1;
}

// One lane per left keypoint; right keypoints are visited in ascending index,
// so the first minimum wins as in the reference's row-bucket order.
struct KP { float x, y, size, angle, response; int octave, class_id; };

__global__ __launch_bounds__(256) void k_stereo_band(const KP* __restrict__ kl,
                                                     const uint32_t* __restrict__ dl, int nl,
                                                     const KP* __restrict__ kr,
                                                     const uint32_t* __restrict__ dr, int nr,
                                                     const float* __restrict__ scaleFactors,
                                                     int nRows, float minD, float maxD,
                                                     int* __restrict__ bestIdx,
                                                     int* __restrict__ bestDist) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nl) return;
  const KP L = kl[i];
  uint32_t a[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) a[w] = dl[(size_t)i * 8 + w];
  const int row = (int)L.y;
  const float minU = __fsub_rn(L.x, maxD), maxU = __fsub_rn(L.x, minD);
  int best = 100, bi = -1;
  const bool live = row >= 0 && row < nRows && !(maxU < 0.f);
  if (live) {
    for (int j = 0; j < nr; ++j) {
      const KP R = kr[j];
      const float r = __fmul_rn(2.0f, scaleFactors[R.octave]);
      const int maxr = (int)ceilf(__fadd_rn(R.y, r));
      const int minr = (int)floorf(__fsub_rn(R.y, r));
      if (row < minr || row > maxr) continue;
      if (R.octave < L.octave - 1 || R.octave > L.octave + 1) continue;
      if (!(R.x >= minU && R.x <= maxU)) continue;
      int d = 0;
#pragma unroll
      for (int w = 0; w < 8; ++w) d += __builtin_popcount(a[w] ^ dr[(size_t)j * 8 + w]);
      if (d < best) { best = d; bi = j; }
    }
  }
  bestIdx[i] = bi;
  bestDist[i] = best;
}

}  // namespace orbx

namespace orbx {
void launch_knn2_frames(const uint8_t* descL, int capL, const int* nL, const int* monoL, const uint8_t* descR,
                        int capR, const int* nR, const int* monoR, uint32_t* idx, uint16_t* dist, int nframes,
                        hipStream_t s) {
  const char* e = std::getenv("ORBX_KNN2_I8");
  const bool i8 = e && e[0] == '1';
  auto kern = i8 ? k_knn2_mfma_frames : k_knn2_fp4_frames;
  const int qpb = i8 ? KM_QPB : KF_QPB;
  hipLaunchKernelGGL(kern, dim3((capL + qpb - 1) / qpb, nframes), dim3(i8 ? 256 : 64 * KF_WAVES), 0, s,
                     (const uint32_t*)descL, capL, nL, monoL, (const uint32_t*)descR, capR, nR, monoR, idx, dist);
}
void launch_ratio_test_frames(const uint32_t* idx, const uint16_t* dist, int cap, const int* nq, const int* mono,
                              int nframes, double ratio, int32_t* match, hipStream_t s) {
  hipLaunchKernelGGL(k_ratio_test, dim3((cap + 255) / 256, nframes), dim3(256), 0, s, idx, dist, cap, nq, mono, 0,
                     ratio, match);
}
void launch_top2_merge_strided(const uint8_t* idx_parts, size_t idx_stride, const uint8_t* dist_parts,
                               size_t dist_stride, int nparts, int nq, uint32_t* idx, uint16_t* dist,
                               hipStream_t s) {
  hipLaunchKernelGGL(k_top2_merge, dim3((nq + 255) / 256), dim3(256), 0, s, idx_parts, idx_stride, dist_parts,
                     dist_stride, nparts, nq, idx, dist);
}
}  // namespace orbx

namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};
}  // namespace

extern "C" {

int orbx_hamming(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 8; ++i) {
    uint32_t x, y;
    __builtin_memcpy(&x, a + 4 * i, 4);
    __builtin_memcpy(&y, b + 4 * i, 4);
    d += __builtin_popcount(x ^ y);
  }
  return d;
}

// How a scan of nt train rows for nq queries is cut into segments (grid y): enough segments to
// occupy the chip when there are few query blocks, none longer than the key's index field, and
// - when the partial results must fit a fixed buffer - no more than that buffer holds.
struct KnnCut { int qblocks; long long segs; int seg_len; };
static const size_t kPartialBytesPerQuery = 12;   // two uint32 indices + two uint16 distances

// ORBX_KNN2_VALU=1 (timing knob, same results): the vector-ALU scan instead of the matrix-core one
// (read at every call, so that a test can compare the forms inside one process)
static bool knn2_use_valu() {
  const char* e = std::getenv("ORBX_KNN2_VALU");
  return e && e[0] == '1';
}
// ORBX_KNN2_I8=1 (timing knob, same results): the int8 form of the matrix-core scan instead of the FP4 one
static bool knn2_use_i8() {
  const char* e = std::getenv("ORBX_KNN2_I8");
  return e && e[0] == '1';
}

static int knn2_cut(int nq, int64_t nt, size_t partial_budget, KnnCut* c) {
  const int qpb = knn2_use_valu() ? KNN_QPL * KNN_BLOCK : (knn2_use_i8() ? KM_QPB : KF_QPB);   // queries per workgroup
  c->qblocks = (nq + qpb - 1) / qpb;
  long long segs = 1, minSegs = 1;
  if (nt > 0) {
    // enough workgroups for some twenty rounds of the chip, so that the last round's idle CUs cost little
    const long long want = ((knn2_use_valu() ? 2048 : KNN_WANT_WGS) + c->qblocks - 1) / c->qblocks;
    const long long minLen = 512;
    segs = std::max(1LL, std::min(want, (long long)((nt + minLen - 1) / minLen)));
    const long long maxLen = 1LL << KNN_IDX_BITS;
    minSegs = (long long)((nt + maxLen - 1) / maxLen);
    segs = std::max(segs, minSegs);
  }
  if (partial_budget != (size_t)-1 && segs > 1) {
    const long long fit = (long long)(partial_budget / ((size_t)nq * kPartialBytesPerQuery + 512));
    if (fit < minSegs || (fit < 2 && minSegs > 1)) return ORBX_ERR_SCRATCH;
    segs = std::max(minSegs, std::min(segs, std::max(fit, 1LL)));
  }
  c->seg_len = nt > 0 ? (int)((nt + segs - 1) / segs) : 1;
  c->segs = nt > 0 ? (nt + c->seg_len - 1) / c->seg_len : 1;
  if (c->segs > 65535) return ORBX_ERR_UNSUPPORTED;
  return ORBX_OK;
}

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// The scan and, for more than one segment, the merge of the segments' partial results, which
// live in `partial` (align256(segs*nq*8) bytes of indices, then segs*nq*4 bytes of distances).
static int knn2_enqueue(const KnnCut& c, const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt,
                        uint32_t index_offset, uint32_t* d_idx, uint16_t* d_dist, uint8_t* partial,
                        hipStream_t s) {
  auto kern = knn2_use_valu() ? orbx::k_knn2 : (knn2_use_i8() ? orbx::k_knn2_mfma : orbx::k_knn2_fp4);
  const int threads = (knn2_use_valu() || knn2_use_i8()) ? 256 : 64 * KF_WAVES;
  if (c.segs == 1) {
    hipLaunchKernelGGL(kern, dim3(c.qblocks, 1), dim3(threads), 0, s, (const uint32_t*)d_q, nq,
                       (const uint32_t*)d_t, (long long)nt, c.seg_len, index_offset, d_idx, d_dist);
  } else {
    uint8_t* pidx = partial;
    uint8_t* pdist = partial + align256((size_t)c.segs * nq * 8);
    hipLaunchKernelGGL(kern, dim3(c.qblocks, (unsigned)c.segs), dim3(threads), 0, s,
                       (const uint32_t*)d_q, nq, (const uint32_t*)d_t, (long long)nt, c.seg_len,
                       index_offset, (uint32_t*)pidx, (uint16_t*)pdist);
    hipLaunchKernelGGL(orbx::k_top2_merge, dim3((nq + 255) / 256), dim3(256), 0, s, pidx, (size_t)nq * 8,
                       pdist, (size_t)nq * 4, (int)c.segs, nq, d_idx, d_dist);
  }
  return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_HIP;
}

static int knn2_launch(const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt,
                       uint32_t index_offset, uint32_t* d_idx, uint16_t* d_dist, hipStream_t s) {
  if (nq <= 0) return ORBX_OK;
  KnnCut c;
  int st = knn2_cut(nq, nt, (size_t)-1, &c);
  if (st != ORBX_OK) return st;
  uint8_t* partial = nullptr;
  if (c.segs > 1) {
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess) orbx::keep_pool_memory(dev);   // once per device
  }
  if (c.segs > 1 &&
      hipMallocAsync((void**)&partial, align256((size_t)c.segs * nq * 8) + (size_t)c.segs * nq * 4, s) != hipSuccess)
    return ORBX_ERR_HIP;
  st = knn2_enqueue(c, d_q, nq, d_t, nt, index_offset, d_idx, d_dist, partial, s);
  if (partial) (void)hipFreeAsync(partial, s);
  return st;
}

static bool knn2_args_ok(const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt, const uint32_t* d_idx,
                         const uint16_t* d_dist) {
  if (nq < 0 || nt < 0 || (nq > 0 && (!d_q || !d_idx || !d_dist)) || (nt > 0 && !d_t)) return false;
  if (((uintptr_t)d_q | (uintptr_t)d_t) & 15) return false;   // rows are read as 16-byte words
  return true;
}

int orbx_knn2_device(const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt, uint32_t index_offset,
                     uint32_t* d_idx, uint16_t* d_dist, void* stream) {
  if (!knn2_args_ok(d_q, nq, d_t, nt, d_idx, d_dist)) return ORBX_ERR_BAD_ARG;
  return knn2_launch(d_q, nq, d_t, nt, index_offset, d_idx, d_dist, (hipStream_t)stream);
}

// ---- the allocation-free form of orbx_knn2_device --------------------------------------------
// Per-segment partial results live in a file-scope device array instead of stream-ordered
// allocations.  Every GPU the library's code object is loaded on holds its own copy of the
// array at its own address: host code never passes the array's name to a kernel, only the
// address hipGetSymbolAddress returned for the device that is current at the call.
#define ORBX_KNN2_SCRATCH_BYTES ((size_t)64 << 20)
static __device__ __align__(256) uint8_t scratch[ORBX_KNN2_SCRATCH_BYTES];

namespace {
const int kMaxDevices = 64;
std::mutex g_scratch_mu;
uint8_t* g_scratch_ptr[kMaxDevices] = {};

// the current device's copy of `scratch`, looked up once per device
int scratch_of_current_device(uint8_t** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return ORBX_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return ORBX_ERR_HIP;
    g_scratch_ptr[dev] = (uint8_t*)p;
  }
  *out = g_scratch_ptr[dev];
  return ORBX_OK;
}
}  // namespace

int orbx_knn2_device_noalloc(const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt, uint32_t index_offset,
                             uint32_t* d_idx, uint16_t* d_dist, void* stream) {
  if (!knn2_args_ok(d_q, nq, d_t, nt, d_idx, d_dist)) return ORBX_ERR_BAD_ARG;
  if (nq == 0) return ORBX_OK;
  KnnCut c;
  int st = knn2_cut(nq, nt, ORBX_KNN2_SCRATCH_BYTES, &c);
  if (st != ORBX_OK) return st;
  uint8_t* partial = nullptr;
  if (c.segs > 1) {
    st = scratch_of_current_device(&partial);
    if (st != ORBX_OK) return st;
  }
  // every (segment, query) slot the merge reads is written by the scan of the same call, so
  // nothing left in the array by an earlier call is ever read and nothing needs clearing
  return knn2_enqueue(c, d_q, nq, d_t, nt, index_offset, d_idx, d_dist, partial, (hipStream_t)stream);
}

size_t orbx_knn2_noalloc_scratch_bytes(void) { return ORBX_KNN2_SCRATCH_BYTES; }

int orbx_top2_merge_device(const uint32_t* d_idx_parts, const uint16_t* d_dist_parts, int nparts, int nq,
                           uint32_t* d_idx, uint16_t* d_dist, void* stream) {
  if (nparts < 1 || nq < 0 || (nq > 0 && (!d_idx_parts || !d_dist_parts || !d_idx || !d_dist)))
    return ORBX_ERR_BAD_ARG;
  if (((uintptr_t)d_idx_parts & 7) || ((uintptr_t)d_dist_parts & 3)) return ORBX_ERR_BAD_ARG;
  if (nq == 0) return ORBX_OK;
  hipLaunchKernelGGL(orbx::k_top2_merge, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)d_idx_parts, (size_t)nq * 8, (const uint8_t*)d_dist_parts, (size_t)nq * 4,
                     nparts, nq, d_idx, d_dist);
  return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_HIP;
}

int orbx_ratio_test_device(const uint32_t* d_idx, const uint16_t* d_dist, int nq, double ratio, int32_t* d_match,
                           void* stream) {
  if (nq < 0 || !(ratio >= 0.0) || (nq > 0 && (!d_idx || !d_dist || !d_match))) return ORBX_ERR_BAD_ARG;
  if (((uintptr_t)d_idx & 7) || ((uintptr_t)d_dist & 3)) return ORBX_ERR_BAD_ARG;
  if (nq == 0) return ORBX_OK;
  hipLaunchKernelGGL(orbx::k_ratio_test, dim3((nq + 255) / 256, 1), dim3(256), 0, (hipStream_t)stream, d_idx, d_dist,
                     nq, (const int*)nullptr, (const int*)nullptr, nq, ratio, d_match);
  return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_HIP;
}

int orbx_knn2(int device_id, const uint8_t* q, int nq, const uint8_t* t, int64_t nt, uint32_t* idx,
              uint16_t* dist) {
  if (nq < 0 || nt < 0 || (nq > 0 && (!q || !idx || !dist)) || (nt > 0 && !t)) return ORBX_ERR_BAD_ARG;
  if (nq == 0) return ORBX_OK;
  if (hipSetDevice(device_id) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  DevBuf dq, dt, di, dd;
  if (dq.alloc((size_t)nq * 32) != hipSuccess || dt.alloc((size_t)nt * 32) != hipSuccess ||
      di.alloc((size_t)nq * 8) != hipSuccess || dd.alloc((size_t)nq * 4) != hipSuccess)
    return ORBX_ERR_HIP;
  if (hipMemcpy(dq.p, q, (size_t)nq * 32, hipMemcpyHostToDevice) != hipSuccess) return ORBX_ERR_HIP;
  if (nt > 0 && hipMemcpy(dt.p, t, (size_t)nt * 32, hipMemcpyHostToDevice) != hipSuccess)
    return ORBX_ERR_HIP;
  int st = knn2_launch((const uint8_t*)dq.p, nq, (const uint8_t*)dt.p, nt, 0, (uint32_t*)di.p,
                       (uint16_t*)dd.p, nullptr);
  if (st != ORBX_OK) return st;
  if (hipDeviceSynchronize() != hipSuccess) return ORBX_ERR_HIP;
  if (hipMemcpy(idx, di.p, (size_t)nq * 8, hipMemcpyDeviceToHost) != hipSuccess) return ORBX_ERR_HIP;
  if (hipMemcpy(dist, dd.p, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess) return ORBX_ERR_HIP;
  return ORBX_OK;
}

int orbx_stereo_band_match(int device_id, const orbx_keypoint* kps_l, const uint8_t* desc_l, int n_l,
                           const orbx_keypoint* kps_r, const uint8_t* desc_r, int n_r,
                           const float* scale_factors, int nlevels, int n_rows, float min_d,
                           float max_d, int32_t* best_idx, int32_t* best_dist) {
  if (n_l < 0 || n_r < 0 || nlevels < 1 || !scale_factors || (n_l > 0 && (!kps_l || !desc_l || !best_idx || !best_dist)) ||
      (n_r > 0 && (!kps_r || !desc_r)))
    return ORBX_ERR_BAD_ARG;
  if (n_l == 0) return ORBX_OK;
  for (int j = 0; j < n_r; ++j)
    if (kps_r[j].octave < 0 || kps_r[j].octave >= nlevels) return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(device_id) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  DevBuf kl, dl, kr, dr, sf, bi, bd;
  if (kl.alloc((size_t)n_l * 28) != hipSuccess || dl.alloc((size_t)n_l * 32) != hipSuccess ||
      kr.alloc((size_t)n_r * 28) != hipSuccess || dr.alloc((size_t)n_r * 32) != hipSuccess ||
      sf.alloc((size_t)nlevels * 4) != hipSuccess || bi.alloc((size_t)n_l * 4) != hipSuccess ||
      bd.alloc((size_t)n_l * 4) != hipSuccess)
    return ORBX_ERR_HIP;
  hipError_t e = hipMemcpy(kl.p, kps_l, (size_t)n_l * 28, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dl.p, desc_l, (size_t)n_l * 32, hipMemcpyHostToDevice);
  if (e == hipSuccess && n_r) e = hipMemcpy(kr.p, kps_r, (size_t)n_r * 28, hipMemcpyHostToDevice);
  if (e == hipSuccess && n_r) e = hipMemcpy(dr.p, desc_r, (size_t)n_r * 32, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(sf.p, scale_factors, (size_t)nlevels * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) return ORBX_ERR_HIP;
  hipLaunchKernelGGL(orbx::k_stereo_band, dim3((n_l + 255) / 256), dim3(256), 0, nullptr,
                     (const orbx::KP*)kl.p, (const uint32_t*)dl.p, n_l, (const orbx::KP*)kr.p,
                     (const uint32_t*)dr.p, n_r, (const float*)sf.p, n_rows, min_d, max_d, (int*)bi.p,
                     (int*)bd.p);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(best_idx, bi.p, (size_t)n_l * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(best_dist, bd.p, (size_t)n_l * 4, hipMemcpyDeviceToHost);
  return e == hipSuccess ? ORBX_OK : ORBX_ERR_HIP;
}

}  // extern "C"
