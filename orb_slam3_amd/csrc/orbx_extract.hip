// gfx950 kernels of the ORB extractor and their launch functions.
//
// Stage -> reference code it reproduces (paths relative to the ORB-SLAM3 tree):
//   k_pyr_level0 / k_pyr_resize   ORBextractor::ComputePyramid      core/lib/ORB/extractor.cc:1103-1127
//   k_fast_cells                  grid + FAST part of ComputeKeyPointsOctTree   :747-831
//   k_octree                      DistributeOctTree + DivideNode    :474-745
//   k_assemble                    key fix-up + two-ended output fill :840-849, 1052-1100
//   k_blur                        GaussianBlur 7x7 sigma 2          :1063-1064
//   k_orient_describe             IC_Angle + computeOrbDescriptor   :75-145
//
// All arithmetic that decides an output bit is integer, or float with every
// operation individually rounded (this file is compiled with
// -ffp-contract=off), so results do not depend on scheduling.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_plan.h"
#include "stdsort_emul.h"

namespace orbx {

// ---------------------------------------------------------------------------
// small wave-level helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// Synchronisation inside a one-wave workgroup.  The LDS and vector-memory operations of a wave
// are performed in program order, so its lanes see each other's writes without a hardware
// barrier and without waiting for outstanding loads or stores; only the compiler must keep the
// order.  (__syncthreads() would also drain every outstanding global load and store.)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// number of set bits of m below the calling lane
__device__ __forceinline__ int mbcnt64(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Inclusive prefix sum over the 64 lanes through the data-parallel-primitive forms of v_add: four
// shifts inside every row of 16 lanes, then the last lane of row 0 / 2 into rows 1 / 3 and lane 31
// into rows 2 and 3 - six instructions and no LDS round trip (a shuffle is an LDS permute plus a
// select per step).  All 64 lanes must be active.
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

// the sum over the 64 lanes, in every lane
__device__ __forceinline__ int wave_sum(int v) {
  return __builtin_amdgcn_readlane(wave_incl_scan(v), 63);
}

// a.lo*b.lo + a.hi*b.hi + c on unsigned 16-bit halves (v_dot2_u32_u16)
typedef unsigned short dot_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(dot_us2, a), __builtin_bit_cast(dot_us2, b), c, false);
}

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// ---------------------------------------------------------------------------
// K1a: level 0 = input image + 19-pixel REFLECT_101 border.
// A lane produces 16 padded pixels (one 16-byte store) of 2 rows.  Interior
// groups read 20 source bytes per row with dword-aligned wide loads and
// re-base them with v_alignbyte; groups that touch the border go per byte.
// ---------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) U4A4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) U3A4 { uint32_t x, y, z; };
typedef uint32_t U3V4 __attribute__((ext_vector_type(3), aligned(4)));   // the same 12 bytes as a vector (loadable through any address space)
typedef uint32_t U4V4 __attribute__((ext_vector_type(4), aligned(4)));

// A wave-uniform plane address pinned to a pair of scalar registers, so that every access to the
// plane is "scalar base + 32-bit lane offset" (one 32-bit add per access).  Left to itself the
// optimiser folds the lane's offset into the base once and then redoes 64-bit row arithmetic per
// lane for every load and store.  (The pointer is rebuilt from integers, so its address space is
// stated.)
#define GLOBAL_AS __attribute__((address_space(1)))
__device__ __forceinline__ const GLOBAL_AS uint8_t* uniform_base(const uint8_t* base) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return (const GLOBAL_AS uint8_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

#define L0_ROWS 4  // rows per lane

// High 32 bits of the product of two values below 2^24 (v_mul_hi_u32_u24).
__device__ __forceinline__ uint32_t mul_hi_u24(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)) >> 32);
}

// Workgroups are handed to the 8 XCDs round-robin by linear id and every XCD has a private
// L2.  For a 1-D launch of 8 * chunk workgroups this returns a work unit such that each XCD
// walks one contiguous range of units, so neighbouring units (which share cache lines of the
// image planes) meet in the same L2.  Units >= the real count must be skipped by the caller.
__device__ __forceinline__ int xcd_contiguous_unit(int chunk) {
  const int b = blockIdx.x;
  return (b & 7) * chunk + (b >> 3);
}

// The same for a (gx, gy, gz) arrangement of workgroups launched as 8 * chunk: x runs fastest
// inside an XCD's range, so the workgroups that share source lines - the column blocks of a row
// block, and row blocks above each other - fetch them through one L2 (handed out round-robin,
// neighbouring blocks each fetched the shared lines into an L2 of their own: the level chain read
// 1.4 times its source bytes past the L2).  False for the surplus workgroups.
__device__ __forceinline__ bool xcd_contiguous_block(int chunk, int gx, int gy, int nblocks, int& bx, int& by,
                                                     int& bz) {
  // chunk 0 (timing knob): workgroups taken in launch order, i.e. neighbours on different XCDs
  const int u = chunk > 0 ? xcd_contiguous_unit(chunk) : (int)blockIdx.x;
  if (u >= nblocks) return false;
  const int gxy = gx * gy;
  bz = u / gxy;
  const int r = u - bz * gxy;
  by = r / gx;
  bx = r - by * gx;
  return true;
}

__global__ __launch_bounds__(256) void k_pyr_level0(const uint8_t* __restrict__ img, int W, int H,
                                                    int istride, size_t iframe,
                                                    uint8_t* __restrict__ dst, int pstride,
                                                    int prows, size_t dplane, int nbx, int f0,
                                                    int gx, int gy, int nblocks, int chunk) {
  int bx, by, bz;
  if (!xcd_contiguous_block(chunk, gx, gy, nblocks, bx, by, bz)) return;
  const int f = bz + f0;
  const uint8_t* fimg = img + (size_t)f * iframe;
  // 16-pixel groups [2, ghi] are interior: all 16 pixels and the slack bytes the aligned
  // loads may touch lie inside the source row
  const int ghi = (W - 1) / 16;
  if (bx < nbx) {
    const int g = bx * 64 + threadIdx.x;
    // (a wave's rows are the same for all its lanes: said explicitly, so that row arithmetic is
    // scalar and every access is a scalar base plus a 32-bit lane offset)
    const int py0 = (by * (int)blockDim.y + __builtin_amdgcn_readfirstlane((int)threadIdx.y)) * L0_ROWS;
    if (g < 2 || g > ghi) return;
    const uint32_t P = (uint32_t)g * 16u;
    const GLOBAL_AS uint8_t* sbase = uniform_base(fimg);
    GLOBAL_AS uint8_t* dbase = (GLOBAL_AS uint8_t*)uniform_base(dst + (size_t)f * dplane);
#pragma unroll
    for (int r = 0; r < L0_ROWS; ++r) {
      const int py = py0 + r;
      if (py >= prows) break;
      const int sy = reflect101(py - ORBX_EDGE, H);
      // the 16 pixels start at byte rowOff + P - EDGE of the frame; loads are dword-aligned (P is
      // a multiple of 16, so the misalignment is the same for all lanes of the row)
      const uint32_t rowOff = (uint32_t)sy * (uint32_t)istride - (uint32_t)ORBX_EDGE;
      const uint32_t sh = ((uint32_t)(uintptr_t)fimg + rowOff) & 3u;
      const uint32_t voff = rowOff - sh + P;
      const U4V4 v = *(const GLOBAL_AS U4V4*)(sbase + voff);
      const uint32_t v4 = *(const GLOBAL_AS uint32_t*)(sbase + voff + 16u);
      U4V4 out;
      out.x = __builtin_amdgcn_alignbyte(v.y, v.x, sh);
      out.y = __builtin_amdgcn_alignbyte(v.z, v.y, sh);
      out.z = __builtin_amdgcn_alignbyte(v.w, v.z, sh);
      out.w = __builtin_amdgcn_alignbyte(v4, v.w, sh);
      *(GLOBAL_AS U4V4*)(dbase + ((uint32_t)py * (uint32_t)pstride + P)) = out;
    }
  } else {
    // border block: one lane per (row, dword) of the left 8 and the right border dwords
    const int cr0 = 4 * (ghi + 1);
    const int nright = (W + 2 * ORBX_EDGE + 3) / 4 - cr0;
    const int nb = 8 + nright;
    const int rows = (int)blockDim.y * L0_ROWS;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const float rnb = 1.0f / (float)nb;
    for (int t = tid; t < rows * nb; t += 64 * (int)blockDim.y) {
      const int r = (int)(((float)t + 0.5f) * rnb);
      const int j = t - r * nb;
      const int py = by * rows + r;
      if (py >= prows) break;
      const int c = j < 8 ? j : cr0 + (j - 8);
      const uint8_t* srow = fimg + (size_t)reflect101(py - ORBX_EDGE, H) * istride;
      uint32_t out = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int px = 4 * c + k;
        const int sx = reflect101(min(px, W + 2 * ORBX_EDGE - 1) - ORBX_EDGE, W);
        const uint32_t v = srow[sx];
        out |= (px < W + 2 * ORBX_EDGE ? v : 0u) << (8 * k);
      }
      *(uint32_t*)(dst + (size_t)f * dplane + (size_t)py * pstride + 4 * c) = out;
    }
  }
}

// ---------------------------------------------------------------------------
// K1b: level l = cv::resize(level l-1, INTER_LINEAR) + REFLECT_101 border.
// The tap tables are indexed by padded destination coordinates, so border
// pixels are produced by the same code as interior ones.
//
// Workgroups are few and fat: an empty launch of this stage's former one-wave
// workgroups already cost half of the stage (the dispatcher hands out about two
// workgroups per nanosecond).  A main wave owns 64 destination dword columns
// (256 pixels) and 4 * GROUPS rows, walked 4 rows at a time with the next
// step's source rows already in flight; four waves form a workgroup.  Columns
// in the reflected border or at the clamped right edge go through the
// per-pixel tap table in separate border workgroups, so that no wave mixes the
// two code paths.
// ---------------------------------------------------------------------------
struct ResizeRows {
  uint32_t a[4][3], b[4][3];   // upper / lower source row of 4 destination rows
  uint32_t wy0[4], wy1[4];     // vertical weights << 12
};

// Source rows of 4 destination rows.  prev = the row tap before tys[0] (null for the first
// step): a destination row whose upper source row is the previous destination row's lower one
// takes that row's horizontal sums over, so its upper row is not loaded (wave-uniform test).
// base: the plane's first interior row at the interior's first column, the same for the whole
// wave; laneOff: the lane's byte offset in a row.  Kept apart so that a row's address is a scalar
// base plus a 32-bit lane offset (no 64-bit vector arithmetic per load).
__device__ __forceinline__ void resize_fetch(ResizeRows& R, const GLOBAL_AS uint8_t* base, uint32_t laneOff, int sstride,
                                             const Tap* tys, const Tap* prev) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // one 12-byte load per source row (dword-aligned)
    const Tap* before = r > 0 ? tys + r - 1 : prev;
    if (!(before && before->s1 == tys[r].s0)) {
      const U3V4 a = *(const GLOBAL_AS U3V4*)(base + (laneOff + (uint32_t)tys[r].s0 * (uint32_t)sstride));
      R.a[r][0] = a.x; R.a[r][1] = a.y; R.a[r][2] = a.z;
    }
    const U3V4 b = *(const GLOBAL_AS U3V4*)(base + (laneOff + (uint32_t)tys[r].s1 * (uint32_t)sstride));
    R.b[r][0] = b.x; R.b[r][1] = b.y; R.b[r][2] = b.z;
    // (b * (H >> 4)) >> 16 is the high word of the 24-bit product (b << 12) * (H & ~15)
    R.wy0[r] = (uint32_t)(int)tys[r].w0 << 12;
    R.wy1[r] = (uint32_t)(int)tys[r].w1 << 12;
  }
}

template <int GROUPS>
__global__ __launch_bounds__(256) void k_pyr_resize(const uint8_t* __restrict__ src, int sstride,
                                                   size_t splane, uint8_t* __restrict__ dst,
                                                   int pstride, int prows, size_t dplane, int pw,
                                                   const Tap* __restrict__ xtab,
                                                   const Tap* __restrict__ ytab,
                                                   const ColRec* __restrict__ colrec,
                                                   const uint16_t* __restrict__ bcols, int nbc, int nbx,
                                                   int f0, int gx, int gy, int nblocks, int chunk) {
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  int bx, by, bz;
  if (!xcd_contiguous_block(chunk, gx, gy, nblocks, bx, by, bz)) return;
  const int f = bz + f0;
  constexpr int rowsPerWave = 4 * GROUPS;
  const int rowsPerBlock = (int)blockDim.y * rowsPerWave;
  if (bx >= nbx) {
    // border block: one lane per (row, border dword); taps straight from the tables,
    // source bytes straight from the (cache-resident) source plane
    const uint8_t* roi = src + (size_t)f * splane + (size_t)ORBX_EDGE * sstride + ORBX_EDGE;
    const float rnb = 1.0f / (float)nbc;
    for (int t = wave * 64 + lane; t < rowsPerBlock * nbc; t += 64 * (int)blockDim.y) {
      const int r = (int)(((float)t + 0.5f) * rnb);
      const int py = by * rowsPerBlock + r;
      if (py >= prows) break;
      const int c = bcols[t - r * nbc];
      const Tap ty = ytab[py];
      const uint8_t* r0 = roi + (size_t)ty.s0 * sstride;
      const uint8_t* r1 = roi + (size_t)ty.s1 * sstride;
      uint32_t out = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int px = 4 * c + k;
        const Tap tx = xtab[min(px, pw - 1)];
        const uint32_t h0 = __umul24(r0[tx.s0], (uint32_t)(int)tx.w0) + __umul24(r0[tx.s1], (uint32_t)(int)tx.w1);
        const uint32_t h1 = __umul24(r1[tx.s0], (uint32_t)(int)tx.w0) + __umul24(r1[tx.s1], (uint32_t)(int)tx.w1);
        const uint32_t v = ((__umul24((uint32_t)(int)ty.w0, h0 >> 4) >> 16) +
                            (__umul24((uint32_t)(int)ty.w1, h1 >> 4) >> 16) + 2u) >> 2;
        out |= (px < pw ? (v & 0xFFu) : 0u) << (8 * k);
      }
      *(uint32_t*)(dst + (size_t)f * dplane + (size_t)py * pstride + 4 * c) = out;
    }
    return;
  }
  // main waves: one lane = one destination dword column (4 pixels).  The 8 source bytes that
  // hold all taps of the 4 pixels are fetched per source row as three dword-aligned words
  // (neighbouring lanes overlap, the vector cache serves most of it), re-based with
  // v_alignbyte, and each pixel's tap pair is picked with one v_perm_b32.
  const int c = bx * 64 + lane;
  if (c * 4 >= pstride) return;
  const ColRec cr = colrec[c];
  if (!(cr.deltas & 0x8000)) return;   // border column: done by the border waves
  const int sh = (ORBX_EDGE + (int)cr.sxa) & 3;   // planes and rows are 64-byte aligned
  const GLOBAL_AS uint8_t* base = uniform_base(src + (size_t)f * splane + (size_t)ORBX_EDGE * sstride);
  const uint32_t laneOff = (uint32_t)(ORBX_EDGE + (int)cr.sxa - sh);   // >= 0: a main column's taps lie inside the row
  uint32_t sel[4], wp[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t d = (cr.deltas >> (3 * k)) & 7u;
    sel[k] = 0x0c000c00u | d | ((d + 1u) << 16);
    wp[k] = (2048u - cr.w1[k]) | ((uint32_t)cr.w1[k] << 16);  // (left, right) weights
  }
  const uint32_t keep = cr.nvalid >= 4 ? 0xFFFFFFFFu : ((1u << (8 * cr.nvalid)) - 1u);
  GLOBAL_AS uint8_t* dframe = (GLOBAL_AS uint8_t*)uniform_base(dst + (size_t)f * dplane);
  const uint32_t dOff = 4u * (uint32_t)c;
  const int pyBase = (by * (int)blockDim.y + wave) * rowsPerWave;
  if (pyBase >= prows) return;
  // the row taps of all steps first: wave-uniform, read before any store so that they can
  // come through the scalar cache
  // (made scalar explicitly: with the taps in scalar registers a row's address is a scalar base
  // plus the lane's 32-bit offset, and the row weights are scalar operands)
  Tap tys[rowsPerWave];
#pragma unroll
  for (int r = 0; r < rowsPerWave; ++r) {
    const uint2 t = *(const uint2*)&ytab[min(pyBase + r, prows - 1)];
    const uint32_t ss = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.x), ww = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.y);
    tys[r].s0 = (uint16_t)(ss & 0xFFFFu); tys[r].s1 = (uint16_t)(ss >> 16);
    tys[r].w0 = (int16_t)(ww & 0xFFFFu); tys[r].w1 = (int16_t)(ww >> 16);
  }
  ResizeRows buf[2];
  resize_fetch(buf[0], base, laneOff, sstride, tys, nullptr);
  uint32_t hprev[4] = {0u, 0u, 0u, 0u};   // horizontal sums (low 4 bits cleared) of the previous row's lower source row
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    const int py0 = pyBase + 4 * g;
    if (py0 >= prows) break;
    if (g + 1 < GROUPS) resize_fetch(buf[(g + 1) & 1], base, laneOff, sstride, tys + 4 * (g + 1), tys + 4 * g + 3);
    const ResizeRows& cur = buf[g & 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int py = py0 + r;
      if (py < prows) {
        const int rg = 4 * g + r;
        const bool shared = rg > 0 && tys[rg - 1].s1 == tys[rg].s0;   // wave-uniform
        const uint32_t blo = __builtin_amdgcn_alignbyte(cur.b[r][1], cur.b[r][0], sh), bhi = __builtin_amdgcn_alignbyte(cur.b[r][2], cur.b[r][1], sh);
        uint32_t h0[4], h1[4];
        // horizontal: S[sx]*a0 + S[sx+1]*a1 (<= 255*2048 < 2^20)
        if (shared) {
#pragma unroll
          for (int k = 0; k < 4; ++k) h0[k] = hprev[k];
        } else {
          const uint32_t alo = __builtin_amdgcn_alignbyte(cur.a[r][1], cur.a[r][0], sh), ahi = __builtin_amdgcn_alignbyte(cur.a[r][2], cur.a[r][1], sh);
#pragma unroll
          for (int k = 0; k < 4; ++k) h0[k] = udot2(__builtin_amdgcn_perm(ahi, alo, sel[k]), wp[k], 0u) & 0xFFFFF0u;
        }
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          h1[k] = udot2(__builtin_amdgcn_perm(bhi, blo, sel[k]), wp[k], 0u) & 0xFFFFF0u;
          // vertical: the 8-bit specialisation's ((b*(H>>4))>>16 + ... + 2) >> 2 (<= 255)
          v[k] = (mul_hi_u24(cur.wy0[r], h0[k]) + mul_hi_u24(cur.wy1[r], h1[k]) + 2u) >> 2;
          hprev[k] = h1[k];
        }
        // the four results (each <= 255) into one dword: two byte permutes and an or (left to
        // itself the optimiser rebuilds the shifts as multiply-adds with masks, twelve instructions)
        const uint32_t p01 = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0400u);
        const uint32_t p23 = __builtin_amdgcn_perm(v[3], v[2], 0x04000c0cu);
        *(GLOBAL_AS uint32_t*)(dframe + (dOff + (uint32_t)py * (uint32_t)pstride)) = (p01 | p23) & keep;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K2: per-cell FAST-9/16 + non-max suppression + two-threshold retry +
// ordered compaction.  One wave (a 64-thread workgroup) per grid cell.
//
// For a pixel with centre v and ring pixels I_k let
//   m = max over the 16 contiguous 9-arcs and both signs of min_arc (+-(v - I_k)).
// The pixel is a FAST corner at threshold t iff m > t, and OpenCV's corner
// score is then m - 1, independent of t.  One map of m therefore serves both
// thresholds of the reference's retry (extractor.cc:789-807).  cv::FAST runs
// on the cell's ROI, so suppression only sees neighbours inside the cell's own
// detection domain; domains tile [19, w-19) x [19, h-19) without overlap.
// ---------------------------------------------------------------------------
// Implementation: one wave per cell.
//   1. the cell's pixels plus a 3-pixel ring halo go to LDS, one lane per tile
//      row (dword-aligned 16-byte loads, re-based with v_alignbyte so that cell
//      column 0 sits at LDS column 4 and a 4-pixel group is dword-aligned),
//      together with a copy cut to 6 bits per pixel;
//   2. quick rejection for every pixel on the 6-bit copy, 8 pixels (4 columns
//      x 2 rows) per lane and four pixels per instruction: a 9-arc holds one
//      pixel of each of the 8 opposite ring pairs, so a dark (bright) corner
//      needs a dark (bright) pixel in the vertical and in the horizontal pair;
//      survivors (a few percent of the pixels) are queued in LDS with their side;
//   3. the exact one-sided m of the queued pixels, one per lane; those above
//      the threshold are written to the map of m and kept as the corner list;
//   4. suppression over the corner list, corners into bit masks in raster order;
//   5. output: a corner's place is the number of mask bits before it.
// A cell with more survivors than the queue holds (noise), or a threshold
// below the 6-bit test's resolution, is scored and suppressed pixel by pixel.
// TS = LDS row stride in bytes of the tile and of the m map (cell width <= TS - 8).
// Three-input minimum / maximum as one instruction each.  Written as min(min(a, b), c) the
// optimiser shares two-input minima between neighbouring windows (fewer operations in its own
// count) and the 40 three-input instructions of the exact score come out as 56 mixed ones.
#if defined(__AMDGCN__)
__device__ __forceinline__ int min3i(int a, int b, int c) {
  int r;
  asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ int max3i(int a, int b, int c) {
  int r;
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
#else
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
#endif

template <int TS>
__device__ __forceinline__ int fast_m(const uint8_t* c) {
  const int v = c[0];
  int d[16];
  d[0] = v - c[3 * TS + 0];
  d[1] = v - c[3 * TS + 1];
  d[2] = v - c[2 * TS + 2];
  d[3] = v - c[1 * TS + 3];
  d[4] = v - c[3];
  d[5] = v - c[-1 * TS + 3];
  d[6] = v - c[-2 * TS + 2];
  d[7] = v - c[-3 * TS + 1];
  d[8] = v - c[-3 * TS + 0];
  d[9] = v - c[-3 * TS - 1];
  d[10] = v - c[-2 * TS - 2];
  d[11] = v - c[-1 * TS - 3];
  d[12] = v - c[-3];
  d[13] = v - c[1 * TS - 3];
  d[14] = v - c[2 * TS - 2];
  d[15] = v - c[3 * TS - 1];
  // minimum / maximum of every 3 consecutive differences, then of every 9 (three such windows)
  int lo3[16], hi3[16], lo9[16], hi9[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    hi3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo9[k] = min3i(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    hi9[k] = max3i(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
  }
  int best_dark = max3i(lo9[0], lo9[1], lo9[2]), best_bright = min3i(hi9[0], hi9[1], hi9[2]);
#pragma unroll
  for (int k = 3; k < 15; k += 2) {
    best_dark = max3i(lo9[k], lo9[k + 1], best_dark);
    best_bright = min3i(hi9[k], hi9[k + 1], best_bright);
  }
  best_dark = max(best_dark, lo9[15]);
  best_bright = min(best_bright, hi9[15]);
  return max(max(best_dark, -best_bright), 0);
}

// The exact m of a pixel that the quick test left standing on one side: with s8 = 0 (bright
// side) or 0xff (dark side) and J = I ^ s8 (dark: 255 - I),
//   m = max(0, max over the 16 arcs of (min over the arc's 9 ring pixels of J) - (v ^ s8)).
// A pixel cannot be a dark and a bright corner at once (nine contiguous darker ring pixels leave
// seven others), so where one side's m is positive the other side's is zero: the side the quick
// test queued the pixel on gives the pixel's m whenever that m is positive.
template <int TS>
__device__ __forceinline__ int fast_m_sel(const uint8_t* c, int s8) {
  int d[16];
  d[0] = c[3 * TS + 0];
  d[1] = c[3 * TS + 1];
  d[2] = c[2 * TS + 2];
  d[3] = c[1 * TS + 3];
  d[4] = c[3];
  d[5] = c[-1 * TS + 3];
  d[6] = c[-2 * TS + 2];
  d[7] = c[-3 * TS + 1];
  d[8] = c[-3 * TS + 0];
  d[9] = c[-3 * TS - 1];
  d[10] = c[-2 * TS - 2];
  d[11] = c[-1 * TS - 3];
  d[12] = c[-3];
  d[13] = c[1 * TS - 3];
  d[14] = c[2 * TS - 2];
  d[15] = c[3 * TS - 1];
  const int v = (int)c[0] ^ s8;
#pragma unroll
  for (int k = 0; k < 16; ++k) d[k] ^= s8;
  // three-input min / max (v_min3 / v_max3 cost what the two-input forms cost): the minimum of
  // every 3 consecutive ring pixels, then of every 9 = three such windows, then the largest of
  // the sixteen: 16 + 16 + 8 instructions
  int lo3[16], lo9[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) lo3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) lo9[k] = min3i(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
  int best = max3i(lo9[0], lo9[1], lo9[2]);
#pragma unroll
  for (int k = 3; k < 15; k += 2) best = max3i(lo9[k], lo9[k + 1], best);
  best = max(best, lo9[15]);
  return max(best - v, 0);
}

// Quick corner test for the four pixels at window bytes 4..7 of window row WR, four pixels per
// instruction on pixel values cut to 6 bits (bits 6 and 7 of a byte free, so that a per-byte
// comparison can neither carry nor borrow across bytes and needs no saturation).  R = 8 window
// rows x 3 dwords of the 6-bit tile.
//
// With I6 = I >> 2, v6 = v >> 2 and c = (t + 1) >> 2: a ring pixel darker than the centre by more
// than t has I6 <= v6 - c, a brighter one I6 >= v6 + c (necessary, not sufficient: the cut
// loosens both by at most three grey levels).  A 9-arc holds one pixel of each of the 8 opposite
// ring pairs, so a dark (bright) corner needs a dark (bright) pixel in every pair; the vertical and
// the horizontal pair are tested (on the bench frames they leave 4.6 % of the pixels standing, the
// two diagonal pairs would bring that to 4.1 % at twice the cost).  Per byte, with nt = 0x7f + c - v6 and ug = 0x7f + c + v6
// (both below 0x100 for 1 <= c <= 64):
//   bit 7 of (I6 + nt) = "not dark",   bit 7 of (ug - I6) = "not bright".
// k = (0x7f + c) * 0x01010101.  Returns in bit 7 of each byte of dark / bright whether the
// pixel can still be a dark / bright corner.
template <int WR>
__device__ __forceinline__ void quick4(const uint32_t (&R)[8][3], uint32_t k, uint32_t& dark, uint32_t& bright) {
  const uint32_t V = R[WR][1];
  const uint32_t nt = k - V;
  const uint32_t ug = k + V;
  uint32_t nd, nb;
  {   // (0,3) / (0,-3): aligned
    const uint32_t a = R[WR + 3][1], b = R[WR - 3][1];
    nd = (a + nt) & (b + nt);
    nb = (ug - a) & (ug - b);
  }
  {   // (3,0) / (-3,0): bytes 7..10 and 1..4 of the centre row
    const uint32_t a = __builtin_amdgcn_alignbyte(R[WR][2], R[WR][1], 3), b = __builtin_amdgcn_alignbyte(R[WR][1], R[WR][0], 1);
    nd |= (a + nt) & (b + nt);
    nb |= (ug - a) & (ug - b);
  }
  dark = ~nd & 0x80808080u;
  bright = ~nb & 0x80808080u;
}

// The tile cut to 6 bits per pixel (quick test), one lane per tile row.
template <int TS>
__device__ __forceinline__ void fast_cut_tile(const uint8_t* tile, uint8_t* q6, int rows, int lane) {
  for (int r = lane; r < rows; r += 64) {
#pragma unroll
    for (int n = 0; n < TS / 16; ++n) {
      const uint4 o = *(const uint4*)(tile + r * TS + 16 * n);
      uint4 hv;
      hv.x = (o.x >> 2) & 0x3f3f3f3fu; hv.y = (o.y >> 2) & 0x3f3f3f3fu;
      hv.z = (o.z >> 2) & 0x3f3f3f3fu; hv.w = (o.w >> 2) & 0x3f3f3f3fu;
      *(uint4*)(q6 + r * TS + 16 * n) = hv;
    }
  }
}

// Is m (= q[0]) above its eight neighbours in the map of m?  All eight are loaded before the
// compare: written as a chain of && the test became eight dependent LDS round trips with a branch
// each (detection alone 0.886 -> 0.855 ms per 1024 frames).
template <int TS>
__device__ __forceinline__ bool nms_is_max(const uint8_t* q, int m) {
  const int n0 = q[-1], n1 = q[1], n2 = q[-TS - 1], n3 = q[-TS], n4 = q[-TS + 1], n5 = q[TS - 1], n6 = q[TS],
            n7 = q[TS + 1];
  return m > max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
}

// What a detection pass leaves behind for the output step.
struct FastPass {
  bool anyHi;      // some suppressed corner has m > thHi
  bool perPixel;   // the pass scored every pixel (no corner list)
  int ncorner;     // corner list: queue[0 .. ncorner) = y << 7 | x of the pixels with m > tlow
};

// One detection pass of a cell at threshold tlow (steps 2-4 of the kernel below): quick
// rejection, exact m for the survivors, suppression.  bitsHi collects the suppressed corners
// with m > thHi, bitsLo (may be null) those with m > thLo.  q6 = the tile cut to 6 bits; lr, gx
// = the lane's row and column in the grid of quick-test tasks (4 columns x 2 rows of pixels per
// task, ngx tasks per row, rpt task rows per trip of the wave).
template <int TS>
__device__ __forceinline__ FastPass fast_pass(const uint8_t* tile, const uint8_t* q6, uint8_t* mm, uint16_t* queue,
                                              int* qcnt, int qcap, int cw, int ch, int lane, int lr, int gx, int rpt,
                                              int tlow, int thHi, uint32_t* bitsHi, int thLo, uint32_t* bitsLo) {
  FastPass res;
  const int np = cw * ch;
  const int nry = (ch + 1) >> 1;
  const uint32_t c6 = (uint32_t)(tlow + 1) >> 2;
  if (lane == 0) qcnt[0] = 0;
  wave_sync();
  // ---- 2. quick rejection, 4 columns x 2 rows per lane ----------------------
  // Survivors are queued as (task row << 10 | task column << 5 | flag bit): bits 3..9 of an entry
  // are the pixel's column, bit 2 its row inside the task, bit 1 says that the task lies one trip
  // (rpt task rows) below the entry's task row, bit 0 is the side (0 dark, 1 bright); a pixel that
  // can be either is queued twice.  The position comes from an LDS counter, so the
  // order of the queue is arbitrary - nothing below depends on it.
  if (c6 >= 1) {
    const uint32_t k = (0x7fu + c6) * 0x01010101u;
    const bool laneOn = lr < rpt;
    const uint8_t* wp = q6 + (2 * lr) * TS + 4 * gx;
    uint32_t base = ((uint32_t)lr << 10) | ((uint32_t)gx << 5);
    // Two trips of the task grid per turn: the flags of the second go into the free bits of the
    // first's word (shifted up by two), so that a pair of trips costs one trip through the LDS
    // counter and one append loop, and the window rows of both are requested together.
    auto flags = [&](const uint8_t* p) -> uint32_t {
      uint32_t R[8][3];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t* rp = (const uint32_t*)(p + i * TS);
        R[i][0] = rp[0]; R[i][1] = rp[1]; R[i][2] = rp[2];
      }
      uint32_t d0, b0, d1, b1;
      quick4<3>(R, k, d0, b0);   // upper row of the task
      quick4<4>(R, k, d1, b1);   // lower row
      // dark flags in bits 0, 8, 16, 24 (upper row) and 4, 12, 20, 28 (lower row), bright flags
      // one bit above them.  Pixels of the last column group / row that lie outside the cell
      // are not masked here: the consumers of the queue skip them.
      return (d0 >> 7) | (d1 >> 3) | (b0 >> 6) | (b1 >> 2);
    };
    const int step = 2 * rpt * TS;
    for (int r0 = 0; r0 < nry; r0 += 2 * rpt) {
      if (laneOn && r0 + lr < nry) {
        uint32_t F = flags(wp);
        if (r0 + rpt + lr < nry) F |= flags(wp + step) << 2;
        if (F) {
          const int n = __popc(F);
          const int at = atomicAdd(qcnt, n);
          if (at + n <= qcap) {   // else the count alone matters: the cell is scored pixel by pixel
            uint16_t* qp = queue + at;
            while (F) {
              const int bit = __ffs(F) - 1;
              F &= F - 1;
              *qp++ = (uint16_t)(base | (uint32_t)bit);
            }
          }
        }
      }
      wp += 2 * step;
      base += (uint32_t)(2 * rpt) << 10;
    }
  }
  wave_sync();
  // the map of m shares its LDS with the 6-bit tile, which the quick test is done with
  {
    const int mmq = ((ch + 2) * TS) >> 4;
    for (int i = lane; i < mmq; i += 64) ((uint4*)mm)[i] = make_uint4(0, 0, 0, 0);
  }
  wave_sync();

  const int nq = qcnt[0];
  res.perPixel = c6 < 1 || nq > qcap;
  res.ncorner = 0;
  int anyHi = 0;
  if (res.perPixel) {
    // ---- 3'. / 4'. noise-like content (more survivors than the queue holds), or a threshold
    // below the 6-bit test's resolution: every pixel is scored and suppressed
    const float rcw = 1.0f / (float)cw;
    for (int qi = lane; qi < np; qi += 64) {
      const int y = (int)(((float)qi + 0.5f) * rcw), x = qi - y * cw;
      mm[(y + 1) * TS + x + 4] = (uint8_t)fast_m<TS>(tile + (y + 3) * TS + x + 4);
    }
    wave_sync();
    for (int qi = lane; qi < np; qi += 64) {
      const int y = (int)(((float)qi + 0.5f) * rcw), x = qi - y * cw;
      const uint8_t* q = mm + (y + 1) * TS + x + 4;
      const int m = q[0];
      if (m > tlow) {
        const bool isMax = nms_is_max<TS>(q, m);
        if (isMax) {
          if (m > thHi) { atomicOr(&bitsHi[qi >> 5], 1u << (qi & 31)); anyHi = 1; }
          if (bitsLo && m > thLo) atomicOr(&bitsLo[qi >> 5], 1u << (qi & 31));
        }
      }
    }
  } else {
    // ---- 3. exact m for the survivors, one per lane; those with m > tlow (the corners at this
    // threshold) are written to the map and compacted to the front of the queue (a trip reads
    // its 64 entries before it writes, and writes no further than it has read)
    int nc = 0;
    for (int q0 = 0; q0 < nq; q0 += 64) {
      const int qi = q0 + lane;
      const bool have = qi < nq;
      const uint32_t e = have ? (uint32_t)queue[qi] : 0u;
      const int x = (int)((e >> 3) & 127u),
                y = (int)(((e >> 10) << 1) | ((e >> 2) & 1u)) + ((e & 2u) ? 2 * rpt : 0);
      int m = 0;
      if (have && x < cw && y < ch) m = fast_m_sel<TS>(tile + (y + 3) * TS + x + 4, (e & 1u) ? 0 : 0xff);
      const bool corner = m > tlow;
      const unsigned long long mask = __ballot(corner);
      if (corner) {
        mm[(y + 1) * TS + x + 4] = (uint8_t)m;
        queue[nc + mbcnt64(mask)] = (uint16_t)((y << 7) | x);
      }
      nc += __popcll(mask);
    }
    res.ncorner = nc;
    wave_sync();
    // ---- 4. suppression inside the cell ---------------------------------------
    for (int q0 = 0; q0 < nc; q0 += 64) {
      const int qi = q0 + lane;
      if (qi < nc) {
        const int id = queue[qi], x = id & 127, y = id >> 7;
        const uint8_t* q = mm + (y + 1) * TS + x + 4;
        const int m = q[0];
        const bool isMax = nms_is_max<TS>(q, m);
        if (isMax) {
          const int p = y * cw + x;
          if (m > thHi) { atomicOr(&bitsHi[p >> 5], 1u << (p & 31)); anyHi = 1; }
          if (bitsLo && m > thLo) atomicOr(&bitsLo[p >> 5], 1u << (p & 31));
        }
      }
    }
  }
  res.anyHi = __ballot(anyHi) != 0ull;
  wave_sync();
  return res;
}

template <int TS>
__global__ __launch_bounds__(64) void k_fast_cells(const Plan* __restrict__ plan,
                                                   const FastCell* __restrict__ cells,
                                                   const uint8_t* __restrict__ pyr,
                                                   uint32_t* __restrict__ cand,
                                                   int* __restrict__ cellCount, int f0, int cell0,
                                                   int ncells, int nunits, int chunk) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int lane = threadIdx.x;
  const int unit = xcd_contiguous_unit(chunk);
  if (unit >= nunits) return;
  const int fr = unit / ncells;
  const int f = fr + f0;
  const int cellg = unit - fr * ncells + cell0;
  // all the kernel needs to know about the cell: one 32-byte record of the plan's cell table
  const FastCell c = cells[cellg];
  // (everything that comes from the plan is read here, in one batch of scalar loads beside the
  // record's: a load further down would wait for its own round trip)
  const int ini = plan->ini_th, mn = plan->min_th;
  const int qcap = plan->fast_qcap;
  const bool twoPass = plan->fast_two_pass != 0;
  const size_t candFrame = plan->cand_frame;
  const int cellsPerFrame = plan->cells_per_frame;
  const int X0 = (int)(c.xy & 0xFFFF), Y0 = (int)(c.xy >> 16);
  const int cw = (int)(c.geo & 0xFF), ch = (int)((c.geo >> 8) & 0xFF);
  const int np = cw * ch;
  const int nwords = (np + 31) >> 5;

  uint8_t* tile = smem;                                   // fast_tile_rows x TS
  uint8_t* q6 = tile + plan->fast_tile_rows * TS;         // the same pixels cut to 6 bits (quick test)
  uint8_t* mm = q6;                                       // fast_mm_rows x TS, once the quick test is done
  uint16_t* queue = (uint16_t*)(q6 + plan->fast_tile_rows * TS);
  uint32_t* bitsIni = (uint32_t*)(queue + qcap);
  uint32_t* bitsMin = bitsIni + plan->fast_words;
  uint16_t* prefix = (uint16_t*)(bitsMin + plan->fast_words);          // corners before each mask word
  // (the offset is rounded, not the pointer: a pointer that went through an integer would be
  // accessed with flat instructions instead of LDS ones)
  int* qcnt = (int*)(smem + ((2 * plan->fast_tile_rows * TS + 2 * plan->fast_qcap + 10 * plan->fast_words + 3) & ~3));

  // ---- 1. tile: LDS (row r, col c) = level pixel (X0 - 4 + c, Y0 - 3 + r) ----
  {
    // planes and rows are 64-byte aligned; the record holds the aligned start of the first row
    const int shift = (int)((c.geo >> 20) & 3), pstride = (int)(c.geo >> 24) << 6;
    const uint8_t* g0 = pyr + ((size_t)c.pyr64 << 6) + (size_t)f * ((size_t)c.plane64 << 6) + c.tile;
    // one lane = one tile row: dword-aligned 16-byte loads plus one more dword, shifted into
    // place with v_alignbyte, 16-byte LDS stores of the pixels and of their 6-bit cut
    for (int r = lane; r < ch + 6; r += 64) {
      const uint8_t* gp = g0 + (size_t)r * pstride;
      U4A4 v[TS / 16];
#pragma unroll
      for (int n = 0; n < TS / 16; ++n) v[n] = *(const U4A4*)(gp + 16 * n);
      const uint32_t vlast = *(const uint32_t*)(gp + TS);
#pragma unroll
      for (int n = 0; n < TS / 16; ++n) {
        const uint32_t nx = n + 1 < TS / 16 ? v[n + 1 < TS / 16 ? n + 1 : n].x : vlast;
        uint4 o;
        o.x = __builtin_amdgcn_alignbyte(v[n].y, v[n].x, shift);
        o.y = __builtin_amdgcn_alignbyte(v[n].z, v[n].y, shift);
        o.z = __builtin_amdgcn_alignbyte(v[n].w, v[n].z, shift);
        o.w = __builtin_amdgcn_alignbyte(nx, v[n].w, shift);
        *(uint4*)(tile + r * TS + 16 * n) = o;
        uint4 hv;
        hv.x = (o.x >> 2) & 0x3f3f3f3fu; hv.y = (o.y >> 2) & 0x3f3f3f3fu;
        hv.z = (o.z >> 2) & 0x3f3f3f3fu; hv.w = (o.w >> 2) & 0x3f3f3f3fu;
        *(uint4*)(q6 + r * TS + 16 * n) = hv;
      }
    }
    for (int i = lane; i < nwords; i += 64) { bitsIni[i] = 0; bitsMin[i] = 0; }
  }
  wave_sync();

  // quick-test task grid: ngx tasks of 4 columns per row of tasks, rpt rows of tasks per trip
  const int ngx = (cw + 3) >> 2;
  const uint32_t magic = c.magic;
  const int rpt = (int)((64u * magic) >> 16);
  const int lr = (int)(((uint32_t)lane * magic) >> 16), gx = lane - lr * ngx;

  // ---- 2.-4. detection at iniThFAST, and only in a cell that yields nothing there again at
  // minThFAST (extractor.cc:789-807).  Survivors of the quick test are the pixels whose bound
  // on m exceeds the threshold of the pass, so the first pass queues (and scores exactly) far
  // fewer pixels than one pass at the lower threshold would; the few cells that need the
  // second pass repeat the quick test.
  // When iniThFAST <= minThFAST (or the plan asks for it) one pass at the lower threshold serves
  // both.
  FastPass ps;
  bool useIni;
  if (ini > mn && twoPass) {
    ps = fast_pass<TS>(tile, q6, mm, queue, qcnt, qcap, cw, ch, lane, lr, gx, rpt, ini, ini, bitsIni, 0, nullptr);
    useIni = ps.anyHi;
    if (!useIni) {
      // the first pass's map of m lies where the 6-bit tile was: make that again (the second
      // pass clears the map and writes every score it needs afresh)
      fast_cut_tile<TS>(tile, q6, ch + 6, lane);
      ps = fast_pass<TS>(tile, q6, mm, queue, qcnt, qcap, cw, ch, lane, lr, gx, rpt, mn, mn, bitsMin, 0, nullptr);
    }
  } else {
    ps = fast_pass<TS>(tile, q6, mm, queue, qcnt, qcap, cw, ch, lane, lr, gx, rpt, min(ini, mn), ini, bitsIni, mn, bitsMin);
    useIni = ps.anyHi;
  }
  const uint32_t* bits = useIni ? bitsIni : bitsMin;

  // ---- 5. ordered output: the corners of the chosen mask in raster order -----------
  uint32_t* out = cand + (size_t)f * candFrame + c.cand;
  int running = 0;
  if (!ps.perPixel) {
    // a corner's place = mask bits before it: those of the earlier words (scan) + those below
    // it in its own word
    for (int wbase = 0; wbase < nwords; wbase += 64) {
      const int wi = wbase + lane;
      const uint32_t word = wi < nwords ? bits[wi] : 0u;
      const int c = __popc(word);
      const int inc = wave_incl_scan(c);
      if (wi < nwords) prefix[wi] = (uint16_t)(running + inc - c);
      running += __shfl(inc, 63, 64);
    }
    wave_sync();
    for (int q0 = 0; q0 < ps.ncorner; q0 += 64) {
      const int qi = q0 + lane;
      if (qi < ps.ncorner) {
        const int id = queue[qi], x = id & 127, y = id >> 7;
        const int p = y * cw + x;
        const uint32_t word = bits[p >> 5], bit = 1u << (p & 31);
        if (word & bit) {
          const int o = (int)prefix[p >> 5] + __popc(word & (bit - 1u));
          const int m = mm[(y + 1) * TS + x + 4];
          out[o] = (uint32_t)(X0 + x - ORBX_DET_ORG) | ((uint32_t)(Y0 + y - ORBX_DET_ORG) << 12) |
                   ((uint32_t)(m - 1) << 24);
        }
      }
    }
  } else {
    const float rcw = 1.0f / (float)cw;
    for (int wbase = 0; wbase < nwords; wbase += 64) {
      const int wi = wbase + lane;
      uint32_t word = wi < nwords ? bits[wi] : 0u;
      const int c = __popc(word);
      const int inc = wave_incl_scan(c);
      int o = running + inc - c;
      while (word) {
        const int b = __ffs(word) - 1;
        word &= word - 1;
        const int p = wi * 32 + b;
        const int y = (int)(((float)p + 0.5f) * rcw);
        const int x = p - y * cw;
        const int m = mm[(y + 1) * TS + x + 4];
        out[o++] = (uint32_t)(X0 + x - ORBX_DET_ORG) | ((uint32_t)(Y0 + y - ORBX_DET_ORG) << 12) |
                   ((uint32_t)(m - 1) << 24);
      }
      running += __shfl(inc, 63, 64);
    }
  }
  if (lane == 0) cellCount[(size_t)f * cellsPerFrame + cellg] = running;
}

// ---------------------------------------------------------------------------
// K4: quadtree distribution.  One wave per (frame, level).
//
// Keys are 32-bit: x | y << 12 | score << 24 with x, y relative to (16,16).
// The node list is an array kept in list order; every node owns a contiguous
// segment of one of two key buffers and a split writes the four children into
// the same range of the other buffer as a stable 4-way partition, so key
// order inside a node is the parent's order, as in DivideNode.
// A pass of the reference prepends children as they are created and erases
// the parents, so the list after a pass is: the children in reverse creation
// order, followed by the surviving old nodes in their old order.
//
// A pass is executed in three data-parallel steps over the nodes to split,
// taken in the reference's processing order (list order in the whole-level
// passes, sorted order from the back in the final rounds):
//   count   - per node, keys per quadrant;
//   place   - exclusive scan of the non-empty child counts gives every child
//             its creation index; in the final rounds the running node total
//             also gives the node at which the reference stops (:715-716);
//   scatter - partition the keys and write the child records.
// Nodes with at most OCT_SMALL keys are handled one node per lane (most nodes
// in the later passes hold a handful of keys); larger ones by the whole wave
// with ballots.  The key buffers live in LDS when the level's candidates fit,
// else in the worst-case-sized global buffers.
// ---------------------------------------------------------------------------
#define OCT_SMALL 32

struct QNode {
  short ulx, uly, brx, bry;
  int start;
  int cnt;  // bits 0..30 key count (0 = erased), bit 31 = key buffer id
};

__device__ __forceinline__ int qcount(const QNode& n) { return n.cnt & 0x7fffffff; }

struct OctState {
  QNode* cur;        // node list, list order
  QNode* nxt;
  QNode* newch;      // children of the current pass, creation order
  unsigned short* ord;   // list positions of the nodes to split, processing order
  int4* cnt4;        // per ord entry: keys per quadrant
  int* cbase;        // per ord entry: creation index of its first child
  uint32_t* bufA;
  uint32_t* bufB;
};

__device__ __forceinline__ int quadrant_of(uint32_t key, int mx, int my) {
  return ((int)(key & 0xFFF) >= mx ? 1 : 0) + ((int)((key >> 12) & 0xFFF) >= my ? 2 : 0);
}

// count step for ord[0..m)
__device__ __forceinline__ void oct_count(const OctState& st, int m) {
  const int lane = lane_id();
  for (int base = 0; base < m; base += 64) {
    const int t = base + lane;
    QNode nd;
    nd.cnt = 0;
    if (t < m) nd = st.cur[st.ord[t]];
    const int cnt = qcount(nd);
    const bool big = cnt > OCT_SMALL;
    if (t < m && !big) {
      const uint32_t* src = (((unsigned)nd.cnt >> 31) ? st.bufB : st.bufA) + nd.start;
      const int mx = nd.ulx + ((nd.brx - nd.ulx + 1) >> 1), my = nd.uly + ((nd.bry - nd.uly + 1) >> 1);
      int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
      for (int k = 0; k < cnt; k += 4) {   // 4 independent loads per trip; the tail re-reads key 0
        uint32_t kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kk[u] = src[k + u < cnt ? k + u : 0];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = k + u < cnt ? quadrant_of(kk[u], mx, my) : -1;
          c0 += q == 0; c1 += q == 1; c2 += q == 2; c3 += q == 3;
        }
      }
      st.cnt4[t] = make_int4(c0, c1, c2, c3);
    }
    unsigned long long bigmask = __ballot(big);
    while (bigmask) {
      const int bl = __ffsll((long long)bigmask) - 1;
      bigmask &= bigmask - 1;
      const QNode bn = st.cur[st.ord[base + bl]];
      const int bc = qcount(bn);
      const uint32_t* src = (((unsigned)bn.cnt >> 31) ? st.bufB : st.bufA) + bn.start;
      const int mx = bn.ulx + ((bn.brx - bn.ulx + 1) >> 1), my = bn.uly + ((bn.bry - bn.uly + 1) >> 1);
      int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
      for (int kb = 0; kb < bc; kb += 64) {
        const int i = kb + lane;
        const bool valid = i < bc;
        const int q = valid ? quadrant_of(src[i], mx, my) : -1;
        c0 += __popcll(__ballot(q == 0));
        c1 += __popcll(__ballot(q == 1));
        c2 += __popcll(__ballot(q == 2));
        c3 += __popcll(__ballot(q == 3));
      }
      if (lane == 0) st.cnt4[base + bl] = make_int4(c0, c1, c2, c3);
    }
  }
}

// place step: creation index of every node's first child; returns the number of nodes
// actually split (all of them, or up to the one that lifts the node total to N when
// stopAtN), the number of children created and how many of those can be split again.
__device__ __forceinline__ int oct_place(const OctState& st, int m, int n0, int N, bool stopAtN, int* nchOut,
                         int* nExpOut) {
  const int lane = lane_id();
  int childRun = 0, nodeRun = n0, mEff = m;
  bool stopped = false;
  for (int base = 0; base < m && !stopped; base += 64) {
    const int t = base + lane;
    int mch = 0;
    if (t < m) {
      const int4 c = st.cnt4[t];
      mch = (c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0);
    }
    const int inc = wave_incl_scan(mch);
    if (t < m) st.cbase[t] = childRun + inc - mch;
    // node total after splitting entries 0..t: + children - 1 per split node
    const int nodesAfter = nodeRun + inc - (lane + 1);
    if (stopAtN) {
      const unsigned long long hit = __ballot(t < m && nodesAfter >= N);
      if (hit) {
        const int first = __ffsll((long long)hit) - 1;
        mEff = base + first + 1;
        childRun += __shfl(inc, first, 64);
        stopped = true;
        break;
      }
    }
    const int last = min(63, m - 1 - base);
    childRun += __shfl(inc, last, 64);
    nodeRun += __shfl(inc, last, 64) - (last + 1);
  }
  // children that hold more than one key, among the nodes actually split
  int nexp = 0;
  for (int base = 0; base < mEff; base += 64) {
    const int t = base + lane;
    int e = 0;
    if (t < mEff) {
      const int4 c = st.cnt4[t];
      e = (c.x > 1) + (c.y > 1) + (c.z > 1) + (c.w > 1);
    }
    nexp += wave_sum(e);
  }
  *nchOut = childRun;
  *nExpOut = nexp;
  return mEff;
}

__device__ __forceinline__ void oct_write_children(const OctState& st, const QNode& nd, int4 c, int cb,
                                                   int mx, int my) {
  const int nb = (int)(((unsigned)nd.cnt >> 31) ^ 1u) << 31;
  const int cc[4] = {c.x, c.y, c.z, c.w};
  int start = nd.start;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (cc[q] > 0) {
      QNode ch;
      ch.ulx = (q & 1) ? (short)mx : nd.ulx;
      ch.brx = (q & 1) ? nd.brx : (short)mx;
      ch.uly = (q & 2) ? (short)my : nd.uly;
      ch.bry = (q & 2) ? nd.bry : (short)my;
      ch.start = start;
      ch.cnt = cc[q] | nb;
      st.newch[cb++] = ch;
    }
    start += cc[q];
  }
}

// scatter step for ord[0..mEff)
__device__ __forceinline__ void oct_scatter(const OctState& st, int mEff) {
  const int lane = lane_id();
  for (int base = 0; base < mEff; base += 64) {
    const int t = base + lane;
    QNode nd;
    nd.cnt = 0;
    int p = 0;
    if (t < mEff) { p = st.ord[t]; nd = st.cur[p]; }
    const int cnt = qcount(nd);
    const bool big = cnt > OCT_SMALL;
    if (t < mEff && !big) {
      const int buf = (int)((unsigned)nd.cnt >> 31);
      const uint32_t* src = (buf ? st.bufB : st.bufA) + nd.start;
      uint32_t* dst = (buf ? st.bufA : st.bufB) + nd.start;
      const int mx = nd.ulx + ((nd.brx - nd.ulx + 1) >> 1), my = nd.uly + ((nd.bry - nd.uly + 1) >> 1);
      const int4 c = st.cnt4[t];
      int o0 = 0, o1 = c.x, o2 = c.x + c.y, o3 = c.x + c.y + c.z;
      for (int k = 0; k < cnt; k += 4) {
        uint32_t kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kk[u] = src[k + u < cnt ? k + u : 0];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (k + u < cnt) {
            const int q = quadrant_of(kk[u], mx, my);
            const int pos = q == 0 ? o0 : q == 1 ? o1 : q == 2 ? o2 : o3;
            dst[pos] = kk[u];
            o0 += q == 0; o1 += q == 1; o2 += q == 2; o3 += q == 3;
          }
        }
      }
      oct_write_children(st, nd, c, st.cbase[t], mx, my);
      st.cur[p].cnt = 0;
    }
    unsigned long long bigmask = __ballot(big);
    while (bigmask) {
      const int bl = __ffsll((long long)bigmask) - 1;
      bigmask &= bigmask - 1;
      const int bp = st.ord[base + bl];
      const QNode bn = st.cur[bp];
      const int bc = qcount(bn);
      const int buf = (int)((unsigned)bn.cnt >> 31);
      const uint32_t* src = (buf ? st.bufB : st.bufA) + bn.start;
      uint32_t* dst = (buf ? st.bufA : st.bufB) + bn.start;
      const int mx = bn.ulx + ((bn.brx - bn.ulx + 1) >> 1), my = bn.uly + ((bn.bry - bn.uly + 1) >> 1);
      const int4 c = st.cnt4[base + bl];
      int r0 = 0, r1 = c.x, r2 = c.x + c.y, r3 = c.x + c.y + c.z;
      for (int kb = 0; kb < bc; kb += 64) {
        const int i = kb + lane;
        const bool valid = i < bc;
        const uint32_t key = valid ? src[i] : 0u;
        const int q = valid ? quadrant_of(key, mx, my) : -1;
        const unsigned long long m0 = __ballot(q == 0), m1 = __ballot(q == 1), m2 = __ballot(q == 2),
                                 m3 = __ballot(q == 3);
        if (valid) {
          int pos;
          if (q == 0) pos = r0 + mbcnt64(m0);
          else if (q == 1) pos = r1 + mbcnt64(m1);
          else if (q == 2) pos = r2 + mbcnt64(m2);
          else pos = r3 + mbcnt64(m3);
          dst[pos] = key;
        }
        r0 += __popcll(m0); r1 += __popcll(m1); r2 += __popcll(m2); r3 += __popcll(m3);
      }
      if (lane == 0) {
        oct_write_children(st, bn, c, st.cbase[base + bl], mx, my);
        st.cur[bp].cnt = 0;
      }
    }
  }
}

// new list = reverse(newch[0..nch)) ++ [old nodes with cnt != 0, old order]
__device__ __forceinline__ int rebuild_list(const QNode* cur, int n, const QNode* newch, int nch, QNode* nxt) {
  const int lane = lane_id();
  for (int i = lane; i < nch; i += 64) nxt[i] = newch[nch - 1 - i];
  int m = nch;
  for (int base = 0; base < n; base += 64) {
    const int p = base + lane;
    const bool keep = p < n && cur[p].cnt != 0;
    const unsigned long long mask = __ballot(keep);
    if (keep) nxt[m + mbcnt64(mask)] = cur[p];
    m += __popcll(mask);
  }
  return m;
}

// records of the children that can still be split, in creation order:
// high word = (key count << 12 | ulx) (the reference's comparison key),
// low word = position of the child in the rebuilt list.
__device__ __forceinline__ int build_records(const QNode* newch, int nch, orbx_sort::elem_t* rec) {
  const int lane = lane_id();
  int nrec = 0;
  for (int base = 0; base < nch; base += 64) {
    const int c = base + lane;
    const int cnt = c < nch ? qcount(newch[c]) : 0;
    const bool keep = cnt > 1;
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      const uint32_t key = ((uint32_t)cnt << 12) | (uint32_t)(unsigned short)newch[c].ulx;
      rec[nrec + mbcnt64(mask)] = ((orbx_sort::elem_t)key << 32) | (uint32_t)(nch - 1 - c);
    }
    nrec += __popcll(mask);
  }
  return nrec;
}

// std::sort(v, v + n) of the split records, wave-parallel.  libstdc++'s introsort is
// replayed partition by partition (see stdsort_emul.h for the serial form): the result of
// one unguarded Hoare partition follows from the segment's contents before it -
// the k-th swap exchanges the k-th element from the left that is not less than the pivot
// with the k-th from the right that is not greater, as long as the former lies left of
// the latter - so stopper lists, swap count and cut come from ballots instead of a
// serial scan.  The closing insertion sort is stable and never moves an element across
// a cut, i.e. it is a stable sort inside each final segment.  A segment whose depth
// budget is spent is heap-sorted serially like the original (rare).
// Scratch: posA/posB (n entries), segmask (n / 64 + 1 words of 64 bits), tmp (n elements), stk (3*64 ints).
// words of list / stack scratch the sorts need for n records (oct_sort: 3 * 64; oct_sort_team: two
// lists of n / 16 + 2 segments and their lengths)
__host__ __device__ constexpr int debug_sort_stack_ints(int n, bool team) { return team ? 6 * (n / 16 + 2) + 2 : 192; }

// One unguarded Hoare partition of [first, last) around the median of three, as std::sort does
// it; returns the cut.  One wave; posA / posB: last - first entries each.
__device__ __forceinline__ int oct_partition(orbx_sort::elem_t* v, int first, int last, unsigned short* posA,
                                             unsigned short* posB) {
  using orbx_sort::less;
  const int lane = lane_id();
  const int mid = first + (last - first) / 2;
  if (lane == 0) orbx_sort::move_median_to_first(v, first, first + 1, mid, last - 1);
  wave_sync();
  const orbx_sort::elem_t P = v[first];
  const int lo = first + 1, hi = last;
  int nA = 0, nB = 0;
  for (int base = lo; base < hi; base += 64) {
    const int i = base + lane;
    const bool fl = i < hi && !less(v[i], P);
    const unsigned long long mask = __ballot(fl);
    if (fl) posA[nA + mbcnt64(mask)] = (unsigned short)i;
    nA += __popcll(mask);
  }
  for (int base = hi - 1; base >= lo; base -= 64) {
    const int i = base - lane;
    const bool fl = i >= lo && !less(P, v[i]);
    const unsigned long long mask = __ballot(fl);
    if (fl) posB[nB + mbcnt64(mask)] = (unsigned short)i;
    nB += __popcll(mask);
  }
  wave_sync();
  const int kmax = min(nA, nB);
  int kstar = 0;
  for (int base = 0; base < kmax; base += 64) {
    const int k = base + lane;
    const bool ok = k < kmax && posA[k] < posB[k];
    const unsigned long long mask = __ballot(ok);
    const int c = __popcll(mask);
    kstar += c;
    if (c < 64) break;
  }
  for (int k = lane; k < kstar; k += 64) {
    const int a = posA[k], b = posB[k];
    const orbx_sort::elem_t t = v[a];
    v[a] = v[b];
    v[b] = t;
  }
  const int ia = kstar < nA ? (int)posA[kstar] : 0x7fffffff;
  const int ib = kstar > 0 ? (int)posB[kstar - 1] : 0x7fffffff;
  return min(ia, ib);
}

// The closing pass: a stable sort inside every segment [s, e) between two cuts, by rank.  The
// cuts are bits of segmask (bit i of word i / 64 = a segment starts at i; bits 0 and n are set).
// `step` threads share the work, the caller's being number `me`.
__device__ __forceinline__ void oct_sort_segments(const orbx_sort::elem_t* v, int n, const unsigned long long* segmask,
                                                  orbx_sort::elem_t* tmp, int me, int step) {
  using orbx_sort::less;
  for (int i = me; i < n; i += step) {
    int c = i >> 6;
    const unsigned long long here = segmask[c];
    const unsigned long long upto = here & (~0ull >> (63 - (i & 63)));   // bits <= i
    const unsigned long long after = (i & 63) == 63 ? 0ull : here & (~0ull << ((i & 63) + 1));
    int s0, e;
    if (upto) s0 = (c << 6) + 63 - __clzll((long long)upto);
    else {
      int cc = c - 1;
      unsigned long long m = segmask[cc];
      while (!m) m = segmask[--cc];
      s0 = (cc << 6) + 63 - __clzll((long long)m);
    }
    if (after) e = (c << 6) + __ffsll((long long)after) - 1;
    else {
      int cc = c + 1;
      unsigned long long m = segmask[cc];
      while (!m) m = segmask[++cc];
      e = (cc << 6) + __ffsll((long long)m) - 1;
    }
    const orbx_sort::elem_t mine = v[i];
    int rank = s0;
    for (int j = s0; j < e; j += 4) {   // four loads in flight; the tail re-reads the last element
      orbx_sort::elem_t o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] = v[j + u < e ? j + u : e - 1];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        rank += (j + u < e && (less(o[u], mine) || (!less(mine, o[u]) && j + u < i))) ? 1 : 0;
    }
    tmp[rank] = mine;
  }
}

__device__ __forceinline__ void oct_sort(orbx_sort::elem_t* v, int n, unsigned short* posA, unsigned short* posB,
                         unsigned long long* segmask, orbx_sort::elem_t* tmp, int* stk) {
  const int lane = lane_id();
  if (n <= 1) return;
  for (int c = lane; c <= n >> 6; c += 64)
    segmask[c] = (c == 0 ? 1ull : 0ull) | (c == n >> 6 ? 1ull << (n & 63) : 0ull);
  wave_sync();
  if (n > 16) {
    int sp = 0;
    if (lane == 0) { stk[0] = 0; stk[1] = n; stk[2] = 2 * orbx_sort::floor_log2(n); }
    sp = 1;
    wave_sync();
    while (sp > 0) {
      --sp;
      int first = stk[3 * sp], last = stk[3 * sp + 1], depth = stk[3 * sp + 2];
      wave_sync();
      while (last - first > 16) {
        if (depth == 0) {
          if (lane == 0) orbx_sort::heap_sort_range(v, first, last);
          wave_sync();
          break;
        }
        --depth;
        const int cut = oct_partition(v, first, last, posA, posB);
        if (lane == 0) {
          stk[3 * sp] = cut; stk[3 * sp + 1] = last; stk[3 * sp + 2] = depth;
          segmask[cut >> 6] |= 1ull << (cut & 63);
        }
        ++sp;
        last = cut;
        wave_sync();
      }
    }
  }
  oct_sort_segments(v, n, segmask, tmp, lane, 64);
  wave_sync();
  for (int i = lane; i < n; i += 64) v[i] = tmp[i];
  wave_sync();
}

// The same sort by a team of NW waves.  Segments of a partition tree are independent, so the
// partitions are taken breadth first: a wave takes a segment of this round's list, keeps its left
// part and appends the right part to the next round's list - about log2(n / 16) rounds of one
// partition each instead of n / 16 partitions one after the other.  Which wave partitions a
// segment, and in which order segments are taken, does not change the result.
// seg: two lists of (first, last, depth) with n / 16 + 2 entries each, then the two list lengths.
// posA / posB are indexed from a segment's first element, so waves working on different segments
// use different entries.  All waves of the workgroup must call.
template <int NW>
__device__ __forceinline__ void oct_sort_team(orbx_sort::elem_t* v, int n, unsigned short* posA,
                                              unsigned short* posB, unsigned long long* segmask,
                                              orbx_sort::elem_t* tmp, int* seg) {
  constexpr int T = 64 * NW;
  const int lane = lane_id(), w = (int)threadIdx.y, tid = w * 64 + lane;
  if (n <= 1) return;
  for (int c = tid; c <= n >> 6; c += T)
    segmask[c] = (c == 0 ? 1ull : 0ull) | (c == n >> 6 ? 1ull << (n & 63) : 0ull);
  if (n > 16) {
    const int lcap = n / 16 + 2;
    int* len = seg + 6 * lcap;
    if (tid == 0) {
      seg[0] = 0; seg[1] = n; seg[2] = 2 * orbx_sort::floor_log2(n);
      len[0] = 1; len[1] = 0;
    }
    __syncthreads();
    for (int cur = 0;; cur ^= 1) {
      const int ncur = len[cur];
      if (ncur == 0) break;
      int* list = seg + 3 * lcap * cur;
      int* next = seg + 3 * lcap * (cur ^ 1);
      for (int sgi = w; sgi < ncur; sgi += NW) {
        int first = list[3 * sgi], last = list[3 * sgi + 1], depth = list[3 * sgi + 2];
        while (last - first > 16) {
          if (depth == 0) {
            if (lane == 0) orbx_sort::heap_sort_range(v, first, last);
            wave_sync();
            break;
          }
          --depth;
          const int cut = oct_partition(v, first, last, posA + first, posB + first);
          if (lane == 0) {
            atomicOr(&segmask[cut >> 6], 1ull << (cut & 63));
            if (last - cut > 16) {
              const int at = atomicAdd(&len[cur ^ 1], 1);
              next[3 * at] = cut; next[3 * at + 1] = last; next[3 * at + 2] = depth;
            }
          }
          last = cut;
          wave_sync();
        }
      }
      __syncthreads();
      if (tid == 0) len[cur] = 0;
      __syncthreads();
    }
  } else {
    __syncthreads();
  }
  oct_sort_segments(v, n, segmask, tmp, tid, T);
  __syncthreads();
  for (int i = tid; i < n; i += T) v[i] = tmp[i];
  __syncthreads();
}

// Test hook: sorts n records in LDS with oct_sort (NW = 1) or oct_sort_team and returns them.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_debug_sort(orbx_sort::elem_t* data, int n) {
  extern __shared__ __align__(16) uint8_t smem[];
  orbx_sort::elem_t* v = (orbx_sort::elem_t*)smem;
  orbx_sort::elem_t* tmp = v + n;
  unsigned long long* segmask = (unsigned long long*)(tmp + n);
  int* stk = (int*)(segmask + (n >> 6) + 1);
  unsigned short* posA = (unsigned short*)(stk + debug_sort_stack_ints(n, NW > 1));
  unsigned short* posB = posA + n;
  const int tid = (int)(threadIdx.y * 64 + threadIdx.x);
  for (int i = tid; i < n; i += 64 * NW) v[i] = data[i];
  __syncthreads();
  if constexpr (NW == 1) oct_sort(v, n, posA, posB, segmask, tmp, stk);
  else oct_sort_team<NW>(v, n, posA, posB, segmask, tmp, stk);
  __syncthreads();
  for (int i = tid; i < n; i += 64 * NW) data[i] = v[i];
}

// The tree of one (frame, level) once its candidates are counted: `st` holds the node arrays and
// the two key buffers.  Inlined once per kind of memory the caller puts them in, so that every
// access is an LDS or a global instruction rather than a flat one (a flat access to LDS takes the
// long way through the vector-memory pipeline and ties the two wait counters together).
__device__ __forceinline__ void octree_level(const Plan* __restrict__ plan, const LevelDesc& L, OctState st,
                                             orbx_sort::elem_t* rec, const int* cellpref, int* rootCnt,
                                             const uint32_t* __restrict__ candL, int nc, int total,
                                             uint32_t* __restrict__ out, int* __restrict__ selCountSlot) {
  const int lane = lane_id();
  const int nroots = L.n_roots;
  const float hX = L.hX;
  // flatten into B and count keys per root: root = int(x / hX) (extractor.cc:571-574)
  // four chunks per trip so that the (global) candidate loads of a trip are in flight together
  for (int base = 0; base < total; base += 256) {
    uint32_t key[4];
    bool valid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * 64 + lane;
      valid[u] = i < total;
      const int ic = valid[u] ? i : total - 1;
      int lo = 0, hi = nc;  // largest c with cellpref[c] <= ic
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cellpref[mid] <= ic) lo = mid; else hi = mid;
      }
      key[u] = candL[(size_t)lo * L.cand_cap_cell + (ic - cellpref[lo])];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * 64 + lane;
      int r = -1;
      if (valid[u]) {
        st.bufB[i] = key[u];
        r = (int)__fdiv_rn((float)(key[u] & 0xFFF), hX);
      }
      for (int rr = 0; rr < nroots; ++rr) {
        const unsigned long long m = __ballot(r == rr);
        if (lane == 0 && m) rootCnt[rr] += __popcll(m);
      }
    }
  }
  wave_sync();
  // scatter B -> A grouped by root, order kept
  {
    int rbase[ORBX_MAX_ROOTS];
    int acc = 0;
#pragma unroll
    for (int rr = 0; rr < ORBX_MAX_ROOTS; ++rr) {
      rbase[rr] = acc;
      acc += rr < nroots ? rootCnt[rr] : 0;
    }
    for (int base = 0; base < total; base += 64) {
      const int i = base + lane;
      const bool valid = i < total;
      const uint32_t key = valid ? st.bufB[i] : 0u;
      const int r = valid ? (int)__fdiv_rn((float)(key & 0xFFF), hX) : -1;
#pragma unroll
      for (int rr = 0; rr < ORBX_MAX_ROOTS; ++rr) {
        if (rr < nroots) {
          const unsigned long long m = __ballot(r == rr);
          if (r == rr) st.bufA[rbase[rr] + mbcnt64(m)] = key;
          rbase[rr] += __popcll(m);
        }
      }
    }
  }
  // root nodes (extractor.cc:558-586): empty ones are dropped
  int n = 0;
  {
    int acc = 0;
    const int hgt = L.h - 2 * ORBX_DET_ORG;  // maxBorderY - minBorderY
    for (int rr = 0; rr < nroots; ++rr) {
      const int c = rootCnt[rr];
      if (c > 0) {
        if (lane == 0) {
          QNode nd;
          nd.ulx = (short)(int)__fmul_rn(hX, (float)rr);
          nd.brx = (short)(int)__fmul_rn(hX, (float)(rr + 1));
          nd.uly = 0;
          nd.bry = (short)hgt;
          nd.start = acc;
          nd.cnt = c;  // buffer A
          st.cur[n] = nd;
        }
        ++n;
      }
      acc += c;
    }
  }
  wave_sync();

  const int N = L.quota;
  int nrec = 0;
  bool finish = false;
  // whole-level passes (extractor.cc:606-657: every node with more than one key, list order) until
  // one more could overshoot N, then the final rounds (extractor.cc:665-721: largest nodes first,
  // stop at N nodes); both kinds of pass run through the same three steps
  bool finalRounds = false;
  while (!finish) {
    const int prevSize = n;
    int m = 0;
    if (!finalRounds) {
      for (int base = 0; base < n; base += 64) {
        const int p = base + lane;
        const bool exp = p < n && qcount(st.cur[p]) > 1;
        const unsigned long long mask = __ballot(exp);
        if (exp) st.ord[m + mbcnt64(mask)] = (unsigned short)p;
        m += __popcll(mask);
      }
    } else {
      oct_sort(rec, nrec, st.ord, (unsigned short*)st.cbase, (unsigned long long*)st.newch,
               (orbx_sort::elem_t*)st.cnt4, (int*)st.nxt);
      for (int t = lane; t < nrec; t += 64) st.ord[t] = (unsigned short)(uint32_t)rec[nrec - 1 - t];
      m = nrec;
    }
    wave_sync();
    oct_count(st, m);
    wave_sync();
    int nch, nToExpand;
    const int mEff = oct_place(st, m, n, N, finalRounds, &nch, &nToExpand);
    wave_sync();
    oct_scatter(st, mEff);
    wave_sync();
    n = rebuild_list(st.cur, n, st.newch, nch, st.nxt);
    nrec = build_records(st.newch, nch, rec);
    { QNode* t = st.cur; st.cur = st.nxt; st.nxt = t; }
    wave_sync();
    if (n >= N || n == prevSize) finish = true;
    else if (n + nToExpand * 3 > N) finalRounds = true;
  }

  // best key per node: highest score, first one wins ties (extractor.cc:726-742)
  for (int p = lane; p < n; p += 64) {
    const QNode nd = st.cur[p];
    const uint32_t* keys = (((unsigned)nd.cnt >> 31) ? st.bufB : st.bufA) + nd.start;
    const int cnt = qcount(nd);
    uint32_t best = keys[0];
    for (int k = 1; k < cnt; k += 4) {
      uint32_t kk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) kk[u] = keys[k + u < cnt ? k + u : 0];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k + u < cnt && (kk[u] >> 24) > (best >> 24)) best = kk[u];
    }
    if (p < L.sel_cap) out[p] = best;
  }
  if (lane == 0) *selCountSlot = n < L.sel_cap ? n : L.sel_cap;
}

template <bool NODES_WS>
__global__ __launch_bounds__(64) void k_octree(const Plan* __restrict__ plan,
                                               const uint32_t* __restrict__ cand,
                                               const int* __restrict__ cellCount,
                                               uint32_t* __restrict__ flatA,
                                               uint32_t* __restrict__ flatB,
                                               uint32_t* __restrict__ sel,
                                               int* __restrict__ selCount,
                                               uint8_t* __restrict__ workspace, int f0, int l0) {
  extern __shared__ __align__(16) uint8_t smemDyn[];
  // Level-major grid (x = frame, y = level): the trees of level 0 are by far the longest and a CU
  // only holds a few of these waves at a time, so they are dispatched first and the short trees of
  // the small levels fill in behind them; consecutive workgroups (= frames) go to the 8 XCDs
  // round-robin, so every XCD gets its share of every level.
  const int l = (int)blockIdx.y + l0, f = (int)blockIdx.x + f0;
  const int lane = lane_id();
  const LevelDesc& L = plan->lv[l];
  const int cap = plan->oct_nodes_cap;
  // node arrays live in LDS; plans whose arrays exceed it (tens of thousands of
  // features per frame) use a per-(frame, level) slice of a global workspace instead
  uint8_t* smem;
  if constexpr (NODES_WS) smem = workspace + ((size_t)f * plan->nlevels + l) * plan->oct_ws_bytes;
  else smem = smemDyn;
  OctState st;
  st.cnt4 = (int4*)smem;                                    // 16-byte aligned first
  st.cur = (QNode*)(st.cnt4 + cap);
  st.nxt = st.cur + cap;
  st.newch = st.nxt + cap;
  orbx_sort::elem_t* rec = (orbx_sort::elem_t*)(st.newch + cap);
  st.cbase = (int*)(rec + cap);
  int* cellpref = st.cbase + cap;
  uint32_t* ldsKeys = (uint32_t*)(cellpref + plan->oct_cells_max + 1);
  st.ord = (unsigned short*)(ldsKeys + 2 * plan->oct_keys_lds);
  __shared__ int rootCnt[ORBX_MAX_ROOTS];

  // ---- candidate order = cell-row-major, then (y, x) inside a cell ---------
  const int nc = L.ncx * L.ncy;
  const int* cc = cellCount + (size_t)f * plan->cells_per_frame + L.cell_first;
  int total = 0;
  for (int base = 0; base < nc; base += 64) {
    const int c = base + lane;
    const int v = c < nc ? cc[c] : 0;
    const int inc = wave_incl_scan(v);
    if (c < nc) cellpref[c] = total + inc - v;
    total += __shfl(inc, 63, 64);
  }
  if (lane == 0) cellpref[nc] = total;
  if (lane < ORBX_MAX_ROOTS) rootCnt[lane] = 0;
  wave_sync();
  const uint32_t* candL = cand + (size_t)f * plan->cand_frame + L.cand_off;
  uint32_t* out = sel + (size_t)f * plan->sel_frame + L.sel_off;
  int* selCountSlot = selCount + (size_t)f * plan->nlevels + l;
  if (total <= plan->oct_keys_lds) {
    st.bufA = ldsKeys;
    st.bufB = ldsKeys + plan->oct_keys_lds;
    octree_level(plan, L, st, rec, cellpref, rootCnt, candL, nc, total, out, selCountSlot);
  } else {
    st.bufA = flatA + (size_t)f * plan->flat_frame + L.flat_off;
    st.bufB = flatB + (size_t)f * plan->flat_frame + L.flat_off;
    octree_level(plan, L, st, rec, cellpref, rootCnt, candL, nc, total, out, selCountSlot);
  }
}

// ---------------------------------------------------------------------------
// The same tree built by a team of NW waves, for batches too small to fill the chip with one
// wave per (frame, level): there the longest tree is the whole stage, and one wave spends it
// waiting for LDS.  Every key is handled by one thread in every pass instead of one node per
// lane: a prefix count of the keys per quadrant over the key array in its current order (four
// 16-bit counters packed in two words, one scan for all nodes at once) gives a node's four
// counts as the difference between its two ends, and a key's place in its child as its own
// prefix minus the one at its node's start - the stable partition of every node of a pass in
// one sweep.  own[] names the node (list position) that holds a key, OWN_NONE once that node
// is final.  Steps that are short and serial by nature (the list of nodes to split, the
// creation indices, the rebuilt list, the records, the sort of the final rounds) run on wave 0,
// which hands their scalar results to the other waves through bc[].
// Requires at most four roots and a key count within the LDS buffers; the caller falls back to
// the one-wave form otherwise.
// ---------------------------------------------------------------------------
#define OWN_NONE 0xFFFFu

struct TeamState {
  unsigned short* ownCur;   // per key position
  unsigned short* ownNxt;
  uint2* pref;              // per key position + 1: keys before it per class, (c0 | c1 << 16, c2 | c3 << 16)
  short* tOf;               // per list position: index in ord, -1 = not split in this pass
  uint2* wtot;              // 2 x NW wave totals of the scan
  int* bc;                  // 8 words: scalar results of wave 0's steps
};

__device__ __forceinline__ int pref_field(uint2 v, int q) {
  return (int)(((q & 2) ? v.y : v.x) >> ((q & 1) * 16)) & 0xFFFF;
}

// pref[i] = number of keys j < i of each class, cls(j) in {0,1,2,3} or -1 (not counted), for
// i in [0, total].  Thread tid takes four consecutive keys of every block of 4 * T.
template <int NW, class F>
__device__ __forceinline__ void team_prefix(int total, uint2* pref, uint2* wtot, F cls) {
  constexpr int T = 64 * NW;
  const int lane = lane_id(), w = (int)threadIdx.y, tid = w * 64 + lane;
  uint32_t carryLo = 0, carryHi = 0;
  int par = 0;
  for (int cb = 0; cb <= total; cb += 4 * T, par ^= 1) {
    const int i0 = cb + 4 * tid;
    uint32_t lo[4], hi[4], accLo = 0, accHi = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = i0 + j < total ? cls(i0 + j) : -1;
      lo[j] = accLo;
      hi[j] = accHi;
      accLo += (q == 0 ? 1u : 0u) + (q == 1 ? 0x10000u : 0u);
      accHi += (q == 2 ? 1u : 0u) + (q == 3 ? 0x10000u : 0u);
    }
    const uint32_t sLo = (uint32_t)wave_incl_scan((int)accLo), sHi = (uint32_t)wave_incl_scan((int)accHi);
    if (lane == 63) wtot[par * NW + w] = make_uint2(sLo, sHi);
    __syncthreads();
    uint32_t bLo = carryLo + sLo - accLo, bHi = carryHi + sHi - accHi;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      const uint2 t = wtot[par * NW + ww];
      if (ww < w) { bLo += t.x; bHi += t.y; }
      carryLo += t.x;
      carryHi += t.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j <= total) pref[i0 + j] = make_uint2(bLo + lo[j], bHi + hi[j]);
  }
  __syncthreads();
}

template <int NW>
__device__ __forceinline__ void team_octree_level(const LevelDesc& L, OctState st, TeamState ts,
                                                  orbx_sort::elem_t* rec, const int* cellpref,
                                                  const uint32_t* __restrict__ candL, int nc, int total,
                                                  uint32_t* __restrict__ out, int* __restrict__ selCountSlot) {
  constexpr int T = 64 * NW;
  const int lane = lane_id(), w = (int)threadIdx.y, tid = w * 64 + lane;
  const int nroots = L.n_roots;
  const float hX = L.hX;
  // flatten into B: candidate order = cell-row-major, then (y, x) inside a cell
  for (int i = tid; i < total; i += T) {
    int lo = 0, hi = nc;  // largest c with cellpref[c] <= i
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cellpref[mid] <= i) lo = mid; else hi = mid;
    }
    st.bufB[i] = candL[(size_t)lo * L.cand_cap_cell + (i - cellpref[lo])];
  }
  __syncthreads();
  // B -> A grouped by root, order kept: root = int(x / hX) (extractor.cc:571-574)
  team_prefix<NW>(total, ts.pref, ts.wtot, [&](int i) { return (int)__fdiv_rn((float)(st.bufB[i] & 0xFFF), hX); });
  int rc[4], rbase[4], rpos[4];
  {
    const uint2 tot = ts.pref[total];
    int acc = 0, np = 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      rc[rr] = rr < nroots ? pref_field(tot, rr) : 0;
      rbase[rr] = acc;
      rpos[rr] = np;
      acc += rc[rr];
      np += rc[rr] > 0 ? 1 : 0;
    }
  }
  for (int i = tid; i < total; i += T) {
    const uint32_t key = st.bufB[i];
    const int r = (int)__fdiv_rn((float)(key & 0xFFF), hX);
    const int pos = (r == 0 ? rbase[0] : r == 1 ? rbase[1] : r == 2 ? rbase[2] : rbase[3]) + pref_field(ts.pref[i], r);
    st.bufA[pos] = key;
    ts.ownCur[pos] = (unsigned short)(r == 0 ? rpos[0] : r == 1 ? rpos[1] : r == 2 ? rpos[2] : rpos[3]);
  }
  // root nodes (extractor.cc:558-586): empty ones are dropped
  int n = 0;
  {
    const int hgt = L.h - 2 * ORBX_DET_ORG;  // maxBorderY - minBorderY
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (rc[rr] > 0) {
        if (tid == 0) {
          QNode nd;
          nd.ulx = (short)(int)__fmul_rn(hX, (float)rr);
          nd.brx = (short)(int)__fmul_rn(hX, (float)(rr + 1));
          nd.uly = 0;
          nd.bry = (short)hgt;
          nd.start = rbase[rr];
          nd.cnt = rc[rr];  // buffer A
          st.cur[n] = nd;
        }
        ++n;
      }
    }
  }
  __syncthreads();

  const int N = L.quota;
  int nrec = 0;
  bool finish = false, finalRounds = false;
  // passes as in octree_level (extractor.cc:606-657, then :665-721)
  while (!finish) {
    const int prevSize = n;
    int m = 0;
    if (!finalRounds) {
      if (w == 0) {
        for (int base = 0; base < n; base += 64) {
          const int p = base + lane;
          const bool exp = p < n && qcount(st.cur[p]) > 1;
          const unsigned long long mask = __ballot(exp);
          const int t = m + mbcnt64(mask);
          if (exp) st.ord[t] = (unsigned short)p;
          if (p < n) ts.tOf[p] = exp ? (short)t : (short)-1;
          m += __popcll(mask);
        }
        if (lane == 0) ts.bc[0] = m;
      }
    } else {
      oct_sort_team<NW>(rec, nrec, st.ord, (unsigned short*)st.cbase, (unsigned long long*)st.newch,
                        (orbx_sort::elem_t*)st.cnt4, (int*)st.nxt);
      for (int p = tid; p < n; p += T) ts.tOf[p] = -1;
      __syncthreads();
      for (int t = tid; t < nrec; t += T) {
        const unsigned short p = (unsigned short)(uint32_t)rec[nrec - 1 - t];
        st.ord[t] = p;
        ts.tOf[p] = (short)t;
      }
      m = nrec;
    }
    __syncthreads();
    if (!finalRounds) m = ts.bc[0];
    // keys per quadrant before every key, over the keys of the nodes to split
    team_prefix<NW>(total, ts.pref, ts.wtot, [&](int i) {
      const int p = ts.ownCur[i];
      if (p == (int)OWN_NONE) return -1;
      if (ts.tOf[p] < 0) return -1;
      const QNode nd = st.cur[p];
      const uint32_t key = (((unsigned)nd.cnt >> 31) ? st.bufB : st.bufA)[i];
      return quadrant_of(key, nd.ulx + ((nd.brx - nd.ulx + 1) >> 1), nd.uly + ((nd.bry - nd.uly + 1) >> 1));
    });
    for (int t = tid; t < m; t += T) {
      const QNode nd = st.cur[st.ord[t]];
      const uint2 a = ts.pref[nd.start], b = ts.pref[nd.start + qcount(nd)];
      const uint32_t dl = b.x - a.x, dh = b.y - a.y;
      st.cnt4[t] = make_int4((int)(dl & 0xFFFFu), (int)(dl >> 16), (int)(dh & 0xFFFFu), (int)(dh >> 16));
    }
    __syncthreads();
    if (w == 0) {
      int nch0, nExp0;
      const int mEff0 = oct_place(st, m, n, N, finalRounds, &nch0, &nExp0);
      if (lane == 0) { ts.bc[1] = mEff0; ts.bc[2] = nch0; ts.bc[3] = nExp0; }
    }
    __syncthreads();
    const int mEff = ts.bc[1], nch = ts.bc[2], nToExpand = ts.bc[3];
    // every key of a node that is split moves to its child's range in the other buffer
    for (int i = tid; i < total; i += T) {
      const int p = ts.ownCur[i];
      unsigned short nown = (unsigned short)OWN_NONE;
      int pos = i;
      if (p != (int)OWN_NONE) {
        const int t = ts.tOf[p];
        if (t >= 0 && t < mEff) {
          const QNode nd = st.cur[p];
          const int buf = (int)((unsigned)nd.cnt >> 31);
          const uint32_t key = (buf ? st.bufB : st.bufA)[i];
          const int q = quadrant_of(key, nd.ulx + ((nd.brx - nd.ulx + 1) >> 1), nd.uly + ((nd.bry - nd.uly + 1) >> 1));
          const int4 c = st.cnt4[t];
          const int off = q == 0 ? 0 : q == 1 ? c.x : q == 2 ? c.x + c.y : c.x + c.y + c.z;
          pos = nd.start + off + pref_field(ts.pref[i], q) - pref_field(ts.pref[nd.start], q);
          (buf ? st.bufA : st.bufB)[pos] = key;
          const int child = st.cbase[t] + (q > 0 && c.x > 0) + (q > 1 && c.y > 0) + (q > 2 && c.z > 0);
          nown = (unsigned short)(nch - 1 - child);   // the child's place in the rebuilt list
        }
      }
      ts.ownNxt[pos] = nown;
    }
    __syncthreads();
    for (int t = tid; t < mEff; t += T) {
      const int p = st.ord[t];
      const QNode nd = st.cur[p];
      oct_write_children(st, nd, st.cnt4[t], st.cbase[t], nd.ulx + ((nd.brx - nd.ulx + 1) >> 1),
                         nd.uly + ((nd.bry - nd.uly + 1) >> 1));
      st.cur[p].cnt = 0;
    }
    __syncthreads();
    if (w == 0) {
      const int n0 = rebuild_list(st.cur, n, st.newch, nch, st.nxt);
      const int nrec0 = build_records(st.newch, nch, rec);
      if (lane == 0) { ts.bc[4] = n0; ts.bc[5] = nrec0; }
    }
    { QNode* t = st.cur; st.cur = st.nxt; st.nxt = t; }
    { unsigned short* t = ts.ownCur; ts.ownCur = ts.ownNxt; ts.ownNxt = t; }
    __syncthreads();
    n = ts.bc[4];
    nrec = ts.bc[5];
    if (n >= N || n == prevSize) finish = true;
    else if (n + nToExpand * 3 > N) finalRounds = true;
  }

  // best key per node: highest score, first one wins ties (extractor.cc:726-742)
  for (int p = tid; p < n; p += T) {
    const QNode nd = st.cur[p];
    const uint32_t* keys = (((unsigned)nd.cnt >> 31) ? st.bufB : st.bufA) + nd.start;
    const int cnt = qcount(nd);
    uint32_t best = keys[0];
    for (int k = 1; k < cnt; ++k) {
      const uint32_t kk = keys[k];
      if ((kk >> 24) > (best >> 24)) best = kk;
    }
    if (p < L.sel_cap) out[p] = best;
  }
  if (tid == 0) *selCountSlot = n < L.sel_cap ? n : L.sel_cap;
}

// LDS of the team kernel: the node arrays and cell prefix of k_octree, then the team's own arrays
// for plan->oct_team_keys keys.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_octree_team(const Plan* __restrict__ plan,
                                                         const uint32_t* __restrict__ cand,
                                                         const int* __restrict__ cellCount,
                                                         uint32_t* __restrict__ flatA,
                                                         uint32_t* __restrict__ flatB,
                                                         uint32_t* __restrict__ sel,
                                                         int* __restrict__ selCount, int f0, int l0) {
  extern __shared__ __align__(16) uint8_t smemDyn[];
  const int l = (int)blockIdx.y + l0, f = (int)blockIdx.x + f0;
  const int lane = lane_id(), w = (int)threadIdx.y;
  const LevelDesc& L = plan->lv[l];
  const int cap = plan->oct_nodes_cap, tk = plan->oct_team_keys;
  OctState st;
  st.cnt4 = (int4*)smemDyn;                                    // 16-byte aligned first
  st.cur = (QNode*)(st.cnt4 + cap);
  st.nxt = st.cur + cap;
  st.newch = st.nxt + cap;
  orbx_sort::elem_t* rec = (orbx_sort::elem_t*)(st.newch + cap);
  TeamState ts;
  ts.pref = (uint2*)(rec + cap);
  ts.wtot = ts.pref + (tk + 8);
  uint32_t* ldsKeys = (uint32_t*)(ts.wtot + 2 * NW);
  st.cbase = (int*)(ldsKeys + 2 * tk);
  int* cellpref = st.cbase + cap;
  st.ord = (unsigned short*)(cellpref + plan->oct_cells_max + 1);
  ts.tOf = (short*)(st.ord + cap);
  ts.ownCur = (unsigned short*)(ts.tOf + cap);
  ts.ownNxt = ts.ownCur + tk;
  __shared__ int rootCnt[ORBX_MAX_ROOTS];
  __shared__ int waveTot[NW];
  __shared__ int teamVals[8];
  ts.bc = teamVals;

  const int nc = L.ncx * L.ncy;
  const int* cc = cellCount + (size_t)f * plan->cells_per_frame + L.cell_first;
  // exclusive prefix of the cells' candidate counts: every wave scans its share, then the totals
  int total = 0;
  for (int base = 0; base < nc; base += 64 * NW) {
    const int c = base + w * 64 + lane;
    const int v = c < nc ? cc[c] : 0;
    const int inc = wave_incl_scan(v);
    if (lane == 63) waveTot[w] = inc;
    __syncthreads();
    int before = total;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      const int t = waveTot[ww];
      if (ww < w) before += t;
      total += t;
    }
    if (c < nc) cellpref[c] = before + inc - v;
    __syncthreads();
  }
  if (w == 0 && lane == 0) cellpref[nc] = total;
  if (w == 0 && lane < ORBX_MAX_ROOTS) rootCnt[lane] = 0;
  __syncthreads();
  const uint32_t* candL = cand + (size_t)f * plan->cand_frame + L.cand_off;
  uint32_t* out = sel + (size_t)f * plan->sel_frame + L.sel_off;
  int* selCountSlot = selCount + (size_t)f * plan->nlevels + l;
  if (total <= tk && L.n_roots <= 4) {
    st.bufA = ldsKeys;
    st.bufB = ldsKeys + tk;
    team_octree_level<NW>(L, st, ts, rec, cellpref, candL, nc, total, out, selCountSlot);
  } else if (w == 0) {
    // more keys than the team's buffers hold (or more roots than its counters): wave 0 alone,
    // keys in the global buffers
    st.bufA = flatA + (size_t)f * plan->flat_frame + L.flat_off;
    st.bufB = flatB + (size_t)f * plan->flat_frame + L.flat_off;
    octree_level(plan, L, st, rec, cellpref, rootCnt, candL, nc, total, out, selCountSlot);
  }
}

// ---------------------------------------------------------------------------
// K8: output slots.  One workgroup per frame (four waves, or sixteen for small batches).  Walks the selected keys in
// (level, list order), scales coordinates, applies the lapping test and fills
// the output from both ends exactly like extractor.cc:1052-1100.  Writes the
// keypoint records (angle is filled in by k_orient_describe) and the work list.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_assemble(const Plan* __restrict__ plan,
                                                  const uint32_t* __restrict__ sel,
                                                  const int* __restrict__ selCount, int lap0,
                                                  int lap1, float* __restrict__ kps,
                                                  WorkItem* __restrict__ work,
                                                  int* __restrict__ nOut, int* __restrict__ monoOut,
                                                  int f0) {
  __shared__ int lvOff[ORBX_MAX_LEVELS + 1];
  __shared__ int waveCnt[16];
  __shared__ int binCur[ORBX_MAX_LEVELS * 32];
  const int f = blockIdx.x + f0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = (int)blockDim.x, nwaves = nthreads >> 6;   // 4 or 16 waves
  const int nl = plan->nlevels;
  if (tid < 64) {   // the levels' counts in one round of loads (nl <= 64), then a wave scan
    const int c = tid < nl ? selCount[(size_t)f * nl + tid] : 0;
    const int inc = wave_incl_scan(c);
    if (tid < nl) lvOff[tid] = inc - c;
    if (tid == nl - 1) lvOff[nl] = inc;
  }
  __syncthreads();
  const int n = lvOff[nl];
  // The work list of the orientation/descriptor kernel is ordered by (level, band of rows), not
  // by output slot: keypoints that follow each other there then read overlapping image rows, and
  // most of their window lines are still in the vector cache.  Counting sort over 32 bands per
  // level; the order inside a band is whatever the atomics give (every item carries its slot).
  for (int b = tid; b < ORBX_MAX_LEVELS * 32; b += nthreads) binCur[b] = 0;
  __syncthreads();
  for (int k = tid; k < n; k += nthreads) {
    int l = 0;
    while (k >= lvOff[l + 1]) ++l;
    const uint32_t key = sel[(size_t)f * plan->sel_frame + plan->lv[l].sel_off + (k - lvOff[l])];
    const int y = (int)((key >> 12) & 0xFFF) + ORBX_DET_ORG;
    atomicAdd(&binCur[l * 32 + min(31, (y * 32) / plan->lv[l].h)], 1);
  }
  __syncthreads();
  if (tid < 64) {   // exclusive scan of the 512 bins by one wave, 8 per lane
    int v[8], t = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = binCur[tid * 8 + u]; t += v[u]; }
    int inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d);
      if (tid >= d) inc += o;
    }
    int run = inc - t;
#pragma unroll
    for (int u = 0; u < 8; ++u) { binCur[tid * 8 + u] = run; run += v[u]; }
  }
  __syncthreads();
  const float flap0 = (float)lap0, flap1 = (float)lap1;
  int stereoBefore = 0;  // stereo keys in earlier chunks
  float* kout = kps + (size_t)f * plan->kp_cap * 7;
  WorkItem* wout = work + (size_t)f * plan->kp_cap;
  for (int base = 0; base < n; base += nthreads) {
    const int k = base + tid;
    const bool valid = k < n;
    int l = 0;
    uint32_t key = 0;
    float x = 0.f, y = 0.f;
    bool stereo = false;
    if (valid) {
      while (k >= lvOff[l + 1]) ++l;
      key = sel[(size_t)f * plan->sel_frame + plan->lv[l].sel_off + (k - lvOff[l])];
      x = (float)((int)(key & 0xFFF) + ORBX_DET_ORG);
      y = (float)((int)((key >> 12) & 0xFFF) + ORBX_DET_ORG);
      if (l != 0) {
        const float s = plan->lv[l].scale;
        x = __fmul_rn(x, s);
        y = __fmul_rn(y, s);
      }
      stereo = x >= flap0 && x <= flap1;
    }
    const unsigned long long mask = __ballot(stereo);
    if (lane == 0) waveCnt[wave] = __popcll(mask);
    __syncthreads();
    int before = stereoBefore + mbcnt64(mask);
    int chunkTotal = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int c = w < nwaves ? waveCnt[w] : 0;
      if (w < wave) before += c;
      chunkTotal += c;
    }
    if (valid) {
      // stereo keys go to n-1, n-2, ...; the others to 0, 1, ...
      const int dst = stereo ? (n - 1 - before) : (k - before);
      float* o = kout + (size_t)dst * 7;
      o[0] = x;
      o[1] = y;
      o[2] = (float)plan->lv[l].patch_size;
      o[3] = -1.0f;
      o[4] = (float)(key >> 24);
      ((int*)o)[5] = l;
      ((int*)o)[6] = -1;
      WorkItem wi;
      const LevelDesc& L = plan->lv[l];
      const int cx = (int)(key & 0xFFF) + ORBX_DET_ORG, cy = (int)((key >> 12) & 0xFFF) + ORBX_DET_ORG;
      if (plan->blur_in_describe) {
        // one 43-row window of the un-blurred plane: row cy - 21, the dword that holds column cx - 21
        wi.pyr_win = L.pyr_off + (size_t)f * L.pyr_plane + (size_t)(cy + ORBX_EDGE - 21) * L.pstride +
                     (size_t)((cx + ORBX_EDGE - 21) & ~3);
        wi.blur_win = 0;
      } else {
        wi.pyr_win = L.pyr_off + (size_t)f * L.pyr_plane + (size_t)(cy + ORBX_EDGE - 15) * L.pstride +
                     (size_t)((cx + ORBX_EDGE - 15) & ~3);
        wi.blur_win = L.blur_off + (size_t)f * L.blur_plane + (size_t)(cy - 18) * L.bstride + (size_t)((cx - 18) & ~3);
      }
      wi.xy = (uint32_t)cx | ((uint32_t)cy << 16);
      wi.dst = dst;
      wi.strides = (uint32_t)L.pstride | ((uint32_t)L.bstride << 16);
      wi.level = l;
      const int yl = (int)((key >> 12) & 0xFFF) + ORBX_DET_ORG;
      wout[atomicAdd(&binCur[l * 32 + min(31, (yl * 32) / plan->lv[l].h)], 1)] = wi;
    }
    stereoBefore += chunkTotal;
    __syncthreads();
  }
  if (tid == 0) {
    nOut[f] = n;
    monoOut[f] = n - stereoBefore;
  }
}

// ---------------------------------------------------------------------------
// K6: 7x7 Gaussian blur, sigma 2, OpenCV's 8.8 fixed-point kernel
// {18,34,48,56,48,34,18}: 16-bit row sums, (sum + 2^15) >> 16 after the
// column pass.  The source is the padded pyramid plane, whose own border is
// the REFLECT_101 extension the blur needs.
//
// A lane owns 4 adjacent columns and walks down a band of rows, two rows per
// step.  Per input row it reads 12 aligned bytes (the 10 it needs start at
// padded column x0 + 16, a multiple of 4) and forms the 4 row sums with
// v_alignbyte + two v_dot4_u32_u8 each.  Row sums of consecutive rows are kept
// as 16-bit pairs in a rolling window of 4 pairs per column, from which two
// output rows come out of eight v_dot2_u32_u16 (rounding constant fed in as
// the accumulator).  No LDS.  Block = 4 waves stacked vertically; a wave covers
// 256 columns x BLUR_BAND rows.
// ---------------------------------------------------------------------------
#define BLUR_BAND 64

// the 4 horizontal sums of one input row (12 bytes in d0..d2)
__device__ __forceinline__ void blur_hrow(const uint8_t* rowp, uint32_t (&h)[4]) {
  const uint32_t* rp = (const uint32_t*)rowp;
  const uint32_t d0 = rp[0], d1 = rp[1], d2 = rp[2];
  const uint32_t KLO = 18u | (34u << 8) | (48u << 16) | (56u << 24);
  const uint32_t KHI = 48u | (34u << 8) | (18u << 16);
  h[0] = __builtin_amdgcn_udot4(d0, KLO, __builtin_amdgcn_udot4(d1, KHI, 0u, false), false);
#pragma unroll
  for (int c = 1; c < 4; ++c) {
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, c);
    const uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, c);
    h[c] = __builtin_amdgcn_udot4(lo, KLO, __builtin_amdgcn_udot4(hi, KHI, 0u, false), false);
  }
}

__global__ __launch_bounds__(256) void k_blur(const Plan* __restrict__ plan,
                                              const uint8_t* __restrict__ pyr,
                                              uint8_t* __restrict__ blur, int f0, int ntiles,
                                              int nunits, int chunk) {
  // a workgroup is the four 64-row bands of a tile (256 threads) or one of them (64 threads)
  const int wpb = (int)blockDim.x >> 6;
  const int u = xcd_contiguous_unit(chunk);
  const int unit = wpb == 4 ? u : u >> 2;
  if (unit >= nunits) return;
  const int fr = unit / ntiles;
  const int f = fr + f0;
  const int tg = unit - fr * ntiles;
  int l = 0;
  while (l + 1 < plan->nlevels && tg >= plan->lv[l + 1].tile_first) ++l;
  const LevelDesc& L = plan->lv[l];
  const int t = tg - L.tile_first;
  const int tyi = t / L.tiles_x, txi = t - tyi * L.tiles_x;
  const int lane = threadIdx.x & 63, wave = wpb == 4 ? (int)(threadIdx.x >> 6) : (u & 3);
  const int x0 = txi * 256 + lane * 4;
  const int y0 = (tyi * 4 + wave) * BLUR_BAND;
  if (x0 >= L.w || y0 >= L.h) return;
  const int nout = min(BLUR_BAND, L.h - y0);
  // input row i of the band = level row y0 - 3 + i = padded row y0 + 16 + i
  const int ps = L.pstride, bs = L.bstride;
  const uint8_t* src = pyr + L.pyr_off + (size_t)f * L.pyr_plane + (size_t)(y0 + 16) * ps + (x0 + 16);
  uint8_t* dst = blur + L.blur_off + (size_t)f * L.blur_plane + (size_t)y0 * bs + x0;
  // output rows (2t, 2t+1) need input rows 2t .. 2t+7 = row pairs t .. t+3
  uint32_t Q[4][4];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    uint32_t ha[4], hb[4];
    blur_hrow(src + (size_t)(2 * m) * ps, ha);
    blur_hrow(src + (size_t)(2 * m + 1) * ps, hb);
#pragma unroll
    for (int c = 0; c < 4; ++c) Q[m][c] = ha[c] | (hb[c] << 16);
  }
  const uint32_t A0 = 18u | (34u << 16), A1 = 48u | (56u << 16), A2 = 48u | (34u << 16), A3 = 18u;
  const uint32_t B0 = 18u << 16, B1 = 34u | (48u << 16), B2 = 56u | (48u << 16), B3 = 34u | (18u << 16);
  for (int t0 = 0; 2 * t0 < nout; t0 += 4) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int tt = t0 + s;
      if (2 * tt < nout) {
        uint32_t ha[4], hb[4];
        blur_hrow(src + (size_t)(2 * tt + 6) * ps, ha);
        blur_hrow(src + (size_t)(2 * tt + 7) * ps, hb);
#pragma unroll
        for (int c = 0; c < 4; ++c) Q[(s + 3) & 3][c] = ha[c] | (hb[c] << 16);
        uint32_t ra[4], rb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t p0 = Q[s & 3][c], p1 = Q[(s + 1) & 3][c], p2 = Q[(s + 2) & 3][c], p3 = Q[(s + 3) & 3][c];
          ra[c] = udot2(p0, A0, udot2(p1, A1, udot2(p2, A2, udot2(p3, A3, 32768u))));
          rb[c] = udot2(p0, B0, udot2(p1, B1, udot2(p2, B2, udot2(p3, B3, 32768u))));
        }
        // result byte = bits 16..23 of each sum
        const uint32_t oa = __builtin_amdgcn_perm(ra[1], ra[0], 0x0c0c0602u) |
                            __builtin_amdgcn_perm(ra[3], ra[2], 0x06020c0cu);
        *(uint32_t*)(dst + (size_t)(2 * tt) * bs) = oa;
        if (2 * tt + 1 < nout) {
          const uint32_t ob = __builtin_amdgcn_perm(rb[1], rb[0], 0x0c0c0602u) |
                              __builtin_amdgcn_perm(rb[3], rb[2], 0x06020c0cu);
          *(uint32_t*)(dst + (size_t)(2 * tt + 1) * bs) = ob;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K5 + K7: orientation by intensity centroid, then the 256-bit rotated BRIEF
// descriptor.
// ---------------------------------------------------------------------------
__constant__ __attribute__((aligned(16))) int8_t c_pattern[256 * 4] = {
#include "orb_pattern_31.inc"
};

// the same coordinates as floats, for the kernel that rotates them (no conversions in its loop)
__constant__ __attribute__((aligned(16))) float c_pattern_f[256 * 4] = {
#include "orb_pattern_31.inc"
};

// cv::fastAtan2 (degrees).
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float k = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * k;
  const float p3 = -0.3258083974640975f * k;
  const float p5 = 0.1555786518463281f * k;
  const float p7 = -0.04432655554792128f * k;
  const float eps = 2.2204460492503131e-16f;  // (float)DBL_EPSILON
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

// sin and cos of x in [0, 7) in double precision (the keypoint angle in radians): quadrant
// reduction against pi/2 in two pieces (k * the 33-bit head is exact for k <= 4), then the
// classic minimax polynomials for |r| <= pi/4.  Error below one unit in the last place of a
// double, so the values rounded to float are those of a correctly rounded sin / cos except for
// arguments within ~2^-29 (relative) of a float rounding boundary - the same guarantee the
// math library's sincos gives, at a fifth of its instructions.
__device__ __forceinline__ void sincos_small(double x, double& s, double& c) {
  const double k = __builtin_rint(x * 6.36619772367581382433e-01);          // x * 2/pi
  double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);              // pi/2, first 33 bits
  r = __builtin_fma(-k, 6.07710050650619224932e-11, r);                     // pi/2, the next 53 bits
  const double z = r * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  const double sr = __builtin_fma(z * r, ps, r);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double cr = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
  const int n = (int)k & 3;
  const double s0 = (n & 1) ? cr : sr, c0 = (n & 1) ? sr : cr;
  s = (n & 2) ? -s0 : s0;
  c = ((n + 1) & 2) ? -c0 : c0;
}

// 16 lanes per keypoint, 4 keypoints per wave, one wave per workgroup: a one-wave workgroup with
// 7 KB of LDS finds a place on a CU that the detection kernel of the batch's other part keeps
// full of small workgroups; four-wave workgroups with 28 KB waited for most of that kernel's run.
//  * moments: lane j of a group owns patch columns u = 2j-15 and 2j-14 and walks the
//    31 rows of the patch, which is staged in LDS by row-wise 16-byte loads;
//  * descriptor: lane j evaluates tests 16j .. 16j+15 = descriptor bytes 2j, 2j+1.
// Both windows are staged with 16-byte loads, three per row (48 bytes cover the 36 / 40 bytes
// a row needs from its dword-aligned start), so a window row is 12 dwords in LDS.
#define WIN_ROW_DW 12
#define DESC_WIN_DW (37 * WIN_ROW_DW)   // 37 rows
#define DESC_CHUNKS (37 * 3)
#define DESC_STAGE 7                    // 16-byte chunks per lane (16 lanes per keypoint)
#define IC_CHUNKS (31 * 3)
#define IC_STAGE 6
__global__ __launch_bounds__(64) void k_orient_describe(const Plan* __restrict__ plan,
                                                         const uint8_t* __restrict__ pyr,
                                                         const uint8_t* __restrict__ blur,
                                                         const WorkItem* __restrict__ work,
                                                         const int* __restrict__ nOut,
                                                         float* __restrict__ kps,
                                                         uint8_t* __restrict__ desc, int f0,
                                                         int nblk, int nunits, int chunk) {
  const int unit = xcd_contiguous_unit(chunk);
  if (unit >= nunits) return;
  const int fr = unit / nblk;
  const int f = fr + f0;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, j16 = lane & 15;
  const int n = nOut[f];
  const int kbase = (unit - fr * nblk) * 4;
  if (kbase >= n) return;
  const int k = kbase + g;
  const bool valid = k < n;
  const WorkItem wi = work[(size_t)f * plan->kp_cap + (valid ? k : n - 1)];
  const int cx = wi.xy & 0xFFFF;
  const int ps = (int)(wi.strides & 0xFFFF), bs = (int)(wi.strides >> 16);

  // ---- IC_Angle on the un-blurred level (extractor.cc:75-99) ----------------
  // the 31 x 31 patch goes through the same LDS window as the blurred one below
  __shared__ __align__(16) uint32_t sWin[4][DESC_WIN_DW];
  const int slotInBlock = g;
  int m10 = 0, m01 = 0;
  {
    const int px0 = (cx + ORBX_EDGE - 15) & ~3;   // padded column of the window's first dword
    const uint8_t* pbase = pyr + wi.pyr_win;
    uint4 st2[IC_STAGE];
#pragma unroll
    for (int u = 0; u < IC_STAGE; ++u) {
      const int idx = min(j16 + 16 * u, IC_CHUNKS - 1);
      const int r = idx / 3, c3 = idx - r * 3;
      const U4A4 t = *(const U4A4*)(pbase + (size_t)r * ps + 16 * c3);
      st2[u] = make_uint4(t.x, t.y, t.z, t.w);
    }
#pragma unroll
    for (int u = 0; u < IC_STAGE; ++u) {
      const int idx = j16 + 16 * u;
      if (idx < IC_CHUNKS) *(uint4*)&sWin[slotInBlock][4 * idx] = st2[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // Lane j of the group sums patch rows v = 2j - 15 and 2j - 14 (the second one does not exist
    // for j = 15: its mask is empty).  A row's 31 pixels u = -15 .. 15 start at byte `sh` of the
    // window row; eight re-based dwords hold them (plus u = 16, always masked), the row's mask
    // keeps |u| <= umax[|v|], and two 4-way byte dot products per dword give sum(I) and
    // sum((u + 16) I).
    const int sh = (cx + ORBX_EDGE - px0) - 15;   // 0..3
    const unsigned long long umaxPacked = plan->umax_packed;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * j16 + rr;
      const int v = r - 15;
      const int av = v < 0 ? -v : v;
      // umax[|v|] from the packed table (4 bits per row; row 16 does not exist: nothing passes),
      // then per dword the bytes lo <= 4 i + k <= hi as a byte mask: two per-byte compares
      // through bit 7 (no borrow: all operands are below 0x80), widened to 0xFF
      const int lim = av < 16 ? (int)((umaxPacked >> (4 * av)) & 15u) : -1;
      const uint32_t lo4 = (uint32_t)(15 - lim) * 0x01010101u, hi4 = ((uint32_t)(15 + lim) * 0x01010101u) | 0x80808080u;
      uint32_t mk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t ci = 0x03020100u + 0x04040404u * (uint32_t)i;
        const uint32_t ok = ((ci | 0x80808080u) - lo4) & (hi4 - ci) & 0x80808080u;
        mk[i] = lim < 0 ? 0u : (ok << 1) - (ok >> 7);
      }
      const uint4* rp = (const uint4*)&sWin[slotInBlock][r * WIN_ROW_DW];
      const uint4 q0 = rp[0], q1 = rp[1], q2 = rp[2];
      const uint32_t d[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
      uint32_t sI = 0, sW = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t x = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh) & mk[i];
        const uint32_t wk = (uint32_t)(4 * i + 1) | ((uint32_t)(4 * i + 2) << 8) | ((uint32_t)(4 * i + 3) << 16) |
                            ((uint32_t)(4 * i + 4) << 24);
        sI = __builtin_amdgcn_udot4(x, 0x01010101u, sI, false);
        sW = __builtin_amdgcn_udot4(x, wk, sW, false);
      }
      m10 += (int)sW - 16 * (int)sI;
      m01 += v * (int)sI;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) {
    m10 += __shfl_xor(m10, d, 64);
    m01 += __shfl_xor(m01, d, 64);
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);

  // ---- steered BRIEF on the blurred level (extractor.cc:102-145) ------------
  // The 37 x 37 window the 512 sample points can fall into (reach 18) is staged in LDS
  // with row-wise 16-byte loads (issued before the trigonometry so they overlap it); the
  // rotated samples are then LDS byte reads instead of scattered global gathers.
  const int bx0 = (cx - 18) & ~3;
  const uint8_t* wbase = blur + wi.blur_win;
  uint4 stage[DESC_STAGE];
#pragma unroll
  for (int u = 0; u < DESC_STAGE; ++u) {
    const int idx = min(j16 + 16 * u, DESC_CHUNKS - 1);
    const int r = idx / 3, c3 = idx - r * 3;
    const U4A4 t = *(const U4A4*)(wbase + (size_t)r * bs + 16 * c3);
    stage[u] = make_uint4(t.x, t.y, t.z, t.w);
  }
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.0);
  const float rad = __fmul_rn(angle, factorPI);
  double sind, cosd;
  sincos((double)rad, &sind, &cosd);
  const float a = (float)cosd, b = (float)sind;
#pragma unroll
  for (int u = 0; u < DESC_STAGE; ++u) {
    const int idx = j16 + 16 * u;
    if (idx < DESC_CHUNKS) *(uint4*)&sWin[slotInBlock][4 * idx] = stage[u];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const uint8_t* bc = (const uint8_t*)sWin[slotInBlock] + 18 * (4 * WIN_ROW_DW) + (cx - bx0);
  const uint4* pp = (const uint4*)(c_pattern + 64 * j16);
  uint32_t bits = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 w4 = pp[q];
    const uint32_t ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t w = ws[e];   // x0, y0, x1, y1 of test 16 j + 4 q + e
      const float x0 = (float)(int8_t)(w & 0xFF), y0 = (float)(int8_t)((w >> 8) & 0xFF);
      const float x1 = (float)(int8_t)((w >> 16) & 0xFF), y1 = (float)(int8_t)(w >> 24);
      // cvRound of the rotated coordinates (round half to even), then row * 48 + column; both are
      // small integers held in floats, so the fused multiply-add is exact
      const float r0 = rintf(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
      const float q0 = rintf(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
      const float r1 = rintf(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
      const float q1 = rintf(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
      const int t0 = bc[(int)__fmaf_rn(r0, (float)(4 * WIN_ROW_DW), q0)];
      const int t1 = bc[(int)__fmaf_rn(r1, (float)(4 * WIN_ROW_DW), q1)];
      bits |= (t0 < t1 ? 1u : 0u) << (q * 4 + e);
    }
  }
  if (valid) {
    const size_t slot = (size_t)f * plan->kp_cap + wi.dst;
    *(uint16_t*)(desc + slot * 32 + 2 * j16) = (uint16_t)bits;
    if (j16 == 0) kps[slot * 7 + 3] = angle;
  }
}

// ---------------------------------------------------------------------------
// K5 + K6 + K7 in one kernel: orientation, the 7x7 blur of just the keypoint's sample window, and
// the descriptor.  The descriptor only ever reads blurred pixels within 18 of its keypoint
// (extractor.cc:110-112: pattern coordinates up to 13, rotated), and a blurred pixel is a function
// of the 7 x 7 un-blurred pixels around it, so one 43 x 43 window of the un-blurred level holds
// everything a keypoint needs: the 31 x 31 orientation patch and the support of the 37 x 37
// blurred window.  The blurred planes then need not exist at all: no blur launch, no blurred
// bytes written to or read from memory.  The arithmetic is the full-plane blur's (8.8 fixed-point
// kernel {18,34,48,56,48,34,18}, exact 16-bit row sums, (sum + 2^15) >> 16 after the column
// pass), applied to the padded plane, whose border is the blur's REFLECT_101 extension.
//
// 32 lanes per keypoint, 2 keypoints per one-wave workgroup.  Per keypoint in LDS:
//   raw  43 rows x 12 dwords: window rows cy-21 .. cy+21, columns from the dword that holds cx-21;
//   blw  38 rows x 10 dwords: the blurred window, rows cy-18 .. cy+19 (the last one a by-product),
//        columns from window byte 3.
// Passes: stage -> moments (lane j: patch row j - 15) -> blur (30 tasks of 4 columns x a band
// of 12 or 13 rows: a lane walks down its band, forms the row sums of its 4 columns from three
// window dwords per row and keeps them as 16-bit pairs of consecutive rows in a rolling window
// of registers, from which two blurred rows come out of eight v_dot2 per column) -> 256 tests,
// 8 per lane = one descriptor byte per lane.
// ---------------------------------------------------------------------------
#define FW_ROWS 43
#define FW_RAW_DW (FW_ROWS * WIN_ROW_DW)      // 516 dwords
#define FW_B_ROW_DW 10
#define FW_B_DW (38 * FW_B_ROW_DW)            // 380 dwords
#define FW_KP_DW (FW_RAW_DW + FW_B_DW)        // 896 dwords = 3584 bytes per keypoint
#define FW_CHUNKS (FW_ROWS * 3)               // 16-byte chunks of the raw window
#define FW_STAGE 5                            // chunks per lane (32 lanes per keypoint)

// the 4 horizontal sums of one window row from its dwords g, g+1, g+2 (LDS): column c sums window
// bytes c .. c + 6 with the weights {18,34,48,56,48,34,18}; instead of re-basing the bytes for
// every column the weights are laid where the bytes are (zero elsewhere): 10 byte dot products
__device__ __forceinline__ void blur_hrow_lds(const uint32_t* rp, uint32_t (&h)[4]) {
  const uint32_t d0 = rp[0], d1 = rp[1], d2 = rp[2];
  const uint32_t w0 = 18u, w1 = 34u, w2 = 48u, w3 = 56u;
  const uint32_t K0a = w0 | (w1 << 8) | (w2 << 16) | (w3 << 24), K0b = w2 | (w1 << 8) | (w0 << 16);
  const uint32_t K1a = (w0 << 8) | (w1 << 16) | (w2 << 24), K1b = w3 | (w2 << 8) | (w1 << 16) | (w0 << 24);
  const uint32_t K2a = (w0 << 16) | (w1 << 24), K2b = w2 | (w3 << 8) | (w2 << 16) | (w1 << 24), K2c = w0;
  const uint32_t K3a = w0 << 24, K3b = w1 | (w2 << 8) | (w3 << 16) | (w2 << 24), K3c = w1 | (w0 << 8);
  h[0] = __builtin_amdgcn_udot4(d0, K0a, __builtin_amdgcn_udot4(d1, K0b, 0u, false), false);
  h[1] = __builtin_amdgcn_udot4(d0, K1a, __builtin_amdgcn_udot4(d1, K1b, 0u, false), false);
  h[2] = __builtin_amdgcn_udot4(d0, K2a, __builtin_amdgcn_udot4(d1, K2b, __builtin_amdgcn_udot4(d2, K2c, 0u, false), false), false);
  h[3] = __builtin_amdgcn_udot4(d0, K3a, __builtin_amdgcn_udot4(d1, K3b, __builtin_amdgcn_udot4(d2, K3c, 0u, false), false), false);
}

__global__ __launch_bounds__(64) void k_orient_blur_describe(const Plan* __restrict__ plan,
                                                              const uint8_t* __restrict__ pyr,
                                                              const WorkItem* __restrict__ work,
                                                              const int* __restrict__ nOut,
                                                              float* __restrict__ kps,
                                                              uint8_t* __restrict__ desc, int f0,
                                                              int nblk, int nunits, int chunk) {
  const int unit = xcd_contiguous_unit(chunk);
  if (unit >= nunits) return;
  const int fr = unit / nblk;
  const int f = fr + f0;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 5, j32 = lane & 31;
  const int n = nOut[f];
  const int kbase = (unit - fr * nblk) * 2;
  if (kbase >= n) return;
  const int k = kbase + g;
  const bool valid = k < n;
  const WorkItem wi = work[(size_t)f * plan->kp_cap + (valid ? k : n - 1)];
  const int cx = wi.xy & 0xFFFF;
  const int ps = (int)(wi.strides & 0xFFFF);
  __shared__ __align__(16) uint32_t sWin[2][FW_KP_DW];
  uint32_t* raw = sWin[g];
  uint32_t* blw = raw + FW_RAW_DW;
  // byte of the window row at which padded column cx - 21 + EDGE sits (the window starts at the
  // dword that holds it)
  const int off = (cx + ORBX_EDGE - 21) & 3;

  // ---- stage the un-blurred window ----------------------------------------------
  {
    const uint8_t* pbase = pyr + wi.pyr_win;
    uint4 st[FW_STAGE];
#pragma unroll
    for (int u = 0; u < FW_STAGE; ++u) {
      const int idx = min(j32 + 32 * u, FW_CHUNKS - 1);
      const int r = idx / 3, c3 = idx - r * 3;
      const U4A4 t = *(const U4A4*)(pbase + (size_t)r * ps + 16 * c3);
      st[u] = make_uint4(t.x, t.y, t.z, t.w);
    }
#pragma unroll
    for (int u = 0; u < FW_STAGE; ++u) {
      const int idx = j32 + 32 * u;
      if (idx < FW_CHUNKS) *(uint4*)&raw[4 * idx] = st[u];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();

  // ---- IC_Angle on the un-blurred level (extractor.cc:75-99) ----------------------
  // Lane j sums patch row v = j - 15 = window row j + 6 (lane 31 has none: its mask is empty).
  // A row's 31 pixels u = -15 .. 15 start at window byte off + 6; eight re-based dwords hold
  // them (plus u = 16, always masked), the row's mask keeps |u| <= umax[|v|], and two 4-way byte
  // dot products per dword give sum(I) and sum((u + 16) I).
  int m10, m01;
  {
    const int s0 = off + 6, dw0 = s0 >> 2, sh = s0 & 3;
    const int v = j32 - 15;
    // byte masks of the lane's patch row (|u| <= umax[|v|]; all zero for lane 31, which has no row)
    const uint4 mka = *(const uint4*)&plan->ic_mask[j32][0], mkb = *(const uint4*)&plan->ic_mask[j32][4];
    const uint32_t mk[8] = {mka.x, mka.y, mka.z, mka.w, mkb.x, mkb.y, mkb.z, mkb.w};
    const uint32_t* rp = raw + min(j32 + 6, FW_ROWS - 1) * WIN_ROW_DW + dw0;
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = rp[i];
    uint32_t sI = 0, sW = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t x = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh) & mk[i];
      const uint32_t wk = (uint32_t)(4 * i + 1) | ((uint32_t)(4 * i + 2) << 8) | ((uint32_t)(4 * i + 3) << 16) |
                          ((uint32_t)(4 * i + 4) << 24);
      sI = __builtin_amdgcn_udot4(x, 0x01010101u, sI, false);
      sW = __builtin_amdgcn_udot4(x, wk, sW, false);
    }
    m10 = (int)sW - 16 * (int)sI;
    m01 = v * (int)sI;
  }
  // sums over the keypoint's 32 lanes: prefix sums inside the rows of 16 lanes (no LDS round
  // trips), then the two row totals of the keypoint's half of the wave
  {
    int s10 = m10, s01 = m01;
    s10 += __builtin_amdgcn_update_dpp(0, s10, 0x111, 0xf, 0xf, true);
    s01 += __builtin_amdgcn_update_dpp(0, s01, 0x111, 0xf, 0xf, true);
    s10 += __builtin_amdgcn_update_dpp(0, s10, 0x112, 0xf, 0xf, true);
    s01 += __builtin_amdgcn_update_dpp(0, s01, 0x112, 0xf, 0xf, true);
    s10 += __builtin_amdgcn_update_dpp(0, s10, 0x114, 0xf, 0xf, true);
    s01 += __builtin_amdgcn_update_dpp(0, s01, 0x114, 0xf, 0xf, true);
    s10 += __builtin_amdgcn_update_dpp(0, s10, 0x118, 0xf, 0xf, true);
    s01 += __builtin_amdgcn_update_dpp(0, s01, 0x118, 0xf, 0xf, true);
    const int a10 = __builtin_amdgcn_readlane(s10, 15) + __builtin_amdgcn_readlane(s10, 31);
    const int b10 = __builtin_amdgcn_readlane(s10, 47) + __builtin_amdgcn_readlane(s10, 63);
    const int a01 = __builtin_amdgcn_readlane(s01, 15) + __builtin_amdgcn_readlane(s01, 31);
    const int b01 = __builtin_amdgcn_readlane(s01, 47) + __builtin_amdgcn_readlane(s01, 63);
    m10 = g ? b10 : a10;
    m01 = g ? b01 : a01;
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);

  // ---- blur of the window: blurred row rb (= window row rb + 3) from window rows rb .. rb + 6 ----
  if (j32 < 30) {
    const int band = j32 >= 20 ? 2 : (j32 >= 10 ? 1 : 0);
    const int gq = j32 - 10 * band;                       // dword group: blurred columns 4 gq .. 4 gq + 3
    const int rb0 = band == 0 ? 0 : (band == 1 ? 13 : 25);
    const int nout = band == 0 ? 13 : 12;
    const uint32_t* src = raw + rb0 * WIN_ROW_DW + gq;     // window row rb0, the group's first dword
    uint32_t* dst = blw + rb0 * FW_B_ROW_DW + gq;
    // output rows (2t, 2t+1) of the band need its input rows 2t .. 2t+7 = row pairs t .. t+3
    uint32_t Q[4][4];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      uint32_t ha[4], hb[4];
      blur_hrow_lds(src + (2 * m) * WIN_ROW_DW, ha);
      blur_hrow_lds(src + (2 * m + 1) * WIN_ROW_DW, hb);
#pragma unroll
      for (int c = 0; c < 4; ++c) Q[m][c] = ha[c] | (hb[c] << 16);
    }
    const uint32_t A0 = 18u | (34u << 16), A1 = 48u | (56u << 16), A2 = 48u | (34u << 16), A3 = 18u;
    const uint32_t B0 = 18u << 16, B1 = 34u | (48u << 16), B2 = 56u | (48u << 16), B3 = 34u | (18u << 16);
#pragma unroll
    for (int tt = 0; tt < 7; ++tt) {
      const int s = tt & 3;
      if (2 * tt < nout) {
        uint32_t ha[4], hb[4];
        blur_hrow_lds(src + (2 * tt + 6) * WIN_ROW_DW, ha);
        // the band's last odd input row lies past the window for the lowest band's last pair: its
        // sums only feed the row below the window's last blurred row
        blur_hrow_lds(src + min(2 * tt + 7, FW_ROWS - 1 - rb0) * WIN_ROW_DW, hb);
#pragma unroll
        for (int c = 0; c < 4; ++c) Q[(s + 3) & 3][c] = ha[c] | (hb[c] << 16);
        uint32_t ra[4], rb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t p0 = Q[s & 3][c], p1 = Q[(s + 1) & 3][c], p2 = Q[(s + 2) & 3][c], p3 = Q[(s + 3) & 3][c];
          ra[c] = udot2(p0, A0, udot2(p1, A1, udot2(p2, A2, udot2(p3, A3, 32768u))));
          rb[c] = udot2(p0, B0, udot2(p1, B1, udot2(p2, B2, udot2(p3, B3, 32768u))));
        }
        // result byte = bits 16..23 of each sum
        dst[(2 * tt) * FW_B_ROW_DW] = __builtin_amdgcn_perm(ra[1], ra[0], 0x0c0c0602u) |
                                      __builtin_amdgcn_perm(ra[3], ra[2], 0x06020c0cu);
        if (2 * tt + 1 < nout)
          dst[(2 * tt + 1) * FW_B_ROW_DW] = __builtin_amdgcn_perm(rb[1], rb[0], 0x0c0c0602u) |
                                            __builtin_amdgcn_perm(rb[3], rb[2], 0x06020c0cu);
      }
    }
  }
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.0);
  const float rad = __fmul_rn(angle, factorPI);
  double sind, cosd;
  sincos_small((double)rad, sind, cosd);
  const float a = (float)cosd, b = (float)sind;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();

  // ---- steered BRIEF on the blurred window (extractor.cc:102-145) ---------------------
  // blurred window (row, column) = level (cy - 18 + row, cx - 18 - off + column); lane j
  // evaluates tests 8 j .. 8 j + 7 = descriptor byte j
  // cvRound of a rotated coordinate (round half to even): adding 1.5 * 2^23 rounds the sum to an
  // integer the same way and leaves that integer in the low mantissa bits, 0x400000 + n in the low
  // 24, so row * 40 + column comes out of one 24-bit multiply-add on the raw bits; what the
  // biases add up to is taken off the base address once
  const float RND = 12582912.0f;
  const uint32_t RND_BITS = 0x4B400000u;
  const uint32_t BIAS = 0x400000u * (uint32_t)(4 * FW_B_ROW_DW) + RND_BITS;
  const uint8_t* bc = (const uint8_t*)blw + 18 * (4 * FW_B_ROW_DW) + (off + 18);
  const float4* pp = (const float4*)(c_pattern_f + 32 * j32);
  uint32_t bits = 0;
#pragma unroll
  for (int e = 7; e >= 0; --e) {   // highest bit first: each test shifts the earlier ones up
    const float4 w = pp[e];   // x0, y0, x1, y1 of test 8 j + e
    const float r0 = __fadd_rn(__fadd_rn(__fmul_rn(w.x, b), __fmul_rn(w.y, a)), RND);
    const float q0 = __fadd_rn(__fsub_rn(__fmul_rn(w.x, a), __fmul_rn(w.y, b)), RND);
    const float r1 = __fadd_rn(__fadd_rn(__fmul_rn(w.z, b), __fmul_rn(w.w, a)), RND);
    const float q1 = __fadd_rn(__fsub_rn(__fmul_rn(w.z, a), __fmul_rn(w.w, b)), RND);
    const uint32_t i0 = __umul24(__float_as_uint(r0), (uint32_t)(4 * FW_B_ROW_DW)) + __float_as_uint(q0) - BIAS;
    const uint32_t i1 = __umul24(__float_as_uint(r1), (uint32_t)(4 * FW_B_ROW_DW)) + __float_as_uint(q1) - BIAS;
    const int t0 = bc[(int)i0];
    const int t1 = bc[(int)i1];
    // bits = bits << 1 | (t0 < t1): the sign of t0 - t1 moves in from below
    bits = __builtin_amdgcn_alignbit(bits, (uint32_t)(t0 - t1), 31);
  }
  if (valid) {
    const size_t slot = (size_t)f * plan->kp_cap + wi.dst;
    desc[slot * 32 + j32] = (uint8_t)bits;
    if (j32 == 0) kps[slot * 7 + 3] = angle;
  }
}

// ---------------------------------------------------------------------------
// Results of the first frames of a batch (one per blockIdx.y) into page-locked host memory the
// GPU writes directly: the two counts (hcnt[f], hcnt[G + f]), then the frame's n keypoint records
// (7 words each) and n descriptors (8 words each) at its slot of kc records.  One launch behind
// the descriptor kernel instead of copies handed to a copy engine.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_export_few(const int* __restrict__ d_n, const int* __restrict__ d_mono,
                                                    const uint32_t* __restrict__ kps,
                                                    const uint32_t* __restrict__ desc, int kc, int cap, int G,
                                                    int* __restrict__ hcnt, uint32_t* __restrict__ hk,
                                                    uint32_t* __restrict__ hd) {
  const int f = (int)blockIdx.y;
  const int nAll = d_n[f];
  if (blockIdx.x == 0 && threadIdx.x == 0) { hcnt[f] = nAll; hcnt[G + f] = d_mono[f]; }
  const int n = nAll < cap ? nAll : cap;
  const int nk = n * 7, nw = n * 15;
  const uint32_t* ksrc = kps + (size_t)f * kc * 7;
  const uint32_t* dsrc = desc + (size_t)f * kc * 8;
  uint32_t* kdst = hk + (size_t)f * kc * 7;
  uint32_t* ddst = hd + (size_t)f * kc * 8;
  for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < nw; i += (int)gridDim.x * 256) {
    if (i < nk) kdst[i] = ksrc[i];
    else ddst[i - nk] = dsrc[i - nk];
  }
}

// ---------------------------------------------------------------------------
// launch functions (host): frames [f0, f0 + nframes) of the batch on stream s
// ---------------------------------------------------------------------------
void launch_pyramid(const Plan& hp, const uint8_t* d_img, int istride, size_t iframe, uint8_t* d_pyr,
                    int f0, int nframes, hipStream_t s) {
  // waves per workgroup (1, 2 or 4): alone on the chip the stage is cheapest with fat workgroups
  // (fewer dispatches); beside the other part's detection kernel, which keeps the CUs full of
  // one-wave workgroups, a one-wave workgroup is placed as soon as a single wave slot frees
  const int wpb = hp.pyr_waves;
  const dim3 blk(64, wpb);
  {
    const LevelDesc& L = hp.lv[0];
    const int nbx = (L.pstride / 16 + 63) / 64;
    const int gx = nbx + 1, gy = (L.prows + wpb * L0_ROWS - 1) / (wpb * L0_ROWS);
    const int nblocks = gx * gy * nframes, chunk = hp.pyr_xcd ? (nblocks + 7) / 8 : 0;
    hipLaunchKernelGGL(k_pyr_level0, dim3(chunk ? 8 * chunk : nblocks), blk, 0, s, d_img, hp.W, hp.H, istride, iframe,
                       d_pyr + L.pyr_off, L.pstride, L.prows, L.pyr_plane, nbx, f0, gx, gy, nblocks, chunk);
  }
  for (int l = 1; l < hp.nlevels; ++l) {
    const LevelDesc& S = hp.lv[l - 1];
    const LevelDesc& L = hp.lv[l];
    const int nbx = (L.pstride + 255) / 256;
    // rows per wave: as many 4-row steps (up to 4) as still leave the chip about two rounds of waves
    int groups = 4;
    while (groups > 1 && (long long)nbx * ((L.prows + 4 * groups - 1) / (4 * groups)) * nframes < 16384) groups >>= 1;
    const int gx = nbx + (L.n_bcols > 0 ? 1 : 0), gy = (L.prows + 4 * wpb * groups - 1) / (4 * wpb * groups);
    const int nblocks = gx * gy * nframes, chunk = hp.pyr_xcd ? (nblocks + 7) / 8 : 0;
    auto kern = groups == 4 ? k_pyr_resize<4> : groups == 2 ? k_pyr_resize<2> : k_pyr_resize<1>;
    hipLaunchKernelGGL(kern, dim3(chunk ? 8 * chunk : nblocks), blk, 0, s, d_pyr + S.pyr_off,
                       S.pstride, S.pyr_plane, d_pyr + L.pyr_off, L.pstride, L.prows, L.pyr_plane,
                       L.w + 2 * ORBX_EDGE, L.xtab, L.ytab, L.colrec, L.bcols,
                       L.n_bcols, nbx, f0, gx, gy, nblocks, chunk);
  }
}

size_t fast_lds_bytes(const Plan& hp, int ts) {
  // tile + 6-bit tile (the map of m reuses the latter: fast_mm_rows <= fast_tile_rows) + queue + two masks +
  // per-word corner counts + the queue counter
  return (size_t)(2 * hp.fast_tile_rows) * ts + (size_t)hp.fast_qcap * 2 + (size_t)hp.fast_words * 10 + 8 + 16;
}

// cells of levels [l0, l1)
void launch_fast(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, uint32_t* d_cand,
                 int* d_cellCount, int l0, int l1, int f0, int nframes, hipStream_t s) {
  const int cell0 = hp.lv[l0].cell_first;
  const int cell1 = l1 < hp.nlevels ? hp.lv[l1].cell_first : hp.cells_per_frame;
  if (cell1 <= cell0) return;
  const int ncells = cell1 - cell0, nunits = ncells * nframes, chunk = (nunits + 7) / 8;
  dim3 grid(8 * chunk);
  const int ts = hp.fast_ts;
  const size_t lds = fast_lds_bytes(hp, ts);
  if (ts == 48)
    hipLaunchKernelGGL(k_fast_cells<48>, grid, dim3(64), lds, s, dp, hp.fast_cells, d_pyr, d_cand, d_cellCount, f0, cell0,
                       ncells, nunits, chunk);
  else if (ts == 64)
    hipLaunchKernelGGL(k_fast_cells<64>, grid, dim3(64), lds, s, dp, hp.fast_cells, d_pyr, d_cand, d_cellCount, f0, cell0,
                       ncells, nunits, chunk);
  else
    hipLaunchKernelGGL(k_fast_cells<80>, grid, dim3(64), lds, s, dp, hp.fast_cells, d_pyr, d_cand, d_cellCount, f0, cell0,
                       ncells, nunits, chunk);
}

size_t octree_lds_bytes(const Plan& hp) {
  return (size_t)hp.oct_nodes_cap * (sizeof(int4) + 3 * sizeof(QNode) + sizeof(orbx_sort::elem_t) +
                                      sizeof(int) + sizeof(unsigned short)) +
         (size_t)(hp.oct_cells_max + 1) * sizeof(int) + (size_t)hp.oct_keys_lds * 2 * sizeof(uint32_t) + 16;
}

constexpr int kOctTeamWaves = 4;

size_t octree_team_lds_bytes(const Plan& hp) {
  return (size_t)hp.oct_nodes_cap * (sizeof(int4) + 3 * sizeof(QNode) + sizeof(orbx_sort::elem_t) + sizeof(int) +
                                      2 * sizeof(unsigned short)) +
         (size_t)(hp.oct_team_keys + 8) * sizeof(uint2) + 2 * kOctTeamWaves * sizeof(uint2) +
         (size_t)hp.oct_team_keys * (2 * sizeof(uint32_t) + 2 * sizeof(unsigned short)) +
         (size_t)(hp.oct_cells_max + 1) * sizeof(int) + 16;
}

void launch_octree(const Plan& hp, const Plan* dp, const uint32_t* d_cand, const int* d_cellCount,
                   uint32_t* d_flatA, uint32_t* d_flatB, uint32_t* d_sel, int* d_selCount,
                   uint8_t* d_workspace, int l0, int l1, int f0, int nframes, hipStream_t s) {
  if (l1 <= l0) return;
  dim3 grid(nframes, l1 - l0);
  if (hp.oct_team_keys > 0 && nframes <= hp.oct_team_frames) {
    hipLaunchKernelGGL(k_octree_team<kOctTeamWaves>, grid, dim3(64, kOctTeamWaves), octree_team_lds_bytes(hp), s, dp,
                       d_cand, d_cellCount, d_flatA, d_flatB, d_sel, d_selCount, f0, l0);
    return;
  }
  const size_t lds = hp.oct_ws_bytes ? 0 : octree_lds_bytes(hp);
  hipLaunchKernelGGL(hp.oct_ws_bytes ? k_octree<true> : k_octree<false>, grid, dim3(64), lds, s, dp, d_cand, d_cellCount,
                     d_flatA, d_flatB, d_sel, d_selCount, d_workspace, f0, l0);
}

void launch_assemble(const Plan& hp, const Plan* dp, const uint32_t* d_sel, const int* d_selCount,
                     int lap0, int lap1, float* d_kps, WorkItem* d_work, int* d_n, int* d_mono,
                     int f0, int nframes, hipStream_t s) {
  (void)hp;
  // sixteen waves per frame take most frames in one trip: shortest for a few frames; in a large
  // batch a workgroup that needs sixteen free wave slots on one CU waits behind the detection
  // kernel's one-wave workgroups (headline 421 k -> 392 k frames/s), four waves do not
  if (nframes <= 64)
    hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(1024), 0, s, dp, d_sel, d_selCount, lap0, lap1,
                       d_kps, d_work, d_n, d_mono, f0);
  else
    hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(256), 0, s, dp, d_sel, d_selCount, lap0, lap1,
                     d_kps, d_work, d_n, d_mono, f0);
}

void launch_blur(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, uint8_t* d_blur, int f0,
                 int nframes, hipStream_t s) {
  const int nunits = hp.tiles_per_frame * nframes;
  if (hp.blur_waves == 4) {
    const int chunk = (nunits + 7) / 8;
    hipLaunchKernelGGL(k_blur, dim3(8 * chunk), dim3(256), 0, s, dp, d_pyr, d_blur, f0, hp.tiles_per_frame, nunits,
                       chunk);
  } else {
    const int chunk = (4 * nunits + 7) / 8;
    hipLaunchKernelGGL(k_blur, dim3(8 * chunk), dim3(64), 0, s, dp, d_pyr, d_blur, f0, hp.tiles_per_frame, nunits,
                       chunk);
  }
}

void launch_describe(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, const uint8_t* d_blur,
                     const WorkItem* d_work, const int* d_n, float* d_kps, uint8_t* d_desc,
                     int f0, int nframes, hipStream_t s) {
  const int nblk = (hp.kp_cap + 3) / 4, nunits = nblk * nframes, chunk = (nunits + 7) / 8;
  hipLaunchKernelGGL(k_orient_describe, dim3(8 * chunk), dim3(64), 0, s, dp, d_pyr, d_blur, d_work,
                     d_n, d_kps, d_desc, f0, nblk, nunits, chunk);
}

void launch_describe_fused(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, const WorkItem* d_work,
                           const int* d_n, float* d_kps, uint8_t* d_desc, int f0, int nframes, hipStream_t s) {
  const int nblk = (hp.kp_cap + 1) / 2, nunits = nblk * nframes, chunk = (nunits + 7) / 8;
  hipLaunchKernelGGL(k_orient_blur_describe, dim3(8 * chunk), dim3(64), 0, s, dp, d_pyr, d_work, d_n, d_kps, d_desc,
                     f0, nblk, nunits, chunk);
}

// Two word arrays to wherever the GPU can write (page-locked host memory: results of a blocking
// call without a copy engine).
__global__ __launch_bounds__(256) void k_copy_words2(const uint32_t* __restrict__ srcA, uint32_t* __restrict__ dstA,
                                                     int nA, const uint32_t* __restrict__ srcB,
                                                     uint32_t* __restrict__ dstB, int nB) {
  for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < nA + nB; i += (int)gridDim.x * 256) {
    if (i < nA) dstA[i] = srcA[i];
    else dstB[i - nA] = srcB[i - nA];
  }
}

void launch_copy_words2(const uint32_t* srcA, uint32_t* dstA, int nA, const uint32_t* srcB, uint32_t* dstB, int nB,
                        hipStream_t s) {
  const int blocks = (nA + nB) / 1024 + 1;
  hipLaunchKernelGGL(k_copy_words2, dim3(blocks < 64 ? blocks : 64), dim3(256), 0, s, srcA, dstA, nA, srcB, dstB, nB);
}

void launch_export_few(const int* d_n, const int* d_mono, const float* d_kps, const uint8_t* d_desc, int nframes,
                       int kc, int cap, int G, int* hcnt, void* hk, void* hd, hipStream_t s) {
  const int blocks = cap * 15 / 1024 + 1;
  hipLaunchKernelGGL(k_export_few, dim3(blocks < 64 ? blocks : 64, nframes), dim3(256), 0, s, d_n, d_mono,
                     (const uint32_t*)d_kps, (const uint32_t*)d_desc, kc, cap, G, hcnt, (uint32_t*)hk, (uint32_t*)hd);
}

int launch_debug_sort(uint64_t* d_data, int n, int team, hipStream_t s) {
  const size_t lds = (size_t)n * 16 + (size_t)debug_sort_stack_ints(n, team != 0) * 4 + (size_t)n * 4 + (size_t)(n / 64 + 1) * 8 + 32;
  if (lds > 96 * 1024) return -1;
  if (lds > 64 * 1024 - 64 &&
      hipFuncSetAttribute(team ? (const void*)k_debug_sort<kOctTeamWaves> : (const void*)k_debug_sort<1>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return -1;
  if (team)
    hipLaunchKernelGGL(k_debug_sort<kOctTeamWaves>, dim3(1), dim3(64, kOctTeamWaves), lds, s, (orbx_sort::elem_t*)d_data, n);
  else
    hipLaunchKernelGGL(k_debug_sort<1>, dim3(1), dim3(64, 1), lds, s, (orbx_sort::elem_t*)d_data, n);
  return 0;
}

int octree_set_lds_limit(const Plan& hp) {
  if (hp.oct_team_keys > 0) {
    const size_t need = octree_team_lds_bytes(hp) + 256;
    if (need > 160 * 1024) return -1;
    if (need > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k_octree_team<kOctTeamWaves>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)need) != hipSuccess)
      return -1;
  }
  if (hp.oct_ws_bytes) return 0;
  const size_t need = octree_lds_bytes(hp) + 256;
  if (need > 160 * 1024) return -1;
  if (need > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)k_octree<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)need) != hipSuccess)
      return -1;
  }
  return 0;
}

}  // namespace orbx
