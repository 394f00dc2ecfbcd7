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

// number of set bits of m below the calling lane
__device__ __forceinline__ int mbcnt64(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// ---------------------------------------------------------------------------
// K1a: level 0 = input image + 19-pixel REFLECT_101 border
// Thread = one aligned dword (4 pixels) of the padded plane.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_level0(const uint8_t* __restrict__ img, int W, int H,
                                                    int istride, size_t iframe,
                                                    uint8_t* __restrict__ dst, int pstride,
                                                    int prows, size_t dplane) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int py = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  if (c * 4 >= pstride || py >= prows) return;
  const int sy = reflect101(py - ORBX_EDGE, H);
  const uint8_t* srow = img + (size_t)f * iframe + (size_t)sy * istride;
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = 4 * c + k;
    if (px < W + 2 * ORBX_EDGE) {
      const int sx = reflect101(px - ORBX_EDGE, W);
      out |= (uint32_t)srow[sx] << (8 * k);
    }
  }
  *(uint32_t*)(dst + (size_t)f * dplane + (size_t)py * pstride + 4 * c) = out;
}

// ---------------------------------------------------------------------------
// K1b: level l = cv::resize(level l-1, INTER_LINEAR) + REFLECT_101 border.
// The tap tables are indexed by padded destination coordinates, so border
// pixels are produced by the same code as interior ones.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_resize(const uint8_t* __restrict__ src, int sstride,
                                                    size_t splane, uint8_t* __restrict__ dst,
                                                    int pstride, int prows, size_t dplane, int pw,
                                                    const Tap* __restrict__ xtab,
                                                    const Tap* __restrict__ ytab) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int py = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  if (c * 4 >= pstride || py >= prows) return;
  const Tap ty = ytab[py];
  const uint8_t* roi = src + (size_t)f * splane + (size_t)ORBX_EDGE * sstride + ORBX_EDGE;
  const uint8_t* r0 = roi + (size_t)ty.s0 * sstride;
  const uint8_t* r1 = roi + (size_t)ty.s1 * sstride;
  const int b0 = ty.w0, b1 = ty.w1;
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = 4 * c + k;
    if (px < pw) {
      const Tap tx = xtab[px];
      const int h0 = (int)r0[tx.s0] * tx.w0 + (int)r0[tx.s1] * tx.w1;
      const int h1 = (int)r1[tx.s0] * tx.w0 + (int)r1[tx.s1] * tx.w1;
      const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      out |= (uint32_t)(v & 0xFF) << (8 * k);
    }
  }
  *(uint32_t*)(dst + (size_t)f * dplane + (size_t)py * pstride + 4 * c) = out;
}

// ---------------------------------------------------------------------------
// K2: per-cell FAST-9/16 + non-max suppression + two-threshold retry +
// ordered compaction.  One 256-thread workgroup per grid cell.
//
// For a pixel with centre v and ring pixels I_k let
//   m = max over the 16 contiguous 9-arcs and both signs of min_arc (+-(v - I_k)).
// The pixel is a FAST corner at threshold t iff m > t, and OpenCV's corner
// score is then m - 1, independent of t.  One map of m therefore serves both
// thresholds of the reference's retry (extractor.cc:789-807).  cv::FAST runs
// on the cell's ROI, so suppression only sees neighbours inside the cell's own
// detection domain; domains tile [19, w-19) x [19, h-19) without overlap.
// ---------------------------------------------------------------------------
#define FAST_TS 80  // LDS row stride of the image tile (>= 70 + 6)
#define FAST_MS 76  // LDS row stride of the m map (>= 70 + 2)

__device__ __forceinline__ int fast_m(const uint8_t* c) {
  const int v = c[0];
  int d[16];
  d[0] = v - c[3 * FAST_TS + 0];
  d[1] = v - c[3 * FAST_TS + 1];
  d[2] = v - c[2 * FAST_TS + 2];
  d[3] = v - c[1 * FAST_TS + 3];
  d[4] = v - c[3];
  d[5] = v - c[-1 * FAST_TS + 3];
  d[6] = v - c[-2 * FAST_TS + 2];
  d[7] = v - c[-3 * FAST_TS + 1];
  d[8] = v - c[-3 * FAST_TS + 0];
  d[9] = v - c[-3 * FAST_TS - 1];
  d[10] = v - c[-2 * FAST_TS - 2];
  d[11] = v - c[-1 * FAST_TS - 3];
  d[12] = v - c[-3];
  d[13] = v - c[1 * FAST_TS - 3];
  d[14] = v - c[2 * FAST_TS - 2];
  d[15] = v - c[3 * FAST_TS - 1];
  int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo2[k] = min(d[k], d[(k + 1) & 15]);
    hi2[k] = max(d[k], d[(k + 1) & 15]);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
    hi4[k] = max(hi2[k], hi2[(k + 2) & 15]);
  }
  int best_dark = -256, best_bright = 256;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    // window of 9 starting at k: [k,k+4) u [k+4,k+8) u {k+8}
    const int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
    const int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
    best_dark = max(best_dark, lo9);
    best_bright = min(best_bright, hi9);
  }
  return max(max(best_dark, -best_bright), 0);
}

__global__ __launch_bounds__(256) void k_fast_cells(const Plan* __restrict__ plan,
                                                    const uint8_t* __restrict__ pyr,
                                                    uint32_t* __restrict__ cand,
                                                    int* __restrict__ cellCount) {
  __shared__ uint8_t tile[(ORBX_CELL_MAX + 6) * FAST_TS];
  __shared__ uint8_t mm[(ORBX_CELL_MAX + 2) * FAST_MS];
  __shared__ uint8_t selv[ORBX_CELL_MAX * ORBX_CELL_MAX];
  __shared__ int chunkCnt[96];

  const int tid = threadIdx.x;
  const int f = blockIdx.y;
  const int cellg = blockIdx.x;
  int l = 0;
  while (l + 1 < plan->nlevels && cellg >= plan->lv[l + 1].cell_first) ++l;
  const LevelDesc& L = plan->lv[l];
  const int cell = cellg - L.cell_first;
  const int ci = cell / L.ncx, cj = cell - ci * L.ncx;
  const int X0 = ORBX_EDGE + cj * L.wCell, Y0 = ORBX_EDGE + ci * L.hCell;
  const int cw = min(L.wCell, L.w - ORBX_EDGE - X0);
  const int ch = min(L.hCell, L.h - ORBX_EDGE - Y0);
  const int tw = cw + 6, th = ch + 6;

  // image tile with a 3-pixel ring halo; padded coords of its origin
  const uint8_t* src = pyr + L.pyr_off + (size_t)f * L.pyr_plane +
                       (size_t)(Y0 - 3 + ORBX_EDGE) * L.pstride + (X0 - 3 + ORBX_EDGE);
  for (int r = tid >> 5; r < th; r += 8) {
    const uint8_t* srow = src + (size_t)r * L.pstride;
    for (int c = tid & 31; c < tw; c += 32) tile[r * FAST_TS + c] = srow[c];
  }
  for (int i = tid; i < (ch + 2) * FAST_MS; i += 256) mm[i] = 0;
  __syncthreads();

  const int np = cw * ch;
  const float rcw = 1.0f / (float)cw;
  for (int p = tid; p < np; p += 256) {
    const int y = (int)(((float)p + 0.5f) * rcw);
    const int x = p - y * cw;
    mm[(y + 1) * FAST_MS + x + 1] = (uint8_t)fast_m(&tile[(y + 3) * FAST_TS + x + 3]);
  }
  __syncthreads();

  const int ini = plan->ini_th, mn = plan->min_th;
  int anyIni = 0;
  for (int p = tid; p < np; p += 256) {
    const int y = (int)(((float)p + 0.5f) * rcw);
    const int x = p - y * cw;
    const uint8_t* q = &mm[(y + 1) * FAST_MS + x + 1];
    const int m = q[0];
    const bool isMax = m > 0 && m > q[-1] && m > q[1] && m > q[-FAST_MS - 1] && m > q[-FAST_MS] &&
                       m > q[-FAST_MS + 1] && m > q[FAST_MS - 1] && m > q[FAST_MS] &&
                       m > q[FAST_MS + 1];
    selv[p] = isMax ? (uint8_t)m : (uint8_t)0;
    anyIni |= (isMax && m > ini);
  }
  const int thr = __syncthreads_or(anyIni) ? ini : mn;

  // ordered compaction: chunk = 64 consecutive pixels handled by one wave
  const int wave = tid >> 6, lane = tid & 63;
  const int nchunks = (np + 63) >> 6;
  for (int cidx = wave; cidx < nchunks; cidx += 4) {
    const int p = cidx * 64 + lane;
    const bool keep = p < np && selv[p] > thr;
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) chunkCnt[cidx] = __popcll(mask);
  }
  __syncthreads();
  if (wave == 0) {
    int running = 0;
    for (int base = 0; base < nchunks; base += 64) {
      const int i = base + lane;
      const int v = i < nchunks ? chunkCnt[i] : 0;
      const int inc = wave_incl_scan(v);
      if (i < nchunks) chunkCnt[i] = running + inc - v;
      running += __shfl(inc, 63, 64);
    }
    if (lane == 0) cellCount[(size_t)f * plan->cells_per_frame + cellg] = running;
  }
  __syncthreads();
  uint32_t* out = cand + (size_t)f * plan->cand_frame + L.cand_off + (size_t)cell * L.cand_cap_cell;
  for (int cidx = wave; cidx < nchunks; cidx += 4) {
    const int p = cidx * 64 + lane;
    const int m = p < np ? selv[p] : 0;
    const bool keep = m > thr;
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      const int y = (int)(((float)p + 0.5f) * rcw);
      const int x = p - y * cw;
      const uint32_t key = (uint32_t)(X0 + x - ORBX_DET_ORG) | ((uint32_t)(Y0 + y - ORBX_DET_ORG) << 12) |
                           ((uint32_t)(m - 1) << 24);
      out[chunkCnt[cidx] + mbcnt64(mask)] = key;
    }
  }
}

// ---------------------------------------------------------------------------
// K4: quadtree distribution.  One wave per (frame, level).
//
// Keys are 32-bit: x | y << 12 | score << 24 with x, y relative to (16,16).
// The node list is an array kept in list order; every node owns a contiguous
// segment of one of two key buffers and a split writes the four children into
// the same range of the other buffer (stable 4-way partition by ballots), so
// key order inside a node is the parent's order, as in DivideNode.
// A pass of the reference prepends children as they are created and erases
// the parents, so the list after a pass is: the children in reverse creation
// order, followed by the surviving old nodes in their old order.
// ---------------------------------------------------------------------------
struct QNode {
  short ulx, uly, brx, bry;
  int start;
  int cnt;  // bits 0..30 key count (0 = erased), bit 31 = key buffer id
};

__device__ __forceinline__ int qcount(const QNode& n) { return n.cnt & 0x7fffffff; }

struct OctState {
  QNode* newch;
  int nch;
  int nToExpand;
  uint32_t* bufA;
  uint32_t* bufB;
};

// Split one node (wave-cooperative).  Appends the non-empty children to
// st.newch in n1..n4 order; returns the number of non-empty children.
__device__ int split_node(const QNode nd, OctState& st) {
  const int lane = lane_id();
  const int cnt = qcount(nd);
  const int buf = (int)((unsigned)nd.cnt >> 31);
  const uint32_t* src = (buf ? st.bufB : st.bufA) + nd.start;
  uint32_t* dst = (buf ? st.bufA : st.bufB) + nd.start;
  const int mx = nd.ulx + ((nd.brx - nd.ulx + 1) >> 1);  // ceil(width / 2)
  const int my = nd.uly + ((nd.bry - nd.uly + 1) >> 1);
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  if (cnt <= 64) {
    const bool valid = lane < cnt;
    const uint32_t key = valid ? src[lane] : 0u;
    const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
    const int q = (x >= mx ? 1 : 0) + (y >= my ? 2 : 0);
    const unsigned long long m0 = __ballot(valid && q == 0), m1 = __ballot(valid && q == 1),
                             m2 = __ballot(valid && q == 2), m3 = __ballot(valid && q == 3);
    c0 = __popcll(m0); c1 = __popcll(m1); c2 = __popcll(m2); c3 = __popcll(m3);
    if (valid) {
      int pos;
      if (q == 0) pos = mbcnt64(m0);
      else if (q == 1) pos = c0 + mbcnt64(m1);
      else if (q == 2) pos = c0 + c1 + mbcnt64(m2);
      else pos = c0 + c1 + c2 + mbcnt64(m3);
      dst[pos] = key;
    }
  } else {
    for (int base = 0; base < cnt; base += 64) {
      const int i = base + lane;
      const bool valid = i < cnt;
      const uint32_t key = valid ? src[i] : 0u;
      const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
      const int q = (x >= mx ? 1 : 0) + (y >= my ? 2 : 0);
      c0 += __popcll(__ballot(valid && q == 0));
      c1 += __popcll(__ballot(valid && q == 1));
      c2 += __popcll(__ballot(valid && q == 2));
      c3 += __popcll(__ballot(valid && q == 3));
    }
    int r0 = 0, r1 = c0, r2 = c0 + c1, r3 = c0 + c1 + c2;
    for (int base = 0; base < cnt; base += 64) {
      const int i = base + lane;
      const bool valid = i < cnt;
      const uint32_t key = valid ? src[i] : 0u;
      const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
      const int q = (x >= mx ? 1 : 0) + (y >= my ? 2 : 0);
      const unsigned long long m0 = __ballot(valid && q == 0), m1 = __ballot(valid && q == 1),
                               m2 = __ballot(valid && q == 2), m3 = __ballot(valid && q == 3);
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
  }
  // children records (wave-uniform values; lane 0 stores)
  const int nb = (buf ^ 1) << 31;
  int made = 0;
  int start = nd.start;
  const int cc[4] = {c0, c1, c2, c3};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = cc[q];
    if (c > 0) {
      if (lane == 0) {
        QNode ch;
        ch.ulx = (q & 1) ? (short)mx : nd.ulx;
        ch.brx = (q & 1) ? nd.brx : (short)mx;
        ch.uly = (q & 2) ? (short)my : nd.uly;
        ch.bry = (q & 2) ? nd.bry : (short)my;
        ch.start = start;
        ch.cnt = c | nb;
        st.newch[st.nch + made] = ch;
      }
      ++made;
      if (c > 1) ++st.nToExpand;
    }
    start += c;
  }
  st.nch += made;
  return made;
}

// new list = reverse(newch[0..nch)) ++ [old nodes with cnt != 0, old order]
__device__ int rebuild_list(const QNode* cur, int n, const QNode* newch, int nch, QNode* nxt) {
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
__device__ int build_records(const QNode* newch, int nch, orbx_sort::elem_t* rec) {
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

__global__ __launch_bounds__(64) void k_octree(const Plan* __restrict__ plan,
                                               const uint32_t* __restrict__ cand,
                                               const int* __restrict__ cellCount,
                                               uint32_t* __restrict__ flatA,
                                               uint32_t* __restrict__ flatB,
                                               uint32_t* __restrict__ sel,
                                               int* __restrict__ selCount) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int l = blockIdx.x, f = blockIdx.y;
  const int lane = lane_id();
  const LevelDesc& L = plan->lv[l];
  const int cap = plan->oct_nodes_cap;
  QNode* cur = (QNode*)smem;
  QNode* nxt = cur + cap;
  QNode* newch = nxt + cap;
  orbx_sort::elem_t* rec = (orbx_sort::elem_t*)(newch + cap);
  int* cellpref = (int*)(rec + cap);
  __shared__ int rootCnt[ORBX_MAX_ROOTS];

  OctState st;
  st.newch = newch;
  st.bufA = flatA + (size_t)f * plan->flat_frame + L.flat_off;
  st.bufB = flatB + (size_t)f * plan->flat_frame + L.flat_off;

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
  __syncthreads();

  // flatten into B and count keys per root: root = int(x / hX) (extractor.cc:571-574)
  const uint32_t* candL = cand + (size_t)f * plan->cand_frame + L.cand_off;
  const int nroots = L.n_roots;
  const float hX = L.hX;
  for (int base = 0; base < total; base += 64) {
    const int i = base + lane;
    const bool valid = i < total;
    uint32_t key = 0;
    int r = -1;
    if (valid) {
      int lo = 0, hi = nc;  // largest c with cellpref[c] <= i
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cellpref[mid] <= i) lo = mid; else hi = mid;
      }
      key = candL[(size_t)lo * L.cand_cap_cell + (i - cellpref[lo])];
      st.bufB[i] = key;
      r = (int)__fdiv_rn((float)(key & 0xFFF), hX);
    }
    for (int rr = 0; rr < nroots; ++rr) {
      const unsigned long long m = __ballot(r == rr);
      if (lane == 0 && m) rootCnt[rr] += __popcll(m);
    }
  }
  __syncthreads();
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
          cur[n] = nd;
        }
        ++n;
      }
      acc += c;
    }
  }
  __syncthreads();

  const int N = L.quota;
  int nrec = 0;
  bool finish = false;
  while (!finish) {
    int prevSize = n;
    st.nch = 0;
    st.nToExpand = 0;
    for (int p = 0; p < n; ++p) {
      const QNode nd = cur[p];
      if (qcount(nd) <= 1) continue;
      split_node(nd, st);
      if (lane == 0) cur[p].cnt = 0;
    }
    __syncthreads();
    n = rebuild_list(cur, n, newch, st.nch, nxt);
    nrec = build_records(newch, st.nch, rec);
    { QNode* t = cur; cur = nxt; nxt = t; }
    __syncthreads();
    if (n >= N || n == prevSize) {
      finish = true;
    } else if (n + st.nToExpand * 3 > N) {
      while (!finish) {
        prevSize = n;
        if (lane == 0) orbx_sort::std_sort(rec, nrec);
        __syncthreads();
        st.nch = 0;
        st.nToExpand = 0;
        int running = n;
        for (int j = nrec - 1; j >= 0; --j) {
          const int pos = (int)(uint32_t)rec[j];
          const QNode nd = cur[pos];
          running += split_node(nd, st) - 1;
          if (lane == 0) cur[pos].cnt = 0;
          if (running >= N) break;
        }
        __syncthreads();
        n = rebuild_list(cur, n, newch, st.nch, nxt);
        nrec = build_records(newch, st.nch, rec);
        { QNode* t = cur; cur = nxt; nxt = t; }
        __syncthreads();
        if (n >= N || n == prevSize) finish = true;
      }
    }
  }

  // best key per node: highest score, first one wins ties (extractor.cc:726-742)
  uint32_t* out = sel + (size_t)f * plan->sel_frame + L.sel_off;
  for (int p = lane; p < n; p += 64) {
    const QNode nd = cur[p];
    const uint32_t* keys = (((unsigned)nd.cnt >> 31) ? st.bufB : st.bufA) + nd.start;
    const int cnt = qcount(nd);
    uint32_t best = keys[0];
    for (int k = 1; k < cnt; ++k) {
      const uint32_t key = keys[k];
      if ((key >> 24) > (best >> 24)) best = key;
    }
    if (p < L.sel_cap) out[p] = best;
  }
  if (lane == 0) selCount[(size_t)f * plan->nlevels + l] = n < L.sel_cap ? n : L.sel_cap;
}

// ---------------------------------------------------------------------------
// K8: output slots.  One workgroup per frame.  Walks the selected keys in
// (level, list order), scales coordinates, applies the lapping test and fills
// the output from both ends exactly like extractor.cc:1052-1100.  Writes the
// keypoint records (angle is filled in by k_orient_describe) and the work list.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble(const Plan* __restrict__ plan,
                                                  const uint32_t* __restrict__ sel,
                                                  const int* __restrict__ selCount, int lap0,
                                                  int lap1, float* __restrict__ kps,
                                                  WorkItem* __restrict__ work,
                                                  int* __restrict__ nOut, int* __restrict__ monoOut) {
  __shared__ int lvOff[ORBX_MAX_LEVELS + 1];
  __shared__ int waveCnt[4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nl = plan->nlevels;
  if (tid == 0) {
    int acc = 0;
    for (int l = 0; l < nl; ++l) {
      lvOff[l] = acc;
      acc += selCount[(size_t)f * nl + l];
    }
    lvOff[nl] = acc;
  }
  __syncthreads();
  const int n = lvOff[nl];
  const float flap0 = (float)lap0, flap1 = (float)lap1;
  int stereoBefore = 0;  // stereo keys in earlier chunks
  float* kout = kps + (size_t)f * plan->kp_cap * 7;
  WorkItem* wout = work + (size_t)f * plan->kp_cap;
  for (int base = 0; base < n; base += 256) {
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
    for (int w = 0; w < wave; ++w) before += waveCnt[w];
    const int chunkTotal = waveCnt[0] + waveCnt[1] + waveCnt[2] + waveCnt[3];
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
      wi.level = l;
      wi.xy = ((key & 0xFFF) + ORBX_DET_ORG) | ((((key >> 12) & 0xFFF) + ORBX_DET_ORG) << 16);
      wi.dst = dst;
      wi.pad = 0;
      wout[k] = wi;
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
// the REFLECT_101 extension the blur needs.  Tile = 64 x 32 output pixels.
// ---------------------------------------------------------------------------
#define BLUR_TW 64
#define BLUR_TH 32
__global__ __launch_bounds__(256) void k_blur(const Plan* __restrict__ plan,
                                              const uint8_t* __restrict__ pyr,
                                              uint8_t* __restrict__ blur) {
  __shared__ uint8_t sIn[(BLUR_TH + 6) * (BLUR_TW + 8)];
  __shared__ uint16_t sH[(BLUR_TH + 6) * BLUR_TW];
  const int tid = threadIdx.x, f = blockIdx.y;
  const int tg = blockIdx.x;
  int l = 0;
  while (l + 1 < plan->nlevels && tg >= plan->lv[l + 1].tile_first) ++l;
  const LevelDesc& L = plan->lv[l];
  const int t = tg - L.tile_first;
  const int tyi = t / L.tiles_x, txi = t - tyi * L.tiles_x;
  const int x0 = txi * BLUR_TW, y0 = tyi * BLUR_TH;
  const int SW = BLUR_TW + 8;
  // source rows y0-3 .. y0+TH+2, cols x0-3 .. x0+TW+2 (level coords) -> padded +19
  const uint8_t* src = pyr + L.pyr_off + (size_t)f * L.pyr_plane +
                       (size_t)(y0 - 3 + ORBX_EDGE) * L.pstride + (x0 - 3 + ORBX_EDGE);
  const int maxc = L.pstride - (x0 - 3 + ORBX_EDGE);      // readable columns
  const int maxr = L.prows - (y0 - 3 + ORBX_EDGE);        // readable rows
  for (int i = tid; i < (BLUR_TH + 6) * (BLUR_TW + 6); i += 256) {
    const int r = i / (BLUR_TW + 6), c = i - r * (BLUR_TW + 6);
    uint8_t v = 0;
    if (r < maxr && c < maxc) v = src[(size_t)r * L.pstride + c];
    sIn[r * SW + c] = v;
  }
  __syncthreads();
  for (int i = tid; i < (BLUR_TH + 6) * BLUR_TW; i += 256) {
    const int r = i >> 6, c = i & 63;
    const uint8_t* p = &sIn[r * SW + c];
    const int s = 18 * (p[0] + p[6]) + 34 * (p[1] + p[5]) + 48 * (p[2] + p[4]) + 56 * p[3];
    sH[i] = (uint16_t)s;
  }
  __syncthreads();
  uint8_t* dst = blur + L.blur_off + (size_t)f * L.blur_plane;
  for (int i = tid; i < BLUR_TH * BLUR_TW; i += 256) {
    const int r = i >> 6, c = i & 63;
    const int x = x0 + c, y = y0 + r;
    if (x < L.w && y < L.h) {
      const uint16_t* p = &sH[r * BLUR_TW + c];
      const uint32_t s = 18u * (p[0] + p[6 * BLUR_TW]) + 34u * (p[BLUR_TW] + p[5 * BLUR_TW]) +
                         48u * (p[2 * BLUR_TW] + p[4 * BLUR_TW]) + 56u * p[3 * BLUR_TW];
      dst[(size_t)y * L.bstride + x] = (uint8_t)((s + 32768u) >> 16);
    }
  }
}

// ---------------------------------------------------------------------------
// K5 + K7: orientation by intensity centroid, then the 256-bit rotated BRIEF
// descriptor.  One wave per keypoint.
// ---------------------------------------------------------------------------
__constant__ int8_t c_pattern[256 * 4] = {
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

__global__ __launch_bounds__(256) void k_orient_describe(const Plan* __restrict__ plan,
                                                         const uint8_t* __restrict__ pyr,
                                                         const uint8_t* __restrict__ blur,
                                                         const WorkItem* __restrict__ work,
                                                         const int* __restrict__ nOut,
                                                         float* __restrict__ kps,
                                                         uint8_t* __restrict__ desc) {
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= nOut[f]) return;
  const WorkItem wi = work[(size_t)f * plan->kp_cap + k];
  const LevelDesc& L = plan->lv[wi.level];
  const int cx = wi.xy & 0xFFFF, cy = wi.xy >> 16;

  // ---- IC_Angle on the un-blurred level (extractor.cc:75-99) ----------------
  // lanes: u = (lane & 31) - 15 (lane 31 of each half idles), half 0 takes
  // rows v = -15..0, half 1 rows v = 1..15
  const uint8_t* center = pyr + L.pyr_off + (size_t)f * L.pyr_plane +
                          (size_t)(cy + ORBX_EDGE) * L.pstride + (cx + ORBX_EDGE);
  const int u = (lane & 31) - 15;
  const int au = u < 0 ? -u : u;
  const int half = lane >> 5;
  int m10 = 0, m01 = 0;
  if (u <= 15) {
    const int vbeg = half ? 1 : -15, vend = half ? 15 : 0;
    for (int v = vbeg; v <= vend; ++v) {
      const int av = v < 0 ? -v : v;
      if (au <= plan->umax[av]) {
        const int val = center[v * L.pstride + u];
        m10 += u * val;
        m01 += v * val;
      }
    }
  }
  m10 = wave_sum(m10);
  m01 = wave_sum(m01);
  const float angle = fast_atan2_deg((float)m01, (float)m10);

  // ---- steered BRIEF on the blurred level (extractor.cc:102-145) ------------
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.0);
  const float rad = __fmul_rn(angle, factorPI);
  const float a = (float)cos((double)rad), b = (float)sin((double)rad);
  const uint8_t* bc = blur + L.blur_off + (size_t)f * L.blur_plane + (size_t)cy * L.bstride + cx;
  const int bs = L.bstride;
  unsigned long long bits[4];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int t = pass * 64 + lane;
    const float x0 = (float)c_pattern[4 * t + 0], y0 = (float)c_pattern[4 * t + 1];
    const float x1 = (float)c_pattern[4 * t + 2], y1 = (float)c_pattern[4 * t + 3];
    const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
    const int q0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
    const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
    const int q1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
    const int t0 = bc[r0 * bs + q0];
    const int t1 = bc[r1 * bs + q1];
    bits[pass] = __ballot(t0 < t1);
  }
  if (lane < 4) {
    unsigned long long w = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
    *(unsigned long long*)(desc + ((size_t)f * plan->kp_cap + wi.dst) * 32 + 8 * lane) = w;
  }
  if (lane == 0) kps[((size_t)f * plan->kp_cap + wi.dst) * 7 + 3] = angle;
}

// ---------------------------------------------------------------------------
// launch functions (host)
// ---------------------------------------------------------------------------
void launch_pyramid(const Plan& hp, const uint8_t* d_img, int istride, size_t iframe, uint8_t* d_pyr,
                    int nframes, hipStream_t s) {
  const dim3 blk(64, 4);
  {
    const LevelDesc& L = hp.lv[0];
    dim3 grid((L.pstride / 4 + 63) / 64, (L.prows + 3) / 4, nframes);
    hipLaunchKernelGGL(k_pyr_level0, grid, blk, 0, s, d_img, hp.W, hp.H, istride, iframe,
                       d_pyr + L.pyr_off, L.pstride, L.prows, L.pyr_plane);
  }
  for (int l = 1; l < hp.nlevels; ++l) {
    const LevelDesc& S = hp.lv[l - 1];
    const LevelDesc& L = hp.lv[l];
    dim3 grid((L.pstride / 4 + 63) / 64, (L.prows + 3) / 4, nframes);
    hipLaunchKernelGGL(k_pyr_resize, grid, blk, 0, s, d_pyr + S.pyr_off, S.pstride, S.pyr_plane,
                       d_pyr + L.pyr_off, L.pstride, L.prows, L.pyr_plane, L.w + 2 * ORBX_EDGE,
                       L.xtab, L.ytab);
  }
}

void launch_fast(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, uint32_t* d_cand,
                 int* d_cellCount, int nframes, hipStream_t s) {
  dim3 grid(hp.cells_per_frame, nframes);
  hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, s, dp, d_pyr, d_cand, d_cellCount);
}

size_t octree_lds_bytes(const Plan& hp) {
  return (size_t)hp.oct_nodes_cap * (3 * sizeof(QNode) + sizeof(orbx_sort::elem_t)) +
         (size_t)(hp.oct_cells_max + 1) * sizeof(int);
}

void launch_octree(const Plan& hp, const Plan* dp, const uint32_t* d_cand, const int* d_cellCount,
                   uint32_t* d_flatA, uint32_t* d_flatB, uint32_t* d_sel, int* d_selCount,
                   int nframes, hipStream_t s) {
  dim3 grid(hp.nlevels, nframes);
  hipLaunchKernelGGL(k_octree, grid, dim3(64), octree_lds_bytes(hp), s, dp, d_cand, d_cellCount,
                     d_flatA, d_flatB, d_sel, d_selCount);
}

void launch_assemble(const Plan& hp, const Plan* dp, const uint32_t* d_sel, const int* d_selCount,
                     int lap0, int lap1, float* d_kps, WorkItem* d_work, int* d_n, int* d_mono,
                     int nframes, hipStream_t s) {
  (void)hp;
  hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(256), 0, s, dp, d_sel, d_selCount, lap0, lap1,
                     d_kps, d_work, d_n, d_mono);
}

void launch_blur(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, uint8_t* d_blur, int nframes,
                 hipStream_t s) {
  dim3 grid(hp.tiles_per_frame, nframes);
  hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, s, dp, d_pyr, d_blur);
}

void launch_describe(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, const uint8_t* d_blur,
                     const WorkItem* d_work, const int* d_n, float* d_kps, uint8_t* d_desc,
                     int nframes, hipStream_t s) {
  dim3 grid((hp.kp_cap + 3) / 4, nframes);
  hipLaunchKernelGGL(k_orient_describe, grid, dim3(256), 0, s, dp, d_pyr, d_blur, d_work, d_n, d_kps,
                     d_desc);
}

int octree_set_lds_limit(const Plan& hp) {
  const size_t need = octree_lds_bytes(hp) + 256;
  if (need > 160 * 1024) return -1;
  if (need > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)need) != hipSuccess)
      return -1;
  }
  return 0;
}

}  // namespace orbx
