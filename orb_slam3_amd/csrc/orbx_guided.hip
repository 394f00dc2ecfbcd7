// Guided matchers on a device-resident frame (undistorted keypoints, descriptors and the
// 64 x 48 frame grid of orbx_grid.hip):
//   k_search_by_projection_last   ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)
//                                 core/lib/ORB/matcher.cc:1591-1808 (monocular / rectified case);
//                                 SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist),
//                                 :1810-1923, is the same loop with other inputs
//   k_search_by_projection_points ORBmatcher::SearchByProjection(F, vpMapPoints, th, ...)
//                                 matcher.cc:41-142 (the local-map matcher; same case)
//   k_search_for_initialization   ORBmatcher::SearchForInitialization   matcher.cc:629-733
//   k_bow_match_nodes / k_bow_match_finish   ORBmatcher::SearchByBoW(pKF, F, ...)  matcher.cc:222-405
//                                 (same case; it walks the two feature vectors instead of the grid)
//   three_maxima                  ORBmatcher::ComputeThreeMaxima        matcher.cc:1925-1957
//
// Both reference loops are greedy over the query index: what query i may take depends on what
// the queries before it took (a keypoint that received a map point with observations is closed
// to later queries; a keypoint already matched at distance d only accepts a closer match).
// Here every query decides on its own from the claims that LOWER-indexed queries made in the
// previous round, and rounds repeat until no decision changes.  Query 0 depends on nobody and
// is right after round 1; query i is right once everything below it is; so the fixed point is
// unique, equals the sequential result, and is reached after (longest dependency chain + 1)
// rounds - a handful in practice, never more than the number of queries + 1.
//
// One workgroup per call: the work is a few thousand window walks of a few dozen 256-bit
// distances each, the rounds need workgroup-wide barriers, and the caller (Tracking) has one
// frame at a time.  Latency-bound by design: the first round walks the grid (chains of
// dependent loads: cell list -> keypoint -> descriptor) and remembers every query's candidates
// with their distances; the later rounds only re-read that list (queries with more than
// GUIDED_CACHE candidates walk again).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/orbx.h"
#include "orbx_plan.h"

#define GRID_COLS 64   /* FRAME_GRID_COLS, core/include/Frame.h:47 */
#define GRID_ROWS 48   /* FRAME_GRID_ROWS, core/include/Frame.h:46 */
#define HISTO_LENGTH 30   /* ORBmatcher::HISTO_LENGTH, matcher.cc:36 */
#define TH_HIGH 100       /* matcher.cc:34 */
#define TH_LOW 50         /* matcher.cc:35 */
#define GUIDED_THREADS 1024   /* one query per thread for a usual frame */
#define GUIDED_CACHE 48       /* candidates remembered per query (packed distance << 16 | keypoint) */

namespace orbx {

namespace {

struct GKP { float x, y, size, angle, response; int octave, class_id; };

struct GridView {
  const GKP* U;        // undistorted keypoints of the frame
  const int* cs;       // cell starts, 64*48 + 1
  const int* ent;      // cell entries
  float minX, maxX, minY, maxY, invW, invH;
};

// Frame::GetFeaturesInArea (core/src/Frame.cc:682-749): calls visit(i2) for every keypoint of
// the window, in the reference's order (cell column, cell row, order inside the cell).
template <class F>
__device__ __forceinline__ void for_each_in_area(const GridView& g, float x, float y, float r, int minLevel,
                                                 int maxLevel, F visit) {
  const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, g.minX), r), g.invW)));
  if (cx0 >= GRID_COLS) return;
  const int cx1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, g.minX), r), g.invW)));
  if (cx1 < 0) return;
  const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, g.minY), r), g.invH)));
  if (cy0 >= GRID_ROWS) return;
  const int cy1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, g.minY), r), g.invH)));
  if (cy1 < 0) return;
  const bool levelFilter = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = cx0; ix <= cx1; ++ix)
    for (int iy = cy0; iy <= cy1; ++iy) {
      const int c = ix * GRID_ROWS + iy;
      const int e1 = g.cs[c + 1];
      for (int e = g.cs[c]; e < e1; ++e) {
        const int i2 = g.ent[e];
        const GKP kp = g.U[i2];
        if (levelFilter) {
          if (kp.octave < minLevel) continue;
          if (maxLevel >= 0 && kp.octave > maxLevel) continue;
        }
        const float distx = __fsub_rn(kp.x, x), disty = __fsub_rn(kp.y, y);
        if (fabsf(distx) < r && fabsf(disty) < r) visit(i2);
      }
    }
}

// Words that other threads of the workgroup update with atomics (which execute in L2) are
// read with agent-scope atomic loads, so that a read never depends on how the CU's vector
// cache treats a line it holds while an atomic goes past it.
__device__ __forceinline__ int coherent_load(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ORBmatcher::DescriptorDistance (matcher.cc:1961-1975) of a descriptor held in registers
__device__ __forceinline__ int hamming_reg(const uint32_t (&a)[8], const uint32_t* __restrict__ b) {
  int d = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) d += __builtin_popcount(a[w] ^ b[w]);
  return d;
}

// ---- a window walked by a whole wave -------------------------------------------------------
// One thread walking a window is a chain of dependent loads per keypoint (cell bounds -> entry ->
// keypoint -> descriptor), a few hundred of them for the large windows of SearchForInitialization
// or of a last-frame keypoint from a coarse level.  Here the 64 lanes of a wave share one window:
// the cells of a grid column are consecutive in the cell table (c = column * 48 + row), so the
// window is one contiguous run of entries per column; the lanes take the runs' entries in the
// reference's order (column, row, order inside the cell), 64 at a time, and the accepted ones are
// written to the query's candidate list in that order (ballot + prefix count).
// colS / colP: 64 ints each, the calling wave's own.
__device__ __forceinline__ void wave_mem_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <class Accept, class Encode>
__device__ __forceinline__ void wave_gather_window(const GridView& g, float x, float y, float r, int minLevel,
                                                   int maxLevel, const uint32_t* __restrict__ qdesc,
                                                   const uint32_t* __restrict__ desc2, Accept accept, Encode encode,
                                                   uint32_t* __restrict__ cand, size_t stride, int qi,
                                                   int* __restrict__ countSlot, int* colS, int* colP) {
  const int lane = (int)(threadIdx.x & 63u);
  int k = 0;
  // the cell range of Frame::GetFeaturesInArea, as in for_each_in_area
  const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, g.minX), r), g.invW)));
  const int cx1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, g.minX), r), g.invW)));
  const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, g.minY), r), g.invH)));
  const int cy1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, g.minY), r), g.invH)));
  if (cx0 < GRID_COLS && cx1 >= 0 && cy0 < GRID_ROWS && cy1 >= 0 && cx1 >= cx0 && cy1 >= cy0) {
    const bool levelFilter = (minLevel > 0) || (maxLevel >= 0);
    const int ncols = cx1 - cx0 + 1, ny = cy1 - cy0 + 1;
    int first = 0, len = 0;
    if (lane < ncols) {
      const int c0 = (cx0 + lane) * GRID_ROWS + cy0;
      first = g.cs[c0];
      len = g.cs[c0 + ny] - first;
    }
    int inc = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    const int total = __shfl(inc, 63);
    colS[lane] = first;
    colP[lane] = inc - len;
    wave_mem_sync();
    uint32_t q[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) q[w] = qdesc[w];
    for (int fb = 0; fb < total; fb += 64) {
      const int f = fb + lane;
      bool ok = f < total;
      int i2 = 0, octave = 0, dist = 0;
      if (ok) {
        int lo = 0, hi = ncols;   // the last column whose run starts at or before f
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (colP[mid] <= f) lo = mid; else hi = mid;
        }
        i2 = g.ent[colS[lo] + (f - colP[lo])];
        const GKP kp = g.U[i2];
        octave = kp.octave;
        if (levelFilter && (kp.octave < minLevel || (maxLevel >= 0 && kp.octave > maxLevel))) ok = false;
        const float distx = __fsub_rn(kp.x, x), disty = __fsub_rn(kp.y, y);
        if (!(fabsf(distx) < r && fabsf(disty) < r)) ok = false;
        if (ok && !accept(i2)) ok = false;
        if (ok) dist = hamming_reg(q, desc2 + (size_t)i2 * 8);
      }
      const unsigned long long mask = __ballot(ok);
      const int pos = k + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      if (ok && pos < GUIDED_CACHE) cand[(size_t)pos * stride + qi] = encode(dist, octave, i2);
      k += __popcll(mask);
    }
  }
  if (lane == 0) *countSlot = k <= GUIDED_CACHE ? k : -1;
}

// rotation histogram bin as written at matcher.cc:694-700 / :1691-1699: the product with
// 1/HISTO_LENGTH (not a division by the bin width), round() half away from zero
__device__ __forceinline__ int rotation_bin(float angle1, float angle2) {
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = __fsub_rn(angle1, angle2);
  if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
  int bin = (int)roundf(__fmul_rn(rot, factor));
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

// ORBmatcher::ComputeThreeMaxima, matcher.cc:1925-1957
__device__ void three_maxima(const int* histo, int L, int* ind) {
  int top0 = 0, top1 = 0, top2 = 0;
  int bin0 = -1, bin1 = -1, bin2 = -1;
  for (int i = 0; i < L; ++i) {
    const int s = histo[i];
    if (s > top0) {
      top2 = top1; top1 = top0; top0 = s;
      bin2 = bin1; bin1 = bin0; bin0 = i;
    } else if (s > top1) {
      top2 = top1; top1 = s;
      bin2 = bin1; bin1 = i;
    } else if (s > top2) {
      top2 = s;
      bin2 = i;
    }
  }
  if ((float)top1 < __fmul_rn(0.1f, (float)top0)) { bin1 = -1; bin2 = -1; }
  else if ((float)top2 < __fmul_rn(0.1f, (float)top0)) { bin2 = -1; }
  ind[0] = bin0; ind[1] = bin1; ind[2] = bin2;
}

}  // namespace

struct ProjectionArgs {
  // last frame, one entry per keypoint of the last frame (matcher.cc:1607-1613, 1632-1635, 1655)
  const uint8_t* lastValid;
  const float* uv;
  const float* invz;
  const int* lastOctave;
  const float* lastAngle;
  const uint32_t* lastDesc;
  const uint8_t* lastObs;
  int nLast;
  // current frame
  GridView grid;
  const uint32_t* desc;
  const float* uRight;     // may be null: monocular
  const uint8_t* state;    // may be null: every keypoint free
  int n;
  float scale[ORBX_MAX_LEVELS];
  float th, mbf;
  int forward, backward, checkOrientation;
  int maxDist;             // TH_HIGH for the last-frame matcher, ORBdist for the keyframe one (:1885)
  // work arrays: two generations of choices (nLast) and of closing times (n); the candidate
  // lists of round 1, slot-major ([GUIDED_CACHE][nLast]) so that a wave reads them coalesced
  int* choice[2];
  int* closedAt[2];
  uint32_t* cand;
  int* candCount;          // per query: candidates cached, or -1 = too many, walk again
  // results
  int* assigned;           // n
  int* summary;            // [0] nmatches, [1] rounds
  int gathered;            // the candidate lists were filled by a gather launch before this one
  // ---- the second camera of a rig (matcher.cc:1712-1786); all unset for one camera
  // gate: the first camera's candCount array: a last-frame keypoint is searched in the second
  // camera only if its search in the first one got as far as a non-empty window (:1624-1652)
  const int* gate;
  int skipBounds;          // the projection into the second camera is not checked against the image
  // deferRotation: leave the rotation check to k_rig_rotation_finish (one histogram for both
  // cameras); finalChoice[i] = the keypoint query i ended up with, or -1
  int deferRotation;
  int* finalChoice;
};

// choice of last-frame keypoint i given the closing times of the previous round
// (matcher.cc:1607-1690): the keypoint it matches, or -1.  store: remember the candidates.
// decide = false only walks and stores (closedAt is not read).
__device__ __forceinline__ int projection_choice(const ProjectionArgs& a, int i, const int* closedAt, bool store, bool decide = true) {
  if (store) a.candCount[i] = 0;
  if (!a.lastValid[i]) return -1;
  if (a.gate && a.gate[i] == 0) return -1;
  const float invzc = a.invz[i];
  if (invzc < 0) return -1;
  const float u = a.uv[2 * i], v = a.uv[2 * i + 1];
  if (!a.skipBounds) {
    if (u < a.grid.minX || u > a.grid.maxX) return -1;
    if (v < a.grid.minY || v > a.grid.maxY) return -1;
  }
  const int nLastOctave = a.lastOctave[i];
  const float radius = __fmul_rn(a.th, a.scale[nLastOctave]);
  int minLevel, maxLevel;
  if (a.forward) { minLevel = nLastOctave; maxLevel = -1; }
  else if (a.backward) { minLevel = 0; maxLevel = nLastOctave; }
  else { minLevel = nLastOctave - 1; maxLevel = nLastOctave + 1; }
  uint32_t dMP[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) dMP[w] = a.lastDesc[(size_t)i * 8 + w];
  const float ur = __fsub_rn(u, __fmul_rn(a.mbf, invzc));
  int bestDist = 256, bestIdx2 = -1, k = 0;
  for_each_in_area(a.grid, u, v, radius, minLevel, maxLevel, [&](int i2) {
    if (a.uRight) {
      const float r2 = a.uRight[i2];
      if (r2 > 0) {
        const float er = fabsf(__fsub_rn(ur, r2));
        if (er > radius) return;
      }
    }
    const int dist = hamming_reg(dMP, a.desc + (size_t)i2 * 8);
    if (store) {
      if (k < GUIDED_CACHE) a.cand[(size_t)k * a.nLast + i] = ((uint32_t)dist << 16) | (uint32_t)i2;
      ++k;
    }
    // closed: the keypoint holds a map point with observations, from the start (-1) or
    // since an earlier query put one there (matcher.cc:1663-1666)
    if (!decide || coherent_load(&closedAt[i2]) < i) return;
    if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
  });
  if (store) a.candCount[i] = k <= GUIDED_CACHE ? k : -1;
  return bestDist <= a.maxDist ? bestIdx2 : -1;
}

// the same decision from the remembered candidates (same order, so the same ties)
__device__ __forceinline__ int projection_choice_cached(const ProjectionArgs& a, int i, const int* closedAt, int cnt) {
  int bestDist = 256, bestIdx2 = -1;
  for (int k = 0; k < cnt; ++k) {
    const uint32_t e = a.cand[(size_t)k * a.nLast + i];
    const int i2 = (int)(e & 0xFFFFu), dist = (int)(e >> 16);
    if (coherent_load(&closedAt[i2]) < i) continue;
    if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
  }
  return bestDist <= a.maxDist ? bestIdx2 : -1;
}

// The window walks of all queries, spread over the chip: they do not depend on each other, and
// one workgroup (one CU) would take them a few queries per thread, one dependent load after the
// other.  The rounds kernel then starts from the remembered lists.
#define GATHER_THREADS 256
#define GATHER_WAVES (GATHER_THREADS / 64)   /* one query per wave */
__global__ __launch_bounds__(GATHER_THREADS) void k_gather_projection(ProjectionArgs a) {
  __shared__ int colS[GATHER_WAVES][64], colP[GATHER_WAVES][64];
  const int w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
  const int i = (int)blockIdx.x * GATHER_WAVES + w;
  if (i >= a.nLast) return;
  // the tests of projection_choice before its walk
  bool walk = a.lastValid[i] != 0 && !(a.gate && a.gate[i] == 0);
  const float invzc = walk ? a.invz[i] : 0.0f;
  if (invzc < 0) walk = false;
  const float u = walk ? a.uv[2 * i] : 0.0f, v = walk ? a.uv[2 * i + 1] : 0.0f;
  if (walk && !a.skipBounds && (u < a.grid.minX || u > a.grid.maxX || v < a.grid.minY || v > a.grid.maxY)) walk = false;
  if (!walk) {
    if (lane == 0) a.candCount[i] = 0;
    return;
  }
  const int nLastOctave = a.lastOctave[i];
  const float radius = __fmul_rn(a.th, a.scale[nLastOctave]);
  int minLevel, maxLevel;
  if (a.forward) { minLevel = nLastOctave; maxLevel = -1; }
  else if (a.backward) { minLevel = 0; maxLevel = nLastOctave; }
  else { minLevel = nLastOctave - 1; maxLevel = nLastOctave + 1; }
  const float ur = __fsub_rn(u, __fmul_rn(a.mbf, invzc));
  wave_gather_window(
      a.grid, u, v, radius, minLevel, maxLevel, a.lastDesc + (size_t)i * 8, a.desc,
      [&](int i2) {
        if (!a.uRight) return true;
        const float r2 = a.uRight[i2];
        return !(r2 > 0 && fabsf(__fsub_rn(ur, r2)) > radius);
      },
      [](int dist, int, int i2) { return ((uint32_t)dist << 16) | (uint32_t)i2; }, a.cand, (size_t)a.nLast, i,
      &a.candCount[i], colS[w], colP[w]);
}

// The rounds of the last-frame matcher with the two generations of choices and closing times
// where the caller puts them.  Inlined once with LDS arrays (a usual frame) and once with the
// caller's work memory, so that in the first every access of a round is an LDS instruction:
// the arrays are read and rewritten by all threads in every round, and in global memory every one
// of those is a trip to the L2, where its atomics execute.
#define ROUNDS_LDS_MAX 2048   /* queries and keypoints per LDS array */
__device__ __forceinline__ void projection_rounds(const ProjectionArgs& a, int* choice0, int* choice1, int* closed0,
                                                  int* closed1, int* hist, int* keep, int* nmS) {
  const int tid = threadIdx.x;
  const int OPEN = 0x7fffffff;
  for (int i = tid; i < a.nLast; i += GUIDED_THREADS) choice0[i] = -1;
  for (int i2 = tid; i2 < a.n; i2 += GUIDED_THREADS)
    closed0[i2] = (a.state && a.state[i2] == 1) ? -1 : OPEN;
  if (tid < HISTO_LENGTH) hist[tid] = 0;
  if (tid == 0) *nmS = 0;
  __syncthreads();
  int cur = 0, rounds = 0;
  for (;;) {
    const int* cPrev = cur ? choice1 : choice0;
    int* cNew = cur ? choice0 : choice1;
    const int* bPrev = cur ? closed1 : closed0;
    int* bNew = cur ? closed0 : closed1;
    int changed = 0;
    for (int i = tid; i < a.nLast; i += GUIDED_THREADS) {
      int c;
      if (rounds == 0 && !a.gathered) {
        c = projection_choice(a, i, bPrev, true);
      } else {
        const int cnt = a.candCount[i];
        c = cnt >= 0 ? projection_choice_cached(a, i, bPrev, cnt) : projection_choice(a, i, bPrev, false);
      }
      cNew[i] = c;
      changed |= (c != cPrev[i]) ? 1 : 0;
    }
    for (int i2 = tid; i2 < a.n; i2 += GUIDED_THREADS)
      bNew[i2] = (a.state && a.state[i2] == 1) ? -1 : OPEN;
    __syncthreads();
    // a keypoint closes at the first query that puts a map point with observations there
    for (int i = tid; i < a.nLast; i += GUIDED_THREADS) {
      const int c = cNew[i];
      if (c >= 0 && a.lastObs[i]) atomicMin(&bNew[c], i);
    }
    ++rounds;
    cur ^= 1;
    const int any = __syncthreads_or(changed);
    if (!any || rounds > a.nLast + 1) break;
  }
  // matcher.cc:1682-1700: the last query that took a keypoint keeps it; every take counts
  const int* choice = cur ? choice1 : choice0;
  for (int i2 = tid; i2 < a.n; i2 += GUIDED_THREADS) a.assigned[i2] = -1;
  __syncthreads();
  for (int i = tid; i < a.nLast; i += GUIDED_THREADS) {
    const int c = choice[i];
    if (c >= 0) {
      atomicMax(&a.assigned[c], i);
      atomicAdd(nmS, 1);
      if (a.checkOrientation && !a.deferRotation) {
        const int bin = rotation_bin(a.lastAngle[i], a.grid.U[c].angle);
        if (bin >= 0 && bin < HISTO_LENGTH) atomicAdd(&hist[bin], 1);
      }
    }
    if (a.deferRotation) a.finalChoice[i] = c;
  }
  __syncthreads();
  if (a.checkOrientation && !a.deferRotation) {
    // matcher.cc:1788-1805: keypoints of every bin but the three largest lose their map point
    if (tid == 0) three_maxima(hist, HISTO_LENGTH, keep);
    __syncthreads();
    for (int i = tid; i < a.nLast; i += GUIDED_THREADS) {
      const int c = choice[i];
      if (c >= 0) {
        const int bin = rotation_bin(a.lastAngle[i], a.grid.U[c].angle);
        if (bin >= 0 && bin < HISTO_LENGTH && bin != keep[0] && bin != keep[1] && bin != keep[2]) {
          a.assigned[c] = -2;
          atomicSub(nmS, 1);
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) { a.summary[0] = *nmS; a.summary[1] = rounds; }
}

__global__ __launch_bounds__(GUIDED_THREADS) void k_search_by_projection_last(ProjectionArgs a) {
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int keep[3];
  __shared__ int nmS;
  __shared__ int gen[4 * ROUNDS_LDS_MAX];
  if (a.nLast <= ROUNDS_LDS_MAX && a.n <= ROUNDS_LDS_MAX)
    projection_rounds(a, gen, gen + ROUNDS_LDS_MAX, gen + 2 * ROUNDS_LDS_MAX, gen + 3 * ROUNDS_LDS_MAX, hist, keep, &nmS);
  else
    projection_rounds(a, a.choice[0], a.choice[1], a.closedAt[0], a.closedAt[1], hist, keep, &nmS);
}

// The rotation check of matcher.cc:1788-1805 over the matches of both cameras of a rig: one
// histogram, the three largest bins kept, every entry of the other bins cleared (-2).
// summary[0] = nmatches of both cameras after the check, summary[1] = the larger round count.
struct RigFinishArgs {
  int nLast;
  const int* choiceL;
  const int* choiceR;
  const float* lastAngle;
  const GKP* UL;
  const GKP* UR;
  int* assignedL;
  int* assignedR;
  const int* summaryL;
  const int* summaryR;
  int* summary;
  int checkOrientation;
};

__global__ __launch_bounds__(GUIDED_THREADS) void k_rig_rotation_finish(RigFinishArgs a) {
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int keep[3];
  __shared__ int cleared;
  const int tid = threadIdx.x;
  if (tid < HISTO_LENGTH) hist[tid] = 0;
  if (tid == 0) cleared = 0;
  __syncthreads();
  if (a.checkOrientation) {
    for (int i = tid; i < a.nLast; i += GUIDED_THREADS) {
      const int cl = a.choiceL[i], cr = a.choiceR[i];
      if (cl >= 0) {
        const int bin = rotation_bin(a.lastAngle[i], a.UL[cl].angle);
        if (bin >= 0 && bin < HISTO_LENGTH) atomicAdd(&hist[bin], 1);
      }
      if (cr >= 0) {
        const int bin = rotation_bin(a.lastAngle[i], a.UR[cr].angle);
        if (bin >= 0 && bin < HISTO_LENGTH) atomicAdd(&hist[bin], 1);
      }
    }
    __syncthreads();
    if (tid == 0) three_maxima(hist, HISTO_LENGTH, keep);
    __syncthreads();
    for (int i = tid; i < a.nLast; i += GUIDED_THREADS) {
      const int cl = a.choiceL[i], cr = a.choiceR[i];
      if (cl >= 0) {
        const int bin = rotation_bin(a.lastAngle[i], a.UL[cl].angle);
        if (bin >= 0 && bin < HISTO_LENGTH && bin != keep[0] && bin != keep[1] && bin != keep[2]) {
          a.assignedL[cl] = -2;
          atomicAdd(&cleared, 1);
        }
      }
      if (cr >= 0) {
        const int bin = rotation_bin(a.lastAngle[i], a.UR[cr].angle);
        if (bin >= 0 && bin < HISTO_LENGTH && bin != keep[0] && bin != keep[1] && bin != keep[2]) {
          a.assignedR[cr] = -2;
          atomicAdd(&cleared, 1);
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    a.summary[0] = a.summaryL[0] + a.summaryR[0] - cleared;
    a.summary[1] = max(a.summaryL[1], a.summaryR[1]);
  }
}

struct PointsArgs {
  // map points, one entry each: what Frame::isInFrustum left in them (matcher.cc:51-72, 94)
  const uint8_t* inView;
  const float* proj;       // mTrackProjX, mTrackProjY
  const float* projXR;     // mTrackProjXR
  const int* level;        // mnTrackScaleLevel
  const float* viewCos;    // mTrackViewCos
  const uint32_t* ptDesc;
  const uint8_t* ptObs;
  int nPts;
  // frame
  GridView grid;
  const uint32_t* desc;
  const float* uRight;     // may be null: monocular
  const uint8_t* state;    // may be null: every keypoint free
  int n;
  float scale[ORBX_MAX_LEVELS];
  float th, nnratio;
  // work arrays, as in ProjectionArgs
  int* choice[2];
  int* closedAt[2];
  uint32_t* cand;          // distance << 20 | octave << 16 | keypoint
  int* candCount;
  // results
  int* assigned;           // n
  int* summary;            // [0] nmatches, [1] rounds
  int gathered;            // as in ProjectionArgs
};

// best and second-best candidate with their pyramid levels, and the acceptance rule of
// matcher.cc:77-127: the ratio test only applies when both come from the same level
struct BestTwo {
  int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
  __device__ __forceinline__ void consider(int i2, int dist, int octave) {
    if (dist < bestDist) {
      bestDist2 = bestDist; bestDist = dist;
      bestLevel2 = bestLevel; bestLevel = octave;
      bestIdx = i2;
    } else if (dist < bestDist2) {
      bestLevel2 = octave;
      bestDist2 = dist;
    }
  }
  __device__ __forceinline__ int accepted(float nnratio) const {
    if (bestDist > TH_HIGH) return -1;
    if (bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2)) return -1;
    return bestIdx;
  }
  // the same with the two ways of failing told apart: -1 no candidate within TH_HIGH, -2 the
  // ratio test failed (the `continue` of matcher.cc:123-124, which leaves the map point's loop body)
  __device__ __forceinline__ int accepted3(float nnratio) const {
    if (bestDist > TH_HIGH) return -1;
    if (bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2)) return -2;
    return bestIdx;
  }
};

// The candidates of map point i in window order (matcher.cc:64-118), each offered to `b` unless
// closed(i2) says the keypoint holds a map point with observations (:89-91).  store: remember
// the candidates.  decide = false only walks and stores.
template <class Closed>
__device__ __forceinline__ BestTwo points_walk(const PointsArgs& a, int i, Closed closed, bool store, bool decide) {
  BestTwo b;
  if (store) a.candCount[i] = 0;
  if (!a.inView[i]) return b;
  const int nPredictedLevel = a.level[i];
  // RadiusByViewingCos (matcher.cc:215-220; the comparison is made in double)
  float r = (double)a.viewCos[i] > 0.998 ? 2.5f : 4.0f;
  if ((double)a.th != 1.0) r = __fmul_rn(r, a.th);
  const float radius = __fmul_rn(r, a.scale[nPredictedLevel]);
  uint32_t dMP[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) dMP[w] = a.ptDesc[(size_t)i * 8 + w];
  const float xr = a.uRight ? a.projXR[i] : 0.0f;
  int k = 0;
  for_each_in_area(a.grid, a.proj[2 * i], a.proj[2 * i + 1], radius, nPredictedLevel - 1, nPredictedLevel, [&](int i2) {
    if (a.uRight) {
      const float r2 = a.uRight[i2];
      if (r2 > 0) {
        const float er = fabsf(__fsub_rn(xr, r2));
        if (er > radius) return;
      }
    }
    const int dist = hamming_reg(dMP, a.desc + (size_t)i2 * 8);
    const int octave = a.grid.U[i2].octave;
    if (store) {
      if (k < GUIDED_CACHE)
        a.cand[(size_t)k * a.nPts + i] = ((uint32_t)dist << 20) | ((uint32_t)octave << 16) | (uint32_t)i2;
      ++k;
    }
    if (!decide || closed(i2)) return;
    b.consider(i2, dist, octave);
  });
  if (store) a.candCount[i] = k <= GUIDED_CACHE ? k : -1;
  return b;
}

// the same from the remembered candidates (same order, so the same ties)
template <class Closed>
__device__ __forceinline__ BestTwo points_walk_cached(const PointsArgs& a, int i, Closed closed, int cnt) {
  BestTwo b;
  for (int k = 0; k < cnt; ++k) {
    const uint32_t e = a.cand[(size_t)k * a.nPts + i];
    const int i2 = (int)(e & 0xFFFFu);
    if (closed(i2)) continue;
    b.consider(i2, (int)(e >> 20), (int)((e >> 16) & 0xFu));
  }
  return b;
}

// choice of map point i given the closing times of the previous round
__device__ __forceinline__ int points_choice(const PointsArgs& a, int i, const int* closedAt, bool store, bool decide = true) {
  return points_walk(a, i, [&](int i2) { return coherent_load(&closedAt[i2]) < i; }, store, decide).accepted(a.nnratio);
}

__device__ __forceinline__ int points_choice_cached(const PointsArgs& a, int i, const int* closedAt, int cnt) {
  return points_walk_cached(a, i, [&](int i2) { return coherent_load(&closedAt[i2]) < i; }, cnt).accepted(a.nnratio);
}

__global__ __launch_bounds__(GATHER_THREADS) void k_gather_points(PointsArgs a) {
  __shared__ int colS[GATHER_WAVES][64], colP[GATHER_WAVES][64];
  const int w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
  const int i = (int)blockIdx.x * GATHER_WAVES + w;
  if (i >= a.nPts) return;
  if (!a.inView[i]) {
    if (lane == 0) a.candCount[i] = 0;
    return;
  }
  // as points_walk
  const int nPredictedLevel = a.level[i];
  float r = (double)a.viewCos[i] > 0.998 ? 2.5f : 4.0f;
  if ((double)a.th != 1.0) r = __fmul_rn(r, a.th);
  const float radius = __fmul_rn(r, a.scale[nPredictedLevel]);
  const float xr = a.uRight ? a.projXR[i] : 0.0f;
  wave_gather_window(
      a.grid, a.proj[2 * i], a.proj[2 * i + 1], radius, nPredictedLevel - 1, nPredictedLevel,
      a.ptDesc + (size_t)i * 8, a.desc,
      [&](int i2) {
        if (!a.uRight) return true;
        const float r2 = a.uRight[i2];
        return !(r2 > 0 && fabsf(__fsub_rn(xr, r2)) > radius);
      },
      [](int dist, int octave, int i2) { return ((uint32_t)dist << 20) | ((uint32_t)octave << 16) | (uint32_t)i2; },
      a.cand, (size_t)a.nPts, i, &a.candCount[i], colS[w], colP[w]);
}

__device__ __forceinline__ void points_rounds(const PointsArgs& a, int* choice0, int* choice1, int* closed0,
                                              int* closed1, int* nmS) {
  const int tid = threadIdx.x;
  const int OPEN = 0x7fffffff;
  for (int i = tid; i < a.nPts; i += GUIDED_THREADS) choice0[i] = -1;
  for (int i2 = tid; i2 < a.n; i2 += GUIDED_THREADS)
    closed0[i2] = (a.state && a.state[i2] == 1) ? -1 : OPEN;
  if (tid == 0) *nmS = 0;
  __syncthreads();
  int cur = 0, rounds = 0;
  for (;;) {
    const int* cPrev = cur ? choice1 : choice0;
    int* cNew = cur ? choice0 : choice1;
    const int* bPrev = cur ? closed1 : closed0;
    int* bNew = cur ? closed0 : closed1;
    int changed = 0;
    for (int i = tid; i < a.nPts; i += GUIDED_THREADS) {
      int c;
      if (rounds == 0 && !a.gathered) {
        c = points_choice(a, i, bPrev, true);
      } else {
        const int cnt = a.candCount[i];
        c = cnt >= 0 ? points_choice_cached(a, i, bPrev, cnt) : points_choice(a, i, bPrev, false);
      }
      cNew[i] = c;
      changed |= (c != cPrev[i]) ? 1 : 0;
    }
    for (int i2 = tid; i2 < a.n; i2 += GUIDED_THREADS)
      bNew[i2] = (a.state && a.state[i2] == 1) ? -1 : OPEN;
    __syncthreads();
    for (int i = tid; i < a.nPts; i += GUIDED_THREADS) {
      const int c = cNew[i];
      if (c >= 0 && a.ptObs[i]) atomicMin(&bNew[c], i);
    }
    ++rounds;
    cur ^= 1;
    const int any = __syncthreads_or(changed);
    if (!any || rounds > a.nPts + 1) break;
  }
  // matcher.cc:127-138: the last map point that took a keypoint keeps it; every take counts
  const int* choice = cur ? choice1 : choice0;
  for (int i2 = tid; i2 < a.n; i2 += GUIDED_THREADS) a.assigned[i2] = -1;
  __syncthreads();
  for (int i = tid; i < a.nPts; i += GUIDED_THREADS) {
    const int c = choice[i];
    if (c >= 0) {
      atomicMax(&a.assigned[c], i);
      atomicAdd(nmS, 1);
    }
  }
  __syncthreads();
  if (tid == 0) { a.summary[0] = *nmS; a.summary[1] = rounds; }
}

__global__ __launch_bounds__(GUIDED_THREADS) void k_search_by_projection_points(PointsArgs a) {
  __shared__ int nmS;
  __shared__ int gen[4 * ROUNDS_LDS_MAX];
  // (as in k_search_by_projection_last: the generations in LDS when the frame is of usual size)
  if (a.nPts <= ROUNDS_LDS_MAX && a.n <= ROUNDS_LDS_MAX)
    points_rounds(a, gen, gen + ROUNDS_LDS_MAX, gen + 2 * ROUNDS_LDS_MAX, gen + 3 * ROUNDS_LDS_MAX, &nmS);
  else
    points_rounds(a, a.choice[0], a.choice[1], a.closedAt[0], a.closedAt[1], &nmS);
}

// ORBmatcher::SearchByProjection(F, vpMapPoints, th) for a frame with two cameras (F.Nleft != -1,
// matcher.cc:41-213).  Map point i takes two steps, the left camera's search at time 2i and the
// right camera's at time 2i + 1; a step that matches keypoint c writes the map point into c and
// into c's stereo partner in the other camera (:129-135, :197-203), whatever either held before.
// A keypoint is closed to a step at time t when the LATEST write before t put a map point with
// observations there (no write yet: its initial state) - a partner write of a point without
// observations re-opens a keypoint, so a closing time per keypoint is not enough: every keypoint
// keeps the list of the writes of the previous round, and a step looks for the latest one before
// its own time.  Rounds repeat until no step changes its choice; a step at time t depends only on
// steps before t, so round r has every step before time r right and the fixed point is the
// sequential loop's result.
//   write node 4i + k:  k = 0 left keypoint, by the left step;   1 its partner in the right camera;
//                       k = 2 right keypoint, by the right step;  3 its partner in the left camera
struct RigPointsArgs {
  PointsArgs cam[2];        // per camera; cam[1].inView: in the right view AND with a predicted level
                            // there (:144-146), cam[1].th = 1 (the right radius ignores th, :147)
  const int* partner[2];    // [0] mvLeftToRightMatch (nL entries), [1] mvRightToLeftMatch (nR); -1 none
  int* choice[2][2];        // [generation][camera], nPts each: keypoint, -1 none, -2 ratio test failed
  int* head[2];             // per camera, n each: latest pushed write node, -1 none
  int* next;                // 4 nPts
  int* assigned[2];
  int* summary;             // [0] nmatches, [1] rounds
};

__device__ __forceinline__ int rig_node_time(int node) { return 2 * (node >> 2) + ((node & 3) >> 1); }

// whether keypoint k of a camera holds a map point with observations just before time t
__device__ __forceinline__ bool rig_closed(const int* head, const int* next, const uint8_t* state,
                                           const uint8_t* ptObs, int k, int t) {
  int latest = -1, latestT = -1;
  for (int node = coherent_load(&head[k]); node >= 0; node = coherent_load(&next[node])) {
    const int tn = rig_node_time(node);
    if (tn < t && tn > latestT) { latestT = tn; latest = node; }
  }
  if (latest < 0) return state && state[k] == 1;
  return ptObs[latest >> 2] != 0;
}

__device__ __forceinline__ void rig_push(int* head, int* next, int k, int node) {
  next[node] = atomicExch(&head[k], node);
}

__global__ __launch_bounds__(GUIDED_THREADS) void k_search_by_projection_points_rig(RigPointsArgs a) {
  __shared__ int nmS;
  const int tid = threadIdx.x;
  const int nPts = a.cam[0].nPts;
  const uint8_t* ptObs = a.cam[0].ptObs;
  for (int i = tid; i < nPts; i += GUIDED_THREADS) { a.choice[0][0][i] = -1; a.choice[0][1][i] = -1; }
  if (tid == 0) nmS = 0;
  __syncthreads();
  int cur = 0, rounds = 0;
  for (;;) {
    // the writes of the previous round's choices, per keypoint
    for (int cam = 0; cam < 2; ++cam)
      for (int k = tid; k < a.cam[cam].n; k += GUIDED_THREADS) a.head[cam][k] = -1;
    __syncthreads();
    for (int i = tid; i < nPts; i += GUIDED_THREADS) {
      const int cl = coherent_load(&a.choice[cur][0][i]), cr = coherent_load(&a.choice[cur][1][i]);
      if (cl >= 0) {
        rig_push(a.head[0], a.next, cl, 4 * i);
        const int p = a.partner[0][cl];
        if (p >= 0) rig_push(a.head[1], a.next, p, 4 * i + 1);
      }
      if (cr >= 0) {
        rig_push(a.head[1], a.next, cr, 4 * i + 2);
        const int p = a.partner[1][cr];
        if (p >= 0) rig_push(a.head[0], a.next, p, 4 * i + 3);
      }
    }
    __syncthreads();
    int changed = 0;
    for (int i = tid; i < nPts; i += GUIDED_THREADS) {
      int c[2];
      for (int cam = 0; cam < 2; ++cam) {
        const PointsArgs& pa = a.cam[cam];
        if (cam == 1 && c[0] == -2) { c[1] = -1; break; }   // :123-124 skips the right camera too
        const int t = 2 * i + cam;
        auto closed = [&](int i2) { return rig_closed(a.head[cam], a.next, pa.state, ptObs, i2, t); };
        const int cnt = pa.candCount[i];
        const BestTwo b = cnt >= 0 ? points_walk_cached(pa, i, closed, cnt) : points_walk(pa, i, closed, false, true);
        c[cam] = b.accepted3(pa.nnratio);
      }
      for (int cam = 0; cam < 2; ++cam) {
        changed |= (c[cam] != coherent_load(&a.choice[cur][cam][i])) ? 1 : 0;
        a.choice[cur ^ 1][cam][i] = c[cam];
      }
    }
    ++rounds;
    cur ^= 1;
    const int any = __syncthreads_or(changed);
    if (!any || rounds > 2 * nPts + 2) break;
  }
  // nothing changed in the last round: the lists hold exactly the final choices' writes.  A
  // keypoint keeps the map point of its latest write; every write counts (:132-138, :200-207)
  for (int cam = 0; cam < 2; ++cam)
    for (int k = tid; k < a.cam[cam].n; k += GUIDED_THREADS) {
      int latest = -1, latestT = -1, cnt = 0;
      for (int node = coherent_load(&a.head[cam][k]); node >= 0; node = coherent_load(&a.next[node])) {
        const int tn = rig_node_time(node);
        if (tn > latestT) { latestT = tn; latest = node; }
        ++cnt;
      }
      a.assigned[cam][k] = latest >= 0 ? (latest >> 2) : -1;
      if (cnt) atomicAdd(&nmS, cnt);
    }
  __syncthreads();
  if (tid == 0) { a.summary[0] = nmS; a.summary[1] = rounds; }
}

struct InitArgs {
  // frame 1 (reference frame of the initialisation)
  const GKP* K1;
  const uint32_t* desc1;
  int n1;
  float* prevMatched;      // n1 (x, y) pairs, in / out (matcher.cc:728-730)
  // frame 2
  GridView grid;
  const uint32_t* desc2;
  int n2;
  float window, nnratio;
  int checkOrientation;
  // work arrays: two generations of claims (keypoint, distance) per query of frame 1;
  // claimants of a keypoint as a linked list (head per keypoint, next per query)
  int* claim[2];
  int* claimDist[2];
  int* head;               // n2; reused as owner at the end
  int* next;               // n1
  uint32_t* cand;          // [GUIDED_CACHE][n1], as above
  int* candCount;          // n1
  // results
  int* matches12;          // n1
  int* summary;            // [0] nmatches, [1] rounds
  int gathered;            // as in ProjectionArgs
};

// vMatchedDistance[i2] as query i1 meets it: the smallest distance accepted by an earlier query
__device__ __forceinline__ int init_seen(const InitArgs& a, int i1, int i2, const int* claimDistPrev) {
  int seen = 0x7fffffff;
  for (int j = coherent_load(&a.head[i2]); j >= 0; j = a.next[j])
    if (j < i1) seen = min(seen, claimDistPrev[j]);
  return seen;
}

// claim of query i1 given the claims of the previous round (matcher.cc:638-704).
// mode 0: walk the window and remember the candidates; 1: walk only; 2: use the remembered ones;
// 3: walk and remember without deciding (the claims are not read).
__device__ void init_choice(const InitArgs& a, int i1, const int* claimDistPrev, int mode, int* kpOut, int* distOut) {
  *kpOut = -1;
  *distOut = 0x7fffffff;
  int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
  auto consider = [&](int i2, int dist) {
    if (init_seen(a, i1, i2, claimDistPrev) <= dist) return;
    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
    else if (dist < bestDist2) { bestDist2 = dist; }
  };
  if (mode == 2) {
    const int cnt = a.candCount[i1];
    for (int k = 0; k < cnt; ++k) {
      const uint32_t e = a.cand[(size_t)k * a.n1 + i1];
      consider((int)(e & 0xFFFFu), (int)(e >> 16));
    }
  } else {
    const bool store = mode == 0 || mode == 3;
    if (store) a.candCount[i1] = 0;
    const GKP kp1 = a.K1[i1];
    const int level1 = kp1.octave;
    if (level1 > 0) return;
    uint32_t d1[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) d1[w] = a.desc1[(size_t)i1 * 8 + w];
    int k = 0;
    for_each_in_area(a.grid, a.prevMatched[2 * i1], a.prevMatched[2 * i1 + 1], a.window, level1, level1, [&](int i2) {
      const int dist = hamming_reg(d1, a.desc2 + (size_t)i2 * 8);
      if (store) {
        if (k < GUIDED_CACHE) a.cand[(size_t)k * a.n1 + i1] = ((uint32_t)dist << 16) | (uint32_t)i2;
        ++k;
      }
      if (mode != 3) consider(i2, dist);
    });
    if (store) a.candCount[i1] = k <= GUIDED_CACHE ? k : -1;
  }
  if (bestDist <= TH_LOW && (float)bestDist < __fmul_rn((float)bestDist2, a.nnratio)) {
    *kpOut = bestIdx2;
    *distOut = bestDist;
  }
}

__global__ __launch_bounds__(GATHER_THREADS) void k_gather_init(InitArgs a) {
  __shared__ int colS[GATHER_WAVES][64], colP[GATHER_WAVES][64];
  const int w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
  const int i = (int)blockIdx.x * GATHER_WAVES + w;
  if (i >= a.n1) return;
  const int level1 = a.K1[i].octave;
  if (level1 > 0) {   // matcher.cc:640-642: only keypoints of the finest level are searched
    if (lane == 0) a.candCount[i] = 0;
    return;
  }
  wave_gather_window(
      a.grid, a.prevMatched[2 * i], a.prevMatched[2 * i + 1], a.window, level1, level1, a.desc1 + (size_t)i * 8,
      a.desc2, [](int) { return true; }, [](int dist, int, int i2) { return ((uint32_t)dist << 16) | (uint32_t)i2; },
      a.cand, (size_t)a.n1, i, &a.candCount[i], colS[w], colP[w]);
}

__global__ __launch_bounds__(GUIDED_THREADS) void k_search_for_initialization(InitArgs a) {
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int keep[3];
  __shared__ int nmS;
  const int tid = threadIdx.x;
  for (int i = tid; i < a.n1; i += GUIDED_THREADS) { a.claim[0][i] = -1; a.claimDist[0][i] = 0x7fffffff; }
  if (tid < HISTO_LENGTH) hist[tid] = 0;
  if (tid == 0) nmS = 0;
  __syncthreads();
  int cur = 0, rounds = 0;
  for (;;) {
    const int* cPrev = a.claim[cur];
    const int* dPrev = a.claimDist[cur];
    int* cNew = a.claim[cur ^ 1];
    int* dNew = a.claimDist[cur ^ 1];
    for (int i2 = tid; i2 < a.n2; i2 += GUIDED_THREADS) a.head[i2] = -1;
    __syncthreads();
    for (int i = tid; i < a.n1; i += GUIDED_THREADS) {
      const int c = cPrev[i];
      if (c >= 0) a.next[i] = atomicExch(&a.head[c], i);
    }
    __syncthreads();
    int changed = 0;
    for (int i = tid; i < a.n1; i += GUIDED_THREADS) {
      int c, d;
      init_choice(a, i, dPrev, (rounds == 0 && !a.gathered) ? 0 : (a.candCount[i] >= 0 ? 2 : 1), &c, &d);
      cNew[i] = c;
      dNew[i] = d;
      changed |= (c != cPrev[i]) ? 1 : 0;
    }
    ++rounds;
    cur ^= 1;
    const int any = __syncthreads_or(changed);
    if (!any || rounds > a.n1 + 1) break;
  }
  // matcher.cc:690-702: the last query that claimed a keypoint owns it (earlier owners were
  // reset to -1); every accepted claim entered the rotation histogram
  const int* claim = a.claim[cur];
  int* owner = a.head;
  for (int i2 = tid; i2 < a.n2; i2 += GUIDED_THREADS) owner[i2] = -1;
  __syncthreads();
  for (int i = tid; i < a.n1; i += GUIDED_THREADS) {
    const int c = claim[i];
    if (c >= 0) {
      atomicMax(&owner[c], i);
      if (a.checkOrientation) {
        const int bin = rotation_bin(a.K1[i].angle, a.grid.U[c].angle);
        if (bin >= 0 && bin < HISTO_LENGTH) atomicAdd(&hist[bin], 1);
      }
    }
  }
  __syncthreads();
  if (a.checkOrientation && tid == 0) three_maxima(hist, HISTO_LENGTH, keep);
  __syncthreads();
  for (int i = tid; i < a.n1; i += GUIDED_THREADS) {
    const int c = claim[i];
    int m = (c >= 0 && coherent_load(&owner[c]) == i) ? c : -1;
    if (m >= 0 && a.checkOrientation) {
      // matcher.cc:708-725
      const int bin = rotation_bin(a.K1[i].angle, a.grid.U[c].angle);
      if (bin >= 0 && bin < HISTO_LENGTH && bin != keep[0] && bin != keep[1] && bin != keep[2]) m = -1;
    }
    a.matches12[i] = m;
    if (m >= 0) {
      atomicAdd(&nmS, 1);
      a.prevMatched[2 * i] = a.grid.U[m].x;       // matcher.cc:728-730
      a.prevMatched[2 * i + 1] = a.grid.U[m].y;
    }
  }
  __syncthreads();
  if (tid == 0) { a.summary[0] = nmS; a.summary[1] = rounds; }
}

// ---- ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches), matcher.cc:222-405 ----------------------
// A frame keypoint belongs to exactly one node of the frame's feature vector, so keyframe
// features of different nodes never compete: one wave per keyframe node.  Inside a node the
// reference is sequential over the keyframe's features (a frame keypoint that received a map
// point is skipped by the later ones, :274-275); the wave keeps that order and spreads the
// node's frame keypoints over its lanes.  The lane that owns a list position is the only one
// that reads or writes matches[] for it, so "already matched" needs no exchange.
struct BowMatchArgs {
  // keyframe: per keypoint, then its feature vector as CSR in ascending node id
  const uint8_t* kfValid;
  const float* kfAngle;
  const uint32_t* kfDesc;
  int nKfNodes;
  const uint32_t* kfNode;
  const int* kfStart;
  const uint32_t* kfFeat;
  // frame: keypoints (angle), descriptors, its feature vector where the transform left it
  const GKP* K;
  const uint32_t* desc;
  const int* nP;           // number of keypoints of the frame
  const uint32_t* fNode;
  const int* fStart;
  const uint32_t* fFeat;
  const int* fCount;       // number of nodes
  float nnratio;
  int checkOrientation;
  int* matches;            // n, preset to -1 by the launcher
  int* summary;            // [0] nmatches
};

#define BOWM_THREADS 256

__global__ __launch_bounds__(BOWM_THREADS) void k_bow_match_nodes(BowMatchArgs a) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (BOWM_THREADS / 64) + (threadIdx.x >> 6);   // keyframe node of this wave
  if (r >= a.nKfNodes) return;
  const uint32_t node = a.kfNode[r];
  // the same node in the frame's vector (ascending ids): lower bound
  int lo = 0, hi = *a.fCount;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a.fNode[mid] < node) lo = mid + 1; else hi = mid;
  }
  if (lo >= *a.fCount || a.fNode[lo] != node) return;
  const int fs = a.fStart[lo], nF = a.fStart[lo + 1] - fs;
  for (int iKF = a.kfStart[r]; iKF < a.kfStart[r + 1]; ++iKF) {
    const int realIdxKF = (int)a.kfFeat[iKF];
    if (!a.kfValid[realIdxKF]) continue;
    uint32_t dKF[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) dKF[w] = a.kfDesc[(size_t)realIdxKF * 8 + w];
    // per lane: smallest (distance << 20 | list position) and the second smallest distance of
    // its positions - what the scan of :270-288 leaves in bestDist1 / bestIdxF / bestDist2
    uint32_t k1 = (256u << 20) | 0xFFFFFu;
    int d2 = 256;
    for (int c = lane; c < nF; c += 64) {
      const int i2 = (int)a.fFeat[fs + c];
      if (a.matches[i2] >= 0) continue;
      const int dist = hamming_reg(dKF, a.desc + (size_t)i2 * 8);
      const uint32_t key = ((uint32_t)dist << 20) | (uint32_t)c;
      if (key < k1) { d2 = (int)(k1 >> 20); k1 = key; }
      else if (dist < d2) d2 = dist;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t ok1 = (uint32_t)__shfl_xor((int)k1, off);
      const int od2 = __shfl_xor(d2, off);
      const uint32_t hiKey = max(k1, ok1);
      k1 = min(k1, ok1);
      d2 = min(min(d2, od2), (int)(hiKey >> 20));
    }
    const int bestDist1 = (int)(k1 >> 20), pos = (int)(k1 & 0xFFFFFu);
    if (bestDist1 <= TH_LOW && (float)bestDist1 < __fmul_rn(a.nnratio, (float)d2)) {
      if ((pos & 63) == lane) a.matches[(int)a.fFeat[fs + pos]] = realIdxKF;   // the owner of that position
    }
  }
}

// rotation consistency (:387-402) and the count, one workgroup
__global__ __launch_bounds__(GUIDED_THREADS) void k_bow_match_finish(BowMatchArgs a) {
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int keep[3];
  __shared__ int nmS;
  const int tid = threadIdx.x, n = *a.nP;
  if (tid < HISTO_LENGTH) hist[tid] = 0;
  if (tid == 0) nmS = 0;
  __syncthreads();
  int mine = 0;
  for (int i2 = tid; i2 < n; i2 += GUIDED_THREADS) {
    const int m = a.matches[i2];
    if (m >= 0) {
      ++mine;
      if (a.checkOrientation) {
        const int bin = rotation_bin(a.kfAngle[m], a.K[i2].angle);
        if (bin >= 0 && bin < HISTO_LENGTH) atomicAdd(&hist[bin], 1);
      }
    }
  }
  if (mine) atomicAdd(&nmS, mine);
  __syncthreads();
  if (a.checkOrientation) {
    if (tid == 0) three_maxima(hist, HISTO_LENGTH, keep);
    __syncthreads();
    int cleared = 0;
    for (int i2 = tid; i2 < n; i2 += GUIDED_THREADS) {
      const int m = a.matches[i2];
      if (m >= 0) {
        const int bin = rotation_bin(a.kfAngle[m], a.K[i2].angle);
        if (bin >= 0 && bin < HISTO_LENGTH && bin != keep[0] && bin != keep[1] && bin != keep[2]) {
          a.matches[i2] = -2;
          ++cleared;
        }
      }
    }
    if (cleared) atomicSub(&nmS, cleared);
    __syncthreads();
  }
  if (tid == 0) a.summary[0] = nmS;
}

static GridView make_grid(const float* d_kpsUn, const int* d_cellStart, const int* d_entries, const float* bounds) {
  GridView g;
  g.U = (const GKP*)d_kpsUn;
  g.cs = d_cellStart;
  g.ent = d_entries;
  g.minX = bounds[0]; g.maxX = bounds[1]; g.minY = bounds[2]; g.maxY = bounds[3];
  g.invW = (float)GRID_COLS / (bounds[1] - bounds[0]);   // Frame.cc:193-196
  g.invH = (float)GRID_ROWS / (bounds[3] - bounds[2]);
  return g;
}

// Whether the window walks get a launch of their own, one wave per query (ORBX_GUIDED_GATHER=0
// leaves them to the threads of the rounds kernel, one query per thread and round trip after
// round trip: the same results, several times the latency; kept as the path a query with more
// candidates than its list holds takes anyway, and tested through the knob).
static int gather_first(int nQueries) {
  (void)nQueries;
  // (read at every call, so that a test can switch it inside one process)
  const char* e = getenv("ORBX_GUIDED_GATHER");
  if (e && e[0]) return atoi(e) != 0 ? 1 : 0;
  return 1;
}

// work memory either launcher needs beyond its inputs and outputs, in ints
size_t projection_work_ints(int nLast, int n) { return (3 + GUIDED_CACHE) * (size_t)nLast + 2 * (size_t)n; }
size_t init_work_ints(int n1, int n2) { return (6 + GUIDED_CACHE) * (size_t)n1 + (size_t)n2; }

void launch_search_by_projection_last(const uint8_t* d_lastValid, const float* d_uv, const float* d_invz,
                                      const int* d_lastOctave, const float* d_lastAngle, const uint8_t* d_lastDesc,
                                      const uint8_t* d_lastObs, int nLast, const float* d_kpsUn,
                                      const uint8_t* d_desc, const float* d_uRight, const uint8_t* d_state, int n,
                                      const int* d_cellStart, const int* d_entries, const float* bounds,
                                      const float* scaleFactors, int nlevels, float th, int forward, int backward,
                                      float mbf, int checkOrientation, int maxDist, int* d_work, int* d_assigned,
                                      int* d_summary, hipStream_t s) {
  ProjectionArgs a;
  a.maxDist = maxDist;
  a.lastValid = d_lastValid; a.uv = d_uv; a.invz = d_invz; a.lastOctave = d_lastOctave;
  a.lastAngle = d_lastAngle; a.lastDesc = (const uint32_t*)d_lastDesc; a.lastObs = d_lastObs; a.nLast = nLast;
  a.grid = make_grid(d_kpsUn, d_cellStart, d_entries, bounds);
  a.desc = (const uint32_t*)d_desc; a.uRight = d_uRight; a.state = d_state; a.n = n;
  for (int l = 0; l < ORBX_MAX_LEVELS; ++l) a.scale[l] = l < nlevels ? scaleFactors[l] : 1.0f;
  a.th = th; a.mbf = mbf; a.forward = forward; a.backward = backward; a.checkOrientation = checkOrientation;
  a.choice[0] = d_work; a.choice[1] = d_work + nLast;
  a.closedAt[0] = d_work + 2 * (size_t)nLast; a.closedAt[1] = a.closedAt[0] + n;
  a.candCount = a.closedAt[1] + n;
  a.cand = (uint32_t*)(a.candCount + nLast);
  a.assigned = d_assigned; a.summary = d_summary;
  a.gate = nullptr; a.skipBounds = 0; a.deferRotation = 0; a.finalChoice = nullptr;
  a.gathered = gather_first(nLast);
  if (a.gathered)
    hipLaunchKernelGGL(k_gather_projection, dim3((nLast + GATHER_WAVES - 1) / GATHER_WAVES), dim3(GATHER_THREADS), 0,
                       s, a);
  hipLaunchKernelGGL(k_search_by_projection_last, dim3(1), dim3(GUIDED_THREADS), 0, s, a);
}

// The two cameras of a rig (matcher.cc:1591-1808 with CurrentFrame.Nleft != -1): the left camera's
// search, the right camera's search gated by the left one's candidate counts, and one rotation
// check over both.  d_workL / d_workR: projection_work_ints(nLast, nL / nR) + nLast ints each
// (the extra nLast hold the final choices); d_sumL / d_sumR: 2 ints each.
void launch_search_by_projection_last_rig(const uint8_t* d_lastValid, const float* d_uvL, const float* d_invz,
                                          const float* d_uvR, const int* d_lastOctave, const float* d_lastAngle,
                                          const uint8_t* d_lastDesc, const uint8_t* d_lastObs, int nLast,
                                          const float* d_kpsL, const uint8_t* d_descL, const uint8_t* d_stateL, int nL,
                                          const int* d_cellStartL, const int* d_entriesL, const float* d_kpsR,
                                          const uint8_t* d_descR, const uint8_t* d_stateR, int nR,
                                          const int* d_cellStartR, const int* d_entriesR, const float* bounds,
                                          const float* scaleFactors, int nlevels, float th, int forward, int backward,
                                          int checkOrientation, int* d_workL, int* d_workR, int* d_assignedL,
                                          int* d_assignedR, int* d_sumL, int* d_sumR, int* d_summary, hipStream_t s) {
  ProjectionArgs a;
  a.maxDist = TH_HIGH;
  a.lastValid = d_lastValid; a.invz = d_invz; a.lastOctave = d_lastOctave;
  a.lastAngle = d_lastAngle; a.lastDesc = (const uint32_t*)d_lastDesc; a.lastObs = d_lastObs; a.nLast = nLast;
  a.uRight = nullptr;   // the mvuRight test needs a frame with one camera (:1669)
  for (int l = 0; l < ORBX_MAX_LEVELS; ++l) a.scale[l] = l < nlevels ? scaleFactors[l] : 1.0f;
  a.th = th; a.mbf = 0.0f; a.forward = forward; a.backward = backward; a.checkOrientation = checkOrientation;
  a.deferRotation = 1;
  a.gathered = gather_first(nLast);
  const int* candCountL = nullptr;
  for (int cam = 0; cam < 2; ++cam) {
    int* w = cam == 0 ? d_workL : d_workR;
    const int n = cam == 0 ? nL : nR;
    a.uv = cam == 0 ? d_uvL : d_uvR;
    a.grid = cam == 0 ? make_grid(d_kpsL, d_cellStartL, d_entriesL, bounds) : make_grid(d_kpsR, d_cellStartR, d_entriesR, bounds);
    a.desc = (const uint32_t*)(cam == 0 ? d_descL : d_descR);
    a.state = cam == 0 ? d_stateL : d_stateR;
    a.n = n;
    a.choice[0] = w; a.choice[1] = w + nLast;
    a.closedAt[0] = w + 2 * (size_t)nLast; a.closedAt[1] = a.closedAt[0] + n;
    a.candCount = a.closedAt[1] + n;
    a.cand = (uint32_t*)(a.candCount + nLast);
    a.finalChoice = w + projection_work_ints(nLast, n);
    a.assigned = cam == 0 ? d_assignedL : d_assignedR;
    a.summary = cam == 0 ? d_sumL : d_sumR;
    a.gate = cam == 0 ? nullptr : candCountL;
    a.skipBounds = cam;
    if (cam == 0) candCountL = a.candCount;
    if (a.gathered)
      hipLaunchKernelGGL(k_gather_projection, dim3((nLast + GATHER_WAVES - 1) / GATHER_WAVES), dim3(GATHER_THREADS),
                         0, s, a);
    hipLaunchKernelGGL(k_search_by_projection_last, dim3(1), dim3(GUIDED_THREADS), 0, s, a);
  }
  RigFinishArgs f;
  f.nLast = nLast;
  f.choiceL = d_workL + projection_work_ints(nLast, nL);
  f.choiceR = d_workR + projection_work_ints(nLast, nR);
  f.lastAngle = d_lastAngle;
  f.UL = (const GKP*)d_kpsL; f.UR = (const GKP*)d_kpsR;
  f.assignedL = d_assignedL; f.assignedR = d_assignedR;
  f.summaryL = d_sumL; f.summaryR = d_sumR; f.summary = d_summary;
  f.checkOrientation = checkOrientation;
  hipLaunchKernelGGL(k_rig_rotation_finish, dim3(1), dim3(GUIDED_THREADS), 0, s, f);
}

void launch_search_by_projection_points(const uint8_t* d_inView, const float* d_proj, const float* d_projXR,
                                        const int* d_level, const float* d_viewCos, const uint8_t* d_ptDesc,
                                        const uint8_t* d_ptObs, int nPts, const float* d_kpsUn, const uint8_t* d_desc,
                                        const float* d_uRight, const uint8_t* d_state, int n, const int* d_cellStart,
                                        const int* d_entries, const float* bounds, const float* scaleFactors,
                                        int nlevels, float th, float nnratio, int* d_work, int* d_assigned,
                                        int* d_summary, hipStream_t s) {
  PointsArgs a;
  a.inView = d_inView; a.proj = d_proj; a.projXR = d_projXR; a.level = d_level; a.viewCos = d_viewCos;
  a.ptDesc = (const uint32_t*)d_ptDesc; a.ptObs = d_ptObs; a.nPts = nPts;
  a.grid = make_grid(d_kpsUn, d_cellStart, d_entries, bounds);
  a.desc = (const uint32_t*)d_desc; a.uRight = d_uRight; a.state = d_state; a.n = n;
  for (int l = 0; l < ORBX_MAX_LEVELS; ++l) a.scale[l] = l < nlevels ? scaleFactors[l] : 1.0f;
  a.th = th; a.nnratio = nnratio;
  // same layout as the last-frame matcher (projection_work_ints)
  a.choice[0] = d_work; a.choice[1] = d_work + nPts;
  a.closedAt[0] = d_work + 2 * (size_t)nPts; a.closedAt[1] = a.closedAt[0] + n;
  a.candCount = a.closedAt[1] + n;
  a.cand = (uint32_t*)(a.candCount + nPts);
  a.assigned = d_assigned; a.summary = d_summary;
  a.gathered = gather_first(nPts);
  if (a.gathered)
    hipLaunchKernelGGL(k_gather_points, dim3((nPts + GATHER_WAVES - 1) / GATHER_WAVES), dim3(GATHER_THREADS), 0, s, a);
  hipLaunchKernelGGL(k_search_by_projection_points, dim3(1), dim3(GUIDED_THREADS), 0, s, a);
}

// work memory of the two-camera form, in ints: per camera the candidate lists and counts and two
// generations of choices, the write lists (heads per keypoint, 4 nodes per map point)
size_t points_rig_work_ints(int nPts, int nL, int nR) {
  return (size_t)nPts * (2 * (GUIDED_CACHE + 1) + 4 + 4) + (size_t)nL + (size_t)nR;
}

// d_inViewR: in the right camera's view and with a predicted level there (the caller folds
// mnTrackScaleLevelR != -1 in, so that no level of -1 reaches the scale table)
void launch_search_by_projection_points_rig(const uint8_t* d_inView, const float* d_proj, const int* d_level,
                                            const float* d_viewCos, const uint8_t* d_inViewR, const float* d_projR,
                                            const int* d_levelR, const float* d_viewCosR, const uint8_t* d_ptDesc,
                                            const uint8_t* d_ptObs, int nPts, const float* d_kpsL,
                                            const uint8_t* d_descL, const uint8_t* d_stateL, int nL,
                                            const int* d_cellStartL, const int* d_entriesL, const float* d_kpsR,
                                            const uint8_t* d_descR, const uint8_t* d_stateR, int nR,
                                            const int* d_cellStartR, const int* d_entriesR, const int* d_leftToRight,
                                            const int* d_rightToLeft, const float* bounds, const float* scaleFactors,
                                            int nlevels, float th, float nnratio, int* d_work, int* d_assignedL,
                                            int* d_assignedR, int* d_summary, hipStream_t s) {
  RigPointsArgs a;
  int* w = d_work;
  auto take = [&](size_t ints) { int* p = w; w += ints; return p; };
  for (int cam = 0; cam < 2; ++cam) {
    PointsArgs& p = a.cam[cam];
    p.inView = cam == 0 ? d_inView : d_inViewR;
    p.proj = cam == 0 ? d_proj : d_projR;
    p.projXR = nullptr;
    p.level = cam == 0 ? d_level : d_levelR;
    p.viewCos = cam == 0 ? d_viewCos : d_viewCosR;
    p.ptDesc = (const uint32_t*)d_ptDesc; p.ptObs = d_ptObs; p.nPts = nPts;
    p.grid = cam == 0 ? make_grid(d_kpsL, d_cellStartL, d_entriesL, bounds) : make_grid(d_kpsR, d_cellStartR, d_entriesR, bounds);
    p.desc = (const uint32_t*)(cam == 0 ? d_descL : d_descR);
    p.uRight = nullptr;   // the mvuRight test needs a frame with one camera (:93)
    p.state = cam == 0 ? d_stateL : d_stateR;
    p.n = cam == 0 ? nL : nR;
    for (int l = 0; l < ORBX_MAX_LEVELS; ++l) p.scale[l] = l < nlevels ? scaleFactors[l] : 1.0f;
    p.th = cam == 0 ? th : 1.0f;
    p.nnratio = nnratio;
    p.choice[0] = p.choice[1] = nullptr; p.closedAt[0] = p.closedAt[1] = nullptr;
    p.candCount = take(nPts);
    p.cand = (uint32_t*)take((size_t)GUIDED_CACHE * nPts);
    p.assigned = nullptr; p.summary = nullptr; p.gathered = 1;
    a.choice[0][cam] = take(nPts);
    a.choice[1][cam] = take(nPts);
    a.head[cam] = take(p.n);
    hipLaunchKernelGGL(k_gather_points, dim3((nPts + GATHER_WAVES - 1) / GATHER_WAVES), dim3(GATHER_THREADS), 0, s, p);
  }
  a.next = take(4 * (size_t)nPts);
  a.partner[0] = d_leftToRight; a.partner[1] = d_rightToLeft;
  a.assigned[0] = d_assignedL; a.assigned[1] = d_assignedR;
  a.summary = d_summary;
  hipLaunchKernelGGL(k_search_by_projection_points_rig, dim3(1), dim3(GUIDED_THREADS), 0, s, a);
}

void launch_search_by_bow(const uint8_t* d_kfValid, const float* d_kfAngle, const uint8_t* d_kfDesc, int nKfNodes,
                          const uint32_t* d_kfNode, const int* d_kfStart, const uint32_t* d_kfFeat,
                          const float* d_kps, const uint8_t* d_desc, const int* d_n, int cap, const uint32_t* d_fNode,
                          const int* d_fStart, const uint32_t* d_fFeat, const int* d_fCount, float nnratio,
                          int checkOrientation, int* d_matches, int* d_summary, hipStream_t s) {
  BowMatchArgs a;
  a.kfValid = d_kfValid; a.kfAngle = d_kfAngle; a.kfDesc = (const uint32_t*)d_kfDesc; a.nKfNodes = nKfNodes;
  a.kfNode = d_kfNode; a.kfStart = d_kfStart; a.kfFeat = d_kfFeat;
  a.K = (const GKP*)d_kps; a.desc = (const uint32_t*)d_desc; a.nP = d_n;
  a.fNode = d_fNode; a.fStart = d_fStart; a.fFeat = d_fFeat; a.fCount = d_fCount;
  a.nnratio = nnratio; a.checkOrientation = checkOrientation;
  a.matches = d_matches; a.summary = d_summary;
  (void)hipMemsetAsync(d_matches, 0xFF, (size_t)cap * 4, s);   // -1: no map point yet (:226)
  if (nKfNodes > 0) {
    const int perBlock = BOWM_THREADS / 64;
    hipLaunchKernelGGL(k_bow_match_nodes, dim3((nKfNodes + perBlock - 1) / perBlock), dim3(BOWM_THREADS), 0, s, a);
  }
  hipLaunchKernelGGL(k_bow_match_finish, dim3(1), dim3(GUIDED_THREADS), 0, s, a);
}

void launch_search_for_initialization(const float* d_kps1Un, const uint8_t* d_desc1, int n1, float* d_prevMatched,
                                      const float* d_kps2Un, const uint8_t* d_desc2, int n2,
                                      const int* d_cellStart2, const int* d_entries2, const float* bounds2,
                                      float window, float nnratio, int checkOrientation, int* d_work,
                                      int* d_matches12, int* d_summary, hipStream_t s) {
  InitArgs a;
  a.K1 = (const GKP*)d_kps1Un; a.desc1 = (const uint32_t*)d_desc1; a.n1 = n1; a.prevMatched = d_prevMatched;
  a.grid = make_grid(d_kps2Un, d_cellStart2, d_entries2, bounds2);
  a.desc2 = (const uint32_t*)d_desc2; a.n2 = n2;
  a.window = window; a.nnratio = nnratio; a.checkOrientation = checkOrientation;
  a.claim[0] = d_work; a.claim[1] = d_work + n1;
  a.claimDist[0] = d_work + 2 * (size_t)n1; a.claimDist[1] = d_work + 3 * (size_t)n1;
  a.next = d_work + 4 * (size_t)n1;
  a.candCount = d_work + 5 * (size_t)n1;
  a.cand = (uint32_t*)(d_work + 6 * (size_t)n1);
  a.head = d_work + (6 + GUIDED_CACHE) * (size_t)n1;
  a.matches12 = d_matches12; a.summary = d_summary;
  a.gathered = gather_first(n1);
  if (a.gathered)
    hipLaunchKernelGGL(k_gather_init, dim3((n1 + GATHER_WAVES - 1) / GATHER_WAVES), dim3(GATHER_THREADS), 0, s, a);
  hipLaunchKernelGGL(k_search_for_initialization, dim3(1), dim3(GUIDED_THREADS), 0, s, a);
}

}  // namespace orbx
