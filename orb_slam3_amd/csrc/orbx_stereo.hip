// Stereo association of two extractions that are still resident on the device:
// Frame::ComputeStereoMatches, core/src/Frame.cc:831-996.
//   k_stereo_rows         row table of the right keypoints       :840-856
//   k_stereo_band_batch   row-band descriptor search             :867-908
//   k_stereo_refine       11x11 L1 search over +-5 columns on the
//                         pyramid level of the left keypoint,
//                         parabola fit, disparity and depth      :910-981
//   k_stereo_outliers     median rule on the L1 distances        :984-995
// The pyramids never leave HBM (the reference reads mvImagePyramid of both
// extractors on the host).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "orbx_plan.h"

namespace orbx {

struct SKP { float x, y, size, angle, response; int octave, class_id; };

// Row table of the right keypoints (Frame.cc:840-856) as a CSR: rowStart[r] .. rowStart[r+1]
// index the right keypoints whose band floor(y - r) .. ceil(y + r), r = 2 * scale[octave],
// contains row r.  One block per frame; counts and cursors live in LDS.  The order inside a row
// is whatever the atomics produce: the search below takes the minimum of (distance, index),
// which is what "first strict minimum in ascending index" selects.
__global__ __launch_bounds__(256) void k_stereo_rows(const Plan* __restrict__ plan,
                                                     const SKP* __restrict__ kpsR,
                                                     const int* __restrict__ nRp, int capR,
                                                     int entriesPerFrame, int* __restrict__ rowStart,
                                                     int* __restrict__ entries) {
  extern __shared__ int sRows[];          // cnt[nRows] then partial[256]
  const int f = blockIdx.x, tid = threadIdx.x;
  const int nRows = plan->lv[0].h, nR = nRp[f];
  int* cnt = sRows;
  int* part = sRows + nRows;
  const SKP* kr = kpsR + (size_t)f * capR;
  for (int r = tid; r < nRows; r += 256) cnt[r] = 0;
  __syncthreads();
  for (int j = tid; j < nR; j += 256) {
    const float y = kr[j].y, rad = __fmul_rn(2.0f, plan->lv[kr[j].octave].scale);
    const int maxr = min((int)ceilf(__fadd_rn(y, rad)), nRows - 1);
    const int minr = max((int)floorf(__fsub_rn(y, rad)), 0);
    for (int r = minr; r <= maxr; ++r) atomicAdd(&cnt[r], 1);
  }
  __syncthreads();
  // exclusive scan: each thread owns a contiguous run of rows
  const int per = (nRows + 255) / 256;
  const int r0 = min(tid * per, nRows), r1 = min(r0 + per, nRows);
  int sum = 0;
  for (int r = r0; r < r1; ++r) sum += cnt[r];
  part[tid] = sum;
  __syncthreads();
  if (tid < 64) {   // wave 0 scans the 256 partial sums, 4 per lane
    int v[4], t = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = part[tid * 4 + u]; t += v[u]; }
    int inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d);
      if (tid >= d) inc += o;
    }
    int run = inc - t;
#pragma unroll
    for (int u = 0; u < 4; ++u) { part[tid * 4 + u] = run; run += v[u]; }
  }
  __syncthreads();
  int run = part[tid];
  int* rs = rowStart + (size_t)f * (nRows + 1);
  for (int r = r0; r < r1; ++r) {
    const int c = cnt[r];
    rs[r] = run;
    cnt[r] = run;       // becomes the fill cursor
    run += c;
  }
  if (tid == 255) rs[nRows] = run;   // threads past the last row own nothing: run = total
  __syncthreads();
  int* ent = entries + (size_t)f * entriesPerFrame;
  for (int j = tid; j < nR; j += 256) {
    const float y = kr[j].y, rad = __fmul_rn(2.0f, plan->lv[kr[j].octave].scale);
    const int maxr = min((int)ceilf(__fadd_rn(y, rad)), nRows - 1);
    const int minr = max((int)floorf(__fsub_rn(y, rad)), 0);
    for (int r = minr; r <= maxr; ++r) {
      const int pos = atomicAdd(&cnt[r], 1);
      if (pos < entriesPerFrame) ent[pos] = j;
    }
  }
}

// One lane per left keypoint over the candidates of its row (Frame.cc:867-908).
__global__ __launch_bounds__(256) void k_stereo_band_batch(
    const Plan* __restrict__ plan, const SKP* __restrict__ kpsL, const uint4* __restrict__ descL,
    const int* __restrict__ nLp, const SKP* __restrict__ kpsR, const uint4* __restrict__ descR,
    int capL, int capR, float minD, float maxD, int entriesPerFrame,
    const int* __restrict__ rowStart, const int* __restrict__ entries, int* __restrict__ bestIdx,
    int* __restrict__ bestDist) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int nL = nLp[f];
  if (i >= nL) return;
  const SKP* kr = kpsR + (size_t)f * capR;
  const uint4* dr = descR + (size_t)f * capR * 2;
  const int nRows = plan->lv[0].h;
  const SKP L = kpsL[(size_t)f * capL + i];
  const uint4 a0 = descL[((size_t)f * capL + i) * 2], a1 = descL[((size_t)f * capL + i) * 2 + 1];
  const int row = (int)L.y;
  const float minU = __fsub_rn(L.x, maxD), maxU = __fsub_rn(L.x, minD);
  const uint32_t none = 100u << 20;          // TH_HIGH, no index
  uint32_t best = none;
  if (row >= 0 && row < nRows && !(maxU < 0.f)) {
    const int* rs = rowStart + (size_t)f * (nRows + 1);
    const int* ent = entries + (size_t)f * entriesPerFrame;
    const int e0 = rs[row], e1 = min(rs[row + 1], entriesPerFrame);
    for (int e = e0; e < e1; ++e) {
      const int j = ent[e];
      const float xr = kr[j].x;
      const int octR = kr[j].octave;
      if (octR < L.octave - 1 || octR > L.octave + 1) continue;
      if (!(xr >= minU && xr <= maxU)) continue;
      const uint4 b0 = dr[(size_t)j * 2], b1 = dr[(size_t)j * 2 + 1];
      const int d = __builtin_popcount(a0.x ^ b0.x) + __builtin_popcount(a0.y ^ b0.y) +
                    __builtin_popcount(a0.z ^ b0.z) + __builtin_popcount(a0.w ^ b0.w) +
                    __builtin_popcount(a1.x ^ b1.x) + __builtin_popcount(a1.y ^ b1.y) +
                    __builtin_popcount(a1.z ^ b1.z) + __builtin_popcount(a1.w ^ b1.w);
      const uint32_t key = ((uint32_t)d << 20) | (uint32_t)j;
      best = key < best ? key : best;
    }
  }
  bestIdx[(size_t)f * capL + i] = best < none ? (int)(best & 0xFFFFF) : -1;
  bestDist[(size_t)f * capL + i] = best < none ? (int)(best >> 20) : 100;
}

// 16 lanes per left keypoint, 4 keypoints per wave: lane j < 11 evaluates the shift incR = j - 5.
__global__ __launch_bounds__(256) void k_stereo_refine(
    const Plan* __restrict__ plan, const uint8_t* __restrict__ pyrL, const uint8_t* __restrict__ pyrR,
    const SKP* __restrict__ kpsL, const int* __restrict__ nLp, const SKP* __restrict__ kpsR,
    int capL, int capR, const int* __restrict__ bandIdx, const int* __restrict__ bandDist,
    float mbf, float minD, float maxD, float* __restrict__ uRight, float* __restrict__ depth,
    int* __restrict__ sadDist) {
  const int f = blockIdx.y;
  const int nL = nLp[f];
  const int lane = threadIdx.x & 63, j = lane & 15;
  const int k = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  if ((blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * 4 >= nL) return;
  const bool valid = k < nL;
  const size_t o = (size_t)f * capL + (valid ? k : 0);
  const int bd = bandDist[o], bidx = bandIdx[o];
  bool go = valid && bd < (100 + 50) / 2 && bidx >= 0;       // thOrbDist, :835,911
  const SKP kpL = kpsL[o];
  const float uR0 = kpsR[(size_t)f * capR + (go ? bidx : 0)].x;
  const int lvl = go ? kpL.octave : 0;
  const LevelDesc& Lv = plan->lv[lvl];
  const float inv = __fdiv_rn(1.0f, Lv.scale);               // mvInvScaleFactors[octave]
  const float scaleduL = roundf(__fmul_rn(kpL.x, inv));
  const float scaledvL = roundf(__fmul_rn(kpL.y, inv));
  const float scaleduR0 = roundf(__fmul_rn(uR0, inv));
  const int w = 5;
  const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
  const float iniu = scaleduR0, endu = __fadd_rn(scaleduR0, 11.0f);   // + L - w, + L + w + 1
  go = go && cv - w >= 0 && cv + w + 1 <= Lv.h && cu - w >= 0 && cu + w + 1 <= Lv.w;
  go = go && !(iniu < 0.f || endu >= (float)Lv.w) && cr - 2 * w >= 0;
  int sum = 0x7FFFFFF;
  if (go && j < 11) {
    const int incR = j - 5;
    const uint8_t* a = pyrL + Lv.pyr_off + (size_t)f * Lv.pyr_plane +
                       (size_t)(cv - w + ORBX_EDGE) * Lv.pstride + (cu - w + ORBX_EDGE);
    const uint8_t* b = pyrR + Lv.pyr_off + (size_t)f * Lv.pyr_plane +
                       (size_t)(cv - w + ORBX_EDGE) * Lv.pstride + (cr + incR - w + ORBX_EDGE);
    sum = 0;
    for (int dy = 0; dy < 11; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 11; ++dx) {
        const int d = (int)a[dx] - (int)b[dx];
        sum += d < 0 ? -d : d;
      }
      a += Lv.pstride;
      b += Lv.pstride;
    }
  }
  // first strict minimum in ascending incR (:944-947): minimum of (distance, lane)
  int key = (sum << 4) | j;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    const int other = __shfl_xor(key, m);
    key = other < key ? other : key;
  }
  const int bj = key & 15, bestSad = key >> 4;
  const int gbase = lane & ~15;
  const int s1 = __shfl(sum, gbase + ((bj + 15) & 15));
  const int s3 = __shfl(sum, gbase + ((bj + 1) & 15));
  if (j != 0 || !valid) return;
  float ur = -1.0f, dp = -1.0f;
  int sd = -1;
  const int bestincR = bj - 5;
  if (go && bestincR != -5 && bestincR != 5) {
    const float dist1 = (float)s1, dist2 = (float)bestSad, dist3 = (float)s3;
    const float num = __fsub_rn(dist1, dist3);
    const float den = __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2)));
    const float deltaR = __fdiv_rn(num, den);
    if (!(deltaR < -1.f || deltaR > 1.f)) {
      float bestuR = __fmul_rn(Lv.scale, __fadd_rn(__fadd_rn(scaleduR0, (float)bestincR), deltaR));
      float disparity = __fsub_rn(kpL.x, bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0.f) {
          disparity = (float)0.01;
          bestuR = (float)((double)kpL.x - 0.01);
        }
        dp = __fdiv_rn(mbf, disparity);
        ur = bestuR;
        sd = bestSad;
      }
    }
  }
  uRight[o] = ur;
  depth[o] = dp;
  sadDist[o] = sd;
}

// One block per frame.  The reference sorts (distance, index) pairs and takes the distance at
// position size/2; only that value matters, and it is found by a bitwise selection over the
// 15-bit distances (121 * 255 < 2^15).  Matches with distance >= 1.5f * 1.4f * median are erased.
__global__ __launch_bounds__(256) void k_stereo_outliers(const int* __restrict__ nLp, int capL,
                                                         const int* __restrict__ sadDist,
                                                         float* __restrict__ uRight,
                                                         float* __restrict__ depth) {
  __shared__ int sCount[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int n = nLp[f];
  const int* sd = sadDist + (size_t)f * capL;
  int m = 0;
  for (int i = tid; i < n; i += 256) m += sd[i] >= 0;
  auto blockSum = [&](int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    __syncthreads();
    if ((tid & 63) == 0) sCount[tid >> 6] = v;
    __syncthreads();
    return sCount[0] + sCount[1] + sCount[2] + sCount[3];
  };
  m = blockSum(m);
  if (m == 0) return;
  int rank = m / 2, prefix = 0;
  for (int bit = 14; bit >= 0; --bit) {
    const int want = prefix << 1;   // candidates continue with a 0 at this bit
    int c = 0;
    for (int i = tid; i < n; i += 256) {
      const int v = sd[i];
      c += v >= 0 && (v >> bit) == want;
    }
    c = blockSum(c);
    if (rank < c) prefix = want;
    else { rank -= c; prefix = want | 1; }
  }
  const float median = (float)prefix;
  const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), median);
  for (int i = tid; i < n; i += 256) {
    const int v = sd[i];
    if (v >= 0 && !((float)v < thDist)) {
      uRight[(size_t)f * capL + i] = -1.0f;
      depth[(size_t)f * capL + i] = -1.0f;
    }
  }
}

// Row-table entries one right keypoint can produce: band = floor(y - r) .. ceil(y + r), r = 2 * scale.
int stereo_entries_per_keypoint(const Plan& hp) {
  return (int)(4.0f * hp.lv[hp.nlevels - 1].scale) + 3;
}

// The association's results for the first frame into page-locked host memory the GPU writes
// directly: the keypoint count, then uRight and depth of the first min(n, cap) keypoints.
__global__ __launch_bounds__(256) void k_stereo_export_one(const int* __restrict__ nL, const float* __restrict__ uRight,
                                                           const float* __restrict__ depth, int cap,
                                                           int* __restrict__ hcnt, float* __restrict__ hu,
                                                           float* __restrict__ hd) {
  const int nAll = nL[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) hcnt[0] = nAll;
  const int n = nAll < cap ? nAll : cap;
  for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < n; i += (int)gridDim.x * 256) {
    hu[i] = uRight[i];
    hd[i] = depth[i];
  }
}

void launch_stereo_export_one(const int* nL, const float* uRight, const float* depth, int cap, int* hcnt, float* hu,
                              float* hd, hipStream_t s) {
  const int blocks = cap / 256 + 1;
  hipLaunchKernelGGL(k_stereo_export_one, dim3(blocks < 16 ? blocks : 16), dim3(256), 0, s, nL, uRight, depth, cap, hcnt,
                     hu, hd);
}

void launch_stereo(const Plan& hp, const Plan* dp, const uint8_t* pyrL, const uint8_t* pyrR,
                   const float* kpsL, const uint8_t* descL, const int* nL, int capL,
                   const float* kpsR, const uint8_t* descR, const int* nR, int capR, float mbf,
                   float mb, int* rowStart, int* entries, int* bandIdx, int* bandDist, int* sadDist,
                   float* uRight, float* depth, int nframes, hipStream_t s) {
  const float minD = 0.f, maxD = mbf / mb;   // Frame.cc:859-861
  const int nRows = hp.lv[0].h;
  const int entriesPerFrame = capR * stereo_entries_per_keypoint(hp);
  hipLaunchKernelGGL(k_stereo_rows, dim3(nframes), dim3(256), (size_t)(nRows + 256) * 4, s, dp,
                     (const SKP*)kpsR, nR, capR, entriesPerFrame, rowStart, entries);
  hipLaunchKernelGGL(k_stereo_band_batch, dim3((capL + 255) / 256, nframes), dim3(256), 0, s, dp,
                     (const SKP*)kpsL, (const uint4*)descL, nL, (const SKP*)kpsR, (const uint4*)descR,
                     capL, capR, minD, maxD, entriesPerFrame, rowStart, entries, bandIdx, bandDist);
  hipLaunchKernelGGL(k_stereo_refine, dim3((capL + 15) / 16, nframes), dim3(256), 0, s, dp, pyrL, pyrR,
                     (const SKP*)kpsL, nL, (const SKP*)kpsR, capL, capR, bandIdx, bandDist, mbf, minD,
                     maxD, uRight, depth, sadDist);
  hipLaunchKernelGGL(k_stereo_outliers, dim3(nframes), dim3(256), 0, s, nL, capL, sadDist, uRight, depth);
}

// Frame::ComputeStereoFromRGBD, core/src/Frame.cc:998-1016: the depth image is read at the
// keypoint's own (distorted) position, indices truncated as Mat::at<float>(float, float) does;
// mvuRight comes from the undistorted x.
__global__ __launch_bounds__(256) void k_stereo_from_rgbd(const SKP* __restrict__ kps, const SKP* __restrict__ kpsUn,
                                                          const int* __restrict__ nP, int cap,
                                                          const float* __restrict__ depthImg, size_t frameFloats,
                                                          int stride, int W, int H, float mbf,
                                                          float* __restrict__ uRight, float* __restrict__ depth,
                                                          int* __restrict__ sadDist) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nP[f]) return;
  const size_t o = (size_t)f * cap + i;
  const SKP kp = kps[o];
  const int u = min(max((int)kp.x, 0), W - 1), v = min(max((int)kp.y, 0), H - 1);
  const float d = depthImg[(size_t)f * frameFloats + (size_t)v * stride + u];
  float ur = -1.0f, dp = -1.0f;
  if (d > 0) {
    dp = d;
    ur = __fsub_rn(kpsUn[o].x, __fdiv_rn(mbf, d));
  }
  uRight[o] = ur;
  depth[o] = dp;
  sadDist[o] = -1;
}

void launch_stereo_from_rgbd(const float* kps, const float* kpsUn, const int* n, int cap, const float* depthImg,
                             size_t frameFloats, int stride, int W, int H, float mbf, float* uRight, float* depth,
                             int* sadDist, int nframes, hipStream_t s) {
  hipLaunchKernelGGL(k_stereo_from_rgbd, dim3((cap + 255) / 256, nframes), dim3(256), 0, s, (const SKP*)kps,
                     (const SKP*)kpsUn, n, cap, depthImg, frameFloats, stride, W, H, mbf, uRight, depth, sadDist);
}

}  // namespace orbx
