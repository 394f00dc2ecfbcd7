// Stereo association of two extractions that are still resident on the device:
// Frame::ComputeStereoMatches, core/src/Frame.cc:831-996.
//   k_stereo_band_batch   row-band descriptor search             :840-908
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

// One lane per left keypoint; right keypoints are visited in ascending index (wave-uniform
// loads), so the first minimum wins as in the reference's row buckets, which hold ascending
// right indices.
__global__ __launch_bounds__(256) void k_stereo_band_batch(
    const Plan* __restrict__ plan, const SKP* __restrict__ kpsL, const uint32_t* __restrict__ descL,
    const int* __restrict__ nLp, const SKP* __restrict__ kpsR, const uint32_t* __restrict__ descR,
    const int* __restrict__ nRp, int capL, int capR, float minD, float maxD,
    int* __restrict__ bestIdx, int* __restrict__ bestDist) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int nL = nLp[f], nR = nRp[f];
  if (blockIdx.x * 256 >= nL) return;
  const bool valid = i < nL;
  const SKP* kl = kpsL + (size_t)f * capL;
  const SKP* kr = kpsR + (size_t)f * capR;
  const uint32_t* dl = descL + (size_t)f * capL * 8;
  const uint32_t* dr = descR + (size_t)f * capR * 8;
  const int nRows = plan->lv[0].h;
  SKP L = kl[valid ? i : 0];
  uint32_t a[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) a[w] = dl[(size_t)(valid ? i : 0) * 8 + w];
  const int row = (int)L.y;
  const float minU = __fsub_rn(L.x, maxD), maxU = __fsub_rn(L.x, minD);
  int best = 100, bi = -1;
  const bool live = valid && row >= 0 && row < nRows && !(maxU < 0.f);
  for (int j = 0; j < nR; ++j) {
    const SKP R = kr[j];
    const float r = __fmul_rn(2.0f, plan->lv[R.octave].scale);
    const int maxr = (int)ceilf(__fadd_rn(R.y, r));
    const int minr = (int)floorf(__fsub_rn(R.y, r));
    bool ok = live && row >= minr && row <= maxr;
    ok = ok && R.octave >= L.octave - 1 && R.octave <= L.octave + 1;
    ok = ok && R.x >= minU && R.x <= maxU;
    if (__ballot(ok) == 0ull) continue;
    int d = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) d += __builtin_popcount(a[w] ^ dr[(size_t)j * 8 + w]);
    if (ok && d < best) { best = d; bi = j; }
  }
  if (valid) {
    bestIdx[(size_t)f * capL + i] = bi;
    bestDist[(size_t)f * capL + i] = best;
  }
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

void launch_stereo(const Plan& hp, const Plan* dp, const uint8_t* pyrL, const uint8_t* pyrR,
                   const float* kpsL, const uint8_t* descL, const int* nL, int capL,
                   const float* kpsR, const uint8_t* descR, const int* nR, int capR, float mbf,
                   float mb, int* bandIdx, int* bandDist, int* sadDist, float* uRight, float* depth,
                   int nframes, hipStream_t s) {
  (void)hp;
  const float minD = 0.f, maxD = mbf / mb;   // Frame.cc:859-861
  hipLaunchKernelGGL(k_stereo_band_batch, dim3((capL + 255) / 256, nframes), dim3(256), 0, s, dp,
                     (const SKP*)kpsL, (const uint32_t*)descL, nL, (const SKP*)kpsR,
                     (const uint32_t*)descR, nR, capL, capR, minD, maxD, bandIdx, bandDist);
  hipLaunchKernelGGL(k_stereo_refine, dim3((capL + 15) / 16, nframes), dim3(256), 0, s, dp, pyrL, pyrR,
                     (const SKP*)kpsL, nL, (const SKP*)kpsR, capL, capR, bandIdx, bandDist, mbf, minD,
                     maxD, uRight, depth, sadDist);
  hipLaunchKernelGGL(k_stereo_outliers, dim3(nframes), dim3(256), 0, s, nL, capL, sadDist, uRight, depth);
}

}  // namespace orbx
