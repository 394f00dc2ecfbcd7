// The consumer right after extraction, on the device-resident keypoints:
//   k_frame_grid        Frame::UndistortKeyPoints (core/src/Frame.cc:771-799) and
//                       Frame::AssignFeaturesToGrid / PosInGrid (:432-459, :751-762)
//   k_features_in_area  Frame::GetFeaturesInArea (:682-749)
// plus the host forms of cv::undistortPoints as the reference calls it and of
// Frame::ComputeImageBounds (:801-829).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/orbx.h"
#include "orbx_plan.h"

#define GRID_COLS 64   /* FRAME_GRID_COLS, core/include/Frame.h:47 */
#define GRID_ROWS 48   /* FRAME_GRID_ROWS, core/include/Frame.h:46 */
#define GRID_CELLS (GRID_COLS * GRID_ROWS)

namespace orbx {

struct GKP { float x, y, size, angle, response; int octave, class_id; };

// cv::undistortPoints(src, dst, K, D, noArray(), K): OpenCV's iterative inverse of the
// radial-tangential model with its fixed 5 iterations, in double; the new projection is K
// itself and the rectification is the identity.
__host__ __device__ inline void undistort_point(const orbx_pinhole& c, float uf, float vf, float* ox,
                                                float* oy) {
  const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
  const double k1 = c.k1, k2 = c.k2, p1 = c.p1, p2 = c.p2, k3 = c.k3;
  const double ifx = 1. / fx, ify = 1. / fy;
  const double u = uf, v = vf;
  double x = (u - cx) * ifx, y = (v - cy) * ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; ++j) {
    const double r2 = x * x + y * y;
    const double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
    if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
    const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
    const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
  *ox = (float)(xx * ww);
  *oy = (float)(yy * ww);
}

// One block per frame.  LDS: cell id per keypoint (u16), 3072 counters, 256 partial sums.
__global__ __launch_bounds__(256) void k_frame_grid(const GKP* __restrict__ kps,
                                                    const int* __restrict__ nP, int cap,
                                                    orbx_pinhole cam, int distorted, float minX,
                                                    float minY, float invW, float invH,
                                                    GKP* __restrict__ kpsUn,
                                                    int* __restrict__ cellStart,
                                                    int* __restrict__ entries) {
  extern __shared__ int sg[];
  int* cnt = sg;                       // GRID_CELLS
  int* part = cnt + GRID_CELLS;        // 256
  unsigned short* cellOf = (unsigned short*)(part + 256);   // cap
  const int f = blockIdx.x, tid = threadIdx.x;
  const int n = nP[f];
  const GKP* K = kps + (size_t)f * cap;
  GKP* U = kpsUn + (size_t)f * cap;
  for (int c = tid; c < GRID_CELLS; c += 256) cnt[c] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    GKP k = K[i];
    if (distorted) undistort_point(cam, k.x, k.y, &k.x, &k.y);
    U[i] = k;
    // PosInGrid: round() of a float is half away from zero
    const int posX = (int)roundf(__fmul_rn(__fsub_rn(k.x, minX), invW));
    const int posY = (int)roundf(__fmul_rn(__fsub_rn(k.y, minY), invH));
    const bool in = posX >= 0 && posX < GRID_COLS && posY >= 0 && posY < GRID_ROWS;
    const int c = in ? posX * GRID_ROWS + posY : 0xFFFF;
    cellOf[i] = (unsigned short)c;
    if (in) atomicAdd(&cnt[c], 1);
  }
  __syncthreads();
  // exclusive scan over the cells: 12 per thread, then 256 partial sums by wave 0
  const int per = GRID_CELLS / 256;
  int sum = 0;
  for (int c = tid * per; c < tid * per + per; ++c) sum += cnt[c];
  part[tid] = sum;
  __syncthreads();
  if (tid < 64) {
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
  int* cs = cellStart + (size_t)f * (GRID_CELLS + 1);
  int run = part[tid];
  for (int c = tid * per; c < tid * per + per; ++c) {
    const int k = cnt[c];
    cs[c] = run;
    cnt[c] = run;   // becomes the fill cursor
    run += k;
  }
  if (tid == 255) cs[GRID_CELLS] = run;
  __syncthreads();
  int* ent = entries + (size_t)f * cap;
  for (int i = tid; i < n; i += 256) {
    const int c = cellOf[i];
    if (c != 0xFFFF) ent[atomicAdd(&cnt[c], 1)] = i;
  }
  __syncthreads();
  // the reference's cells hold indices in push_back order = ascending; the atomics above do
  // not keep it, so every (short) cell list is put in order by the thread that owns the cell
  for (int c = tid * per; c < tid * per + per; ++c) {
    const int e0 = cs[c], e1 = cnt[c];   // cursor has advanced to the end of the cell
    for (int a = e0 + 1; a < e1; ++a) {
      const int v = ent[a];
      int b = a - 1;
      while (b >= e0 && ent[b] > v) { ent[b + 1] = ent[b]; --b; }
      ent[b + 1] = v;
    }
  }
}

// One lane per query (x, y, r, minLevel, maxLevel); the first `percap` hits are written in the
// reference's order (cell column, cell row, order inside the cell), the count is the total.
__global__ __launch_bounds__(256) void k_features_in_area(const GKP* __restrict__ kpsUn, int cap,
                                                          const int* __restrict__ cellStart,
                                                          const int* __restrict__ entries, int frame,
                                                          float minX, float minY, float invW, float invH,
                                                          const float* __restrict__ queries, int nq,
                                                          int* __restrict__ out, int percap,
                                                          int* __restrict__ counts) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const float x = queries[5 * q], y = queries[5 * q + 1], r = queries[5 * q + 2];
  const int minLevel = (int)queries[5 * q + 3], maxLevel = (int)queries[5 * q + 4];
  const GKP* U = kpsUn + (size_t)frame * cap;
  const int* cs = cellStart + (size_t)frame * (GRID_CELLS + 1);
  const int* ent = entries + (size_t)frame * cap;
  int found = 0;
  const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, minX), r), invW)));
  const int cx1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, minX), r), invW)));
  const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, minY), r), invH)));
  const int cy1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, minY), r), invH)));
  if (cx0 < GRID_COLS && cx1 >= 0 && cy0 < GRID_ROWS && cy1 >= 0) {
    const bool levelFilter = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = cx0; ix <= cx1; ++ix)
      for (int iy = cy0; iy <= cy1; ++iy) {
        const int c = ix * GRID_ROWS + iy;
        for (int e = cs[c]; e < cs[c + 1]; ++e) {
          const int i = ent[e];
          const GKP kp = U[i];
          if (levelFilter) {
            if (kp.octave < minLevel) continue;
            if (maxLevel >= 0 && kp.octave > maxLevel) continue;
          }
          const float distx = __fsub_rn(kp.x, x), disty = __fsub_rn(kp.y, y);
          if (fabsf(distx) < r && fabsf(disty) < r) {
            if (found < percap) out[(size_t)q * percap + found] = i;
            ++found;
          }
        }
      }
  }
  counts[q] = found;
}

size_t frame_grid_lds_bytes(int cap) { return (size_t)(GRID_CELLS + 256) * 4 + (size_t)cap * 2 + 16; }

void launch_frame_grid(const float* d_kps, const int* d_n, int cap, const orbx_pinhole& cam, int distorted,
                       const float* bounds, float* d_kpsUn, int* d_cellStart, int* d_entries,
                       int nframes, hipStream_t s) {
  const float invW = (float)GRID_COLS / (bounds[1] - bounds[0]);   // Frame.cc:193-196
  const float invH = (float)GRID_ROWS / (bounds[3] - bounds[2]);
  hipLaunchKernelGGL(k_frame_grid, dim3(nframes), dim3(256), frame_grid_lds_bytes(cap), s,
                     (const GKP*)d_kps, d_n, cap, cam, distorted, bounds[0], bounds[2], invW, invH,
                     (GKP*)d_kpsUn, d_cellStart, d_entries);
}

void launch_features_in_area(const float* d_kpsUn, int cap, const int* d_cellStart, const int* d_entries,
                             int frame, const float* bounds, const float* d_queries, int nq, int* d_out,
                             int percap, int* d_counts, hipStream_t s) {
  const float invW = (float)GRID_COLS / (bounds[1] - bounds[0]);
  const float invH = (float)GRID_ROWS / (bounds[3] - bounds[2]);
  hipLaunchKernelGGL(k_features_in_area, dim3((nq + 255) / 256), dim3(256), 0, s, (const GKP*)d_kpsUn, cap,
                     d_cellStart, d_entries, frame, bounds[0], bounds[2], invW, invH, d_queries, nq, d_out,
                     percap, d_counts);
}

}  // namespace orbx

extern "C" {

int orbx_undistort_points(const orbx_pinhole* cam, const float* xy, int n, float* out_xy) {
  if (!cam || n < 0 || (n > 0 && (!xy || !out_xy))) return ORBX_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i) orbx::undistort_point(*cam, xy[2 * i], xy[2 * i + 1], &out_xy[2 * i], &out_xy[2 * i + 1]);
  return ORBX_OK;
}

int orbx_image_bounds(const orbx_pinhole* cam, int width, int height, float* bounds) {
  if (!bounds || width <= 0 || height <= 0) return ORBX_ERR_BAD_ARG;
  if (cam && cam->k1 != 0.0f) {
    const float c[8] = {0.f, 0.f, (float)width, 0.f, 0.f, (float)height, (float)width, (float)height};
    float m[8];
    orbx_undistort_points(cam, c, 4, m);
    bounds[0] = std::fmin(m[0], m[4]);
    bounds[1] = std::fmax(m[2], m[6]);
    bounds[2] = std::fmin(m[1], m[3]);
    bounds[3] = std::fmax(m[5], m[7]);
  } else {
    bounds[0] = 0.0f; bounds[1] = (float)width; bounds[2] = 0.0f; bounds[3] = (float)height;
  }
  return ORBX_OK;
}

}  // extern "C"
