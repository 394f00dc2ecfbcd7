// Stand-alone timing probe for the level-0 pyramid kernel variants (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../orb_slam3_amd/csrc/orbx_extract.hip"

__global__ void k_copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}
__global__ void k_write16(uint4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = make_uint4(1, 2, 3, 4);
}


template <int ROWS, int MODE>
__global__ __launch_bounds__(256) void k_l0v(const uint8_t* __restrict__ img, int W, int H, int istride, size_t iframe,
                                             uint8_t* __restrict__ dst, int pstride, int prows, size_t dplane) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  const int py0 = (blockIdx.y * 4 + threadIdx.y) * ROWS;
  const int f = blockIdx.z;
  const int P = g * 16;
  if (P >= pstride || py0 >= prows) return;
  const uint8_t* fimg = img + (size_t)f * iframe;
  uint8_t* fdst = dst + (size_t)f * dplane + P;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int py = py0 + r;
    if (py >= prows) break;
    int sy = py - 19; if (sy < 0) sy = -sy; if (sy >= H) sy = 2 * H - 2 - sy;
    const uint8_t* srow = fimg + (size_t)sy * istride;
    uint4 out;
    if (MODE == 0) {   // aligned 16B load (wrong bytes, timing only)
      int off = P - 32; if (off < 0) off = 0; if (off > W - 16) off = W - 16;
      out = *(const uint4*)(srow + off);
    } else if (MODE == 1) {  // unaligned-by-dword x4 + 1, alignbyte
      int off = P - 19; if (off < 4) off = 4; if (off > W - 24) off = W - 24;
      const uintptr_t a = (uintptr_t)(srow + off);
      const int sh = (int)(a & 3);
      const uint8_t* ap = (const uint8_t*)(a - sh);
      const orbx::U4A4 v = *(const orbx::U4A4*)ap;
      const uint32_t v4 = *(const uint32_t*)(ap + 16);
      out.x = __builtin_amdgcn_alignbyte(v.y, v.x, sh);
      out.y = __builtin_amdgcn_alignbyte(v.z, v.y, sh);
      out.z = __builtin_amdgcn_alignbyte(v.w, v.z, sh);
      out.w = __builtin_amdgcn_alignbyte(v4, v.w, sh);
    } else {  // 5 separate dword loads
      int off = P - 19; if (off < 4) off = 4; if (off > W - 24) off = W - 24;
      const uintptr_t a = (uintptr_t)(srow + off);
      const int sh = (int)(a & 3);
      const uint32_t* ap = (const uint32_t*)(a - sh);
      const uint32_t v0 = ap[0], v1 = ap[1], v2 = ap[2], v3 = ap[3], v4 = ap[4];
      out.x = __builtin_amdgcn_alignbyte(v1, v0, sh);
      out.y = __builtin_amdgcn_alignbyte(v2, v1, sh);
      out.z = __builtin_amdgcn_alignbyte(v3, v2, sh);
      out.w = __builtin_amdgcn_alignbyte(v4, v3, sh);
    }
    *(uint4*)(fdst + (size_t)py * pstride) = out;
  }
}

int main() {
  const int W = 752, H = 480, B = 256;
  const int pstride = 832, prows = H + 38;
  const size_t iframe = (size_t)W * H, dplane = (size_t)pstride * prows;
  uint8_t *img, *dst;
  hipMalloc(&img, iframe * B);
  hipMalloc(&dst, dplane * B);
  hipMemset(img, 7, iframe * B);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) fn();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us\n", name, ms * 1000 / 20);
  };
  time("level0", [&] {
    const int nbx = (pstride / 16 + 63) / 64;
    dim3 grid(nbx + 1, (prows + 4 * L0_ROWS - 1) / (4 * L0_ROWS), B);
    hipLaunchKernelGGL(orbx::k_pyr_level0, grid, dim3(64, 4), 0, 0, img, W, H, W, iframe, dst, pstride, prows, dplane, nbx, 0);
  });
#define RUNV(R, M) time("l0v rows=" #R " mode=" #M, [&] { dim3 grid((pstride / 16 + 63) / 64, (prows + 4 * R - 1) / (4 * R), B); \
    hipLaunchKernelGGL((k_l0v<R, M>), grid, dim3(64, 4), 0, 0, img, W, H, W, iframe, dst, pstride, prows, dplane); });
  RUNV(1, 0) RUNV(1, 1) RUNV(1, 2) RUNV(2, 0) RUNV(4, 0) RUNV(4, 1) RUNV(16, 0) RUNV(16, 1)
  const size_t n16 = iframe * B / 16;
  time("copy16 92MB->92MB", [&] { hipLaunchKernelGGL(k_copy16, dim3((n16 + 255) / 256), dim3(256), 0, 0, (const uint4*)img, (uint4*)dst, n16); });
  const size_t w16 = dplane * B / 16;
  time("write16 110MB", [&] { hipLaunchKernelGGL(k_write16, dim3((w16 + 255) / 256), dim3(256), 0, 0, (uint4*)dst, w16); });
  return 0;
}
