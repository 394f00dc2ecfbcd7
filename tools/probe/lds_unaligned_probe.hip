// Cost of LDS dword reads at byte addresses that are not multiples of 4 (gfx950 runs with
// unaligned LDS access enabled): time per ds_read_b32 wave-instruction for byte offset 0..3, at a
// 12-byte lane stride (the access pattern of a lane walking one image row of a 48-byte-stride tile)
// and at a 4-byte lane stride, 4 and 8 waves per CU-SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
struct __attribute__((packed, aligned(1))) U1 { uint32_t v; };
#define ITERS 2048
template <int STRIDE>
__global__ __launch_bounds__(64) void k(uint32_t* out, int c) {
  __shared__ __align__(16) uint32_t s[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = i * 2654435761u;
  __syncthreads();
  const uint8_t* p = (const uint8_t*)s + threadIdx.x * STRIDE + c;
  uint32_t acc = 0;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      uint32_t v;
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((uint32_t)(uintptr_t)p), "n"(u * 256));
      asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
      acc ^= v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main() {
  uint32_t* d; (void)hipMalloc(&d, 256 * 32 * 64 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int occ = 4; occ <= 8; occ *= 2)
    for (int stride = 4; stride <= 12; stride += 8)
      for (int c = 0; c < 4; ++c) {
        auto fn = stride == 4 ? k<4> : k<12>;
        hipLaunchKernelGGL(fn, dim3(256 * 4 * occ), dim3(64), 0, 0, d, c);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(fn, dim3(256 * 4 * occ), dim3(64), 0, 0, d, c);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        // per CU: 4 * occ waves, each ITERS * 8 reads
        const double ns = ms * 1e6 / 5 / ((double)ITERS * 8 * 4 * occ);
        std::printf("occ %d lane stride %2d byte offset %d: %.3f ns per ds_read_b32 wave-instruction per CU\n", occ, stride, c, ns);
      }
  return 0;
}
