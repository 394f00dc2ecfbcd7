// Vector-memory pipeline throughput on gfx950 for the access shapes the resize kernel uses:
// cycles per wave-level load instruction per CU, data resident in L1/L2.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 2048
struct __attribute__((packed, aligned(4))) U3 { uint32_t x, y, z; };
struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ buf, uint32_t* out, int rowBytes) {
  const int lane = threadIdx.x & 63;
  const uint8_t* base = buf + (size_t)(blockIdx.x & 7) * 65536 + (threadIdx.x >> 6) * 1024;
  uint32_t acc = 0;
  int off;
  if (MODE == 0) off = lane * 4;                       // dword, coalesced
  else if (MODE == 1 || MODE == 2 || MODE == 3) off = (lane * 24 / 5) & ~3;   // stride 4.8 B
  else if (MODE == 4) off = lane * 16;                 // dwordx4 coalesced
  else if (MODE == 5) off = lane * 8;                  // dwordx2 coalesced
  else off = lane * 4;
#pragma unroll 8
  for (int it = 0; it < ITERS; ++it) {
    const uint8_t* p = base + ((it * 5) & 63) * rowBytes + off;
    if (MODE == 0 || MODE == 1) acc ^= *(const uint32_t*)p;
    else if (MODE == 2) { acc ^= *(const uint32_t*)p; acc ^= *(const uint32_t*)(p + 4); acc ^= *(const uint32_t*)(p + 8); }
    else if (MODE == 3) { U3 v = *(const U3*)p; acc ^= v.x ^ v.y ^ v.z; }
    else if (MODE == 4) { U4 v = *(const U4*)p; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    else if (MODE == 5) { uint2 v = *(const uint2*)p; acc ^= v.x ^ v.y; }
    else if (MODE == 6) { *(volatile uint32_t*)(out + 65536 + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4096 + ((it * 5) & 63) * 64 + lane) = it; }
  }
  out[(blockIdx.x * 256 + threadIdx.x) & 0xFFFF] = acc;
}

template <int MODE>
void run(const char* name, const uint8_t* d, uint32_t* o, int instrPerIter) {
  const int blocks = 256 * 2;   // 8 waves per CU
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, o, 512);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, o, 512);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wavesPerCU = 8.0;
  const double ns = ms * 1e6 / (ITERS * instrPerIter * wavesPerCU);
  std::printf("%-34s %.2f ns per wave-instruction per CU (~%.1f cycles at 2.4 GHz)\n", name, ns, ns * 2.4);
}

int main() {
  uint8_t* d; uint32_t* o;
  hipMalloc(&d, 1 << 20); hipMemset(d, 1, 1 << 20);
  hipMalloc(&o, 64 << 20);
  run<0>("dword, coalesced", d, o, 1);
  run<1>("dword, lane stride 4.8 B", d, o, 1);
  run<2>("3 x dword, lane stride 4.8 B", d, o, 3);
  run<3>("dwordx3, lane stride 4.8 B", d, o, 1);
  run<4>("dwordx4, coalesced", d, o, 1);
  run<5>("dwordx2, coalesced", d, o, 1);
  run<6>("store dword, coalesced", d, o, 1);
  return 0;
}
