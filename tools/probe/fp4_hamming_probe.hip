// Hamming distances of 32 train x 32 query 256-bit descriptors on the block-scaled FP4 matrix
// instruction (v_mfma_scale_f32_32x32x64_f8f6f4, both operands e2m1, scales 2^0): checks the
// operand / result layout against popcounts, then times back-to-back instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define M2 0x22222222u
__device__ __forceinline__ v8i train_frag(uint32_t w) {   // bit -> nibble 0x2 (1.0) or 0x0
  v8i r = {(int)((w << 1) & M2), (int)(w & M2), (int)((w >> 1) & M2), (int)((w >> 2) & M2), 0, 0, 0, 0};
  return r;
}
__device__ __forceinline__ v8i query_frag(uint32_t w) {   // bit 0 -> 0x2 (+1.0), bit 1 -> 0xA (-1.0)
  v8i t = train_frag(w);
  v8i r = {(int)(M2 | ((uint32_t)t[0] << 2)), (int)(M2 | ((uint32_t)t[1] << 2)), (int)(M2 | ((uint32_t)t[2] << 2)),
           (int)(M2 | ((uint32_t)t[3] << 2)), 0, 0, 0, 0};
  return r;
}
__global__ void k_check(const uint32_t* train, const uint32_t* query, int* dist) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  int qpop = 0;
  for (int s = 0; s < 8; ++s) qpop += __builtin_popcount(query[r * 8 + s]);
  for (int s = 0; s < 4; ++s) {
    const v8i a = train_frag(train[r * 8 + 2 * s + h]);
    const v8i b = query_frag(query[r * 8 + 2 * s + h]);
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;   // train row
    dist[row * 32 + r] = qpop + (int)acc[i];          // [train][query]
  }
}
__global__ __launch_bounds__(64) void k_rate(float* out, int iters) {
  v8i a = {(int)(threadIdx.x * 0x01010101u & M2), 0x22222222, 0x02020202, 0x20202020, 0, 0, 0, 0};
  v8i b = {0x2a2a2a2a, (int)0xa2a2a2a2u, 0x22222222, (int)0xaaaaaaaau, 0, 0, 0, 0};
  v16f acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
// The instruction stream of the 2-NN kernel, one property at a time: NACC accumulators per wave,
// each the sum of four dependent steps; RESET: every group of four starts from the inline zero (else
// the accumulators run on); ROTATE: the train operand sets change places every group; VALU: the
// kernel's vector work between tiles (a 16-value float min3 chain per accumulator).
template <int NACC, int RESET, int ROTATE, int VALU>
__global__ __launch_bounds__(64) void k_rate_real(const int* __restrict__ rnd, float* out, int iters) {
  v8i a[4], b[NACC];
  for (int k = 0; k < 4; ++k) {
    a[k] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
    for (int d = 0; d < 4; ++d) a[k][d] = rnd[(k * 4 + d) * 64 + threadIdx.x] & 0x22222222;
  }
  for (int k = 0; k < NACC; ++k) {
    b[k] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
    for (int d = 0; d < 4; ++d) b[k][d] = (rnd[(16 + (k & 1) * 4 + d) * 64 + threadIdx.x] & (int)0x88888888u) | 0x22222222;
  }
  float best = 1e9f;
  v16f zero;
  for (int i = 0; i < 16; ++i) zero[i] = 0.f;
  v16f acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = zero;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j)
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[u], b[j], (RESET && u == 0) ? zero : acc[j], 4, 4, 0,
                                                                 0x7f7f7f7f, 0, 0x7f7f7f7f);
    if (VALU) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        float mn = acc[j][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mn = fminf(mn, acc[j][i]);
        best = fminf(best, mn);
      }
    } else if (RESET) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) best += acc[j][0];   // keeps the group's results alive
    }
    if (ROTATE) { const v8i t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = t; }
  }
  for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) best += acc[j][i];
  out[blockIdx.x * 64 + threadIdx.x] = best;
}
// Software-pipelined form: the min3 chains of the previous group are placed between the matrix
// instructions of the current one (two accumulator sets), so that the vector work issues while the
// matrix pipe is busy.
template <int NACC>
__global__ __launch_bounds__(64) void k_rate_pipe(const int* __restrict__ rnd, float* out, int iters) {
  v8i a[4], b[NACC];
  for (int k = 0; k < 4; ++k) {
    a[k] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
    for (int d = 0; d < 4; ++d) a[k][d] = rnd[(k * 4 + d) * 64 + threadIdx.x] & 0x22222222;
  }
  for (int k = 0; k < NACC; ++k) {
    b[k] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
    for (int d = 0; d < 4; ++d) b[k][d] = (rnd[(16 + k % 2 * 4 + d) * 64 + threadIdx.x] & (int)0x88888888u) | 0x22222222 | (k << 28);
  }
  float best = 1e9f;
  v16f zero;
  for (int i = 0; i < 16; ++i) zero[i] = 0.f;
  v16f accP[NACC], accQ[NACC];
  for (int j = 0; j < NACC; ++j) { accP[j] = zero; accQ[j] = zero; }
  for (int it = 0; it < iters; it += 2) {
    // group into accP while accQ (previous group) is reduced
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j)
        accP[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[u], b[j], u == 0 ? zero : accP[j], 4, 4, 0, 0x7f7f7f7f,
                                                                  0, 0x7f7f7f7f);
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      float mn = accQ[j][0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mn = fminf(mn, accQ[j][i]);
      best = fminf(best, mn);
    }
#pragma unroll
    for (int k = 0; k < 4 * NACC; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one matrix instruction
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // up to three vector instructions
    }
    { const v8i t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = t; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j)
        accQ[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[u], b[j], u == 0 ? zero : accQ[j], 4, 4, 0, 0x7f7f7f7f,
                                                                  0, 0x7f7f7f7f);
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      float mn = accP[j][0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mn = fminf(mn, accP[j][i]);
      best = fminf(best, mn);
    }
#pragma unroll
    for (int k = 0; k < 4 * NACC; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
    { const v8i t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = t; }
  }
  for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) best += accP[j][i] + accQ[j][i];
  out[blockIdx.x * 64 + threadIdx.x] = best;
}
template <int NACC>
static void time_pipe(const int* drnd, float* df, hipEvent_t e0, hipEvent_t e1) {
  for (int occ = 1; occ <= 4; occ *= 2) {
    const int iters = 20000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL((k_rate_pipe<NACC>), dim3(256 * 4 * occ), dim3(64), 0, 0, drnd, df, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double ns = ms * 1e6 / ((double)iters * 4 * NACC * occ);
    std::printf("fp4 pipelined: min3 chains of the previous group between the matrix instructions, %d accumulators x 2 sets, %d wave(s) per SIMD: %.2f ns per instruction per SIMD -> %.2f T pairs/s\n",
                NACC, occ, ns, 256.0 / ns * 1024 / 1e3);
  }
}
template <int NACC, int RESET, int ROTATE, int VALU>
static void time_real(const int* drnd, float* df, hipEvent_t e0, hipEvent_t e1) {
  for (int occ = 1; occ <= 4; occ *= 2) {
    const int iters = 20000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL((k_rate_real<NACC, RESET, ROTATE, VALU>), dim3(256 * 4 * occ), dim3(64), 0, 0, drnd, df, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double ns = ms * 1e6 / ((double)iters * 4 * NACC * occ);
    std::printf("fp4 random operands, %d accumulators%s%s%s, %d wave(s) per SIMD: %.2f ns per instruction per SIMD -> %.2f T pairs/s\n",
                NACC, RESET ? ", zero start per group" : "", ROTATE ? ", operands rotate" : "", VALU ? ", min3 chains" : "",
                occ, ns, 256.0 / ns * 1024 / 1e3);
  }
}
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void k_rate_i8(int* out, int iters) {
  v4i a = {(int)(threadIdx.x * 0x01010101u & 0x01010101u), 0x01010101, 0x00010001, 0x01000100};
  v4i b = {0x01ff01ff, (int)0xff01ff01u, 0x01010101, (int)0xffffffffu};
  v16i acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
  int s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
  std::vector<uint32_t> t(32 * 8), q(32 * 8);
  srand(7);
  for (auto& v : t) v = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand();
  for (auto& v : q) v = (uint32_t)rand() * 40503u ^ ((uint32_t)rand() << 11);
  for (int i = 0; i < 8; ++i) { q[i] = t[5 * 8 + i]; t[31 * 8 + i] = 0xFFFFFFFFu; t[30 * 8 + i] = 0; }
  uint32_t *dt, *dq; int* dd;
  (void)hipMalloc(&dt, t.size() * 4); (void)hipMalloc(&dq, q.size() * 4); (void)hipMalloc(&dd, 1024 * 4);
  (void)hipMemcpy(dt, t.data(), t.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dt, dq, dd);
  std::vector<int> d(1024);
  (void)hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    int ref = 0;
    for (int s = 0; s < 8; ++s) ref += __builtin_popcount(t[i * 8 + s] ^ q[j * 8 + s]);
    if (ref != d[i * 32 + j]) { if (bad < 5) std::printf("mismatch train %d query %d: %d vs %d\n", i, j, d[i * 32 + j], ref); ++bad; }
  }
  std::printf("layout check: %d mismatches of 1024 (distance of query 0 to train 5: %d)\n", bad, d[5 * 32 + 0]);
  float* df; (void)hipMalloc(&df, 256 * 4 * 8 * 64 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int occ = 1; occ <= 2; ++occ) {
    for (int kind = 0; kind < 2; ++kind) {
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k_rate, dim3(256 * 4 * occ), dim3(64), 0, 0, df, iters);
        else hipLaunchKernelGGL(k_rate_i8, dim3(256 * 4 * occ), dim3(64), 0, 0, (int*)df, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      }
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double ns = ms * 1e6 / ((double)iters * 4 * occ);
      const double pairs = 1024.0 * (kind == 0 ? 0.25 : 0.125);   // 32x32 pairs need 4 (fp4) or 8 (i8) instructions
      std::printf("%s, %d wave(s) per SIMD: %.2f ns per instruction per SIMD -> %.2f T pairs/s chip-wide\n",
                  kind == 0 ? "fp4 32x32x64" : "i8 32x32x32", occ, ns, pairs / ns * 1024 / 1e3);
    }
  }
  {
    std::vector<int> rnd(24 * 64);
    for (auto& v : rnd) v = (int)((uint32_t)rand() * 2654435761u ^ (uint32_t)rand());
    int* drnd; (void)hipMalloc(&drnd, rnd.size() * 4);
    (void)hipMemcpy(drnd, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice);
    time_real<4, 0, 0, 0>(drnd, df, e0, e1);
    time_real<2, 0, 0, 0>(drnd, df, e0, e1);
    time_real<2, 1, 1, 0>(drnd, df, e0, e1);
    time_real<2, 1, 1, 1>(drnd, df, e0, e1);
    time_pipe<2>(drnd, df, e0, e1);
  }
  return 0;
}
