// Host-side stand-in for <hip/hip_runtime.h>, used ONLY by tests/hostsim.
//
// Purpose: compile the library's .hip sources as plain C++ for x86 and execute every
// kernel thread by thread on the CPU, so that
//   * every global-memory access a kernel makes is checked against the allocation it belongs
//     to (allocations sit flush against PROT_NONE guard pages; an out-of-bounds access stops
//     the run with the kernel, workgroup, thread and address),
//   * dynamic LDS is guarded the same way,
//   * wave-level code that is not convergent (lanes of one wave meeting at different
//     cross-lane operations) is reported,
//   * kernel logic can be compared with the CPU oracle where no GPU is available.
// This is test infrastructure.  It is never built into, loaded by or shipped with the
// product library (orb_slam3_amd/liborbx_hip.so), and it says nothing about speed.
//
// Execution model: one workgroup at a time per OS thread; every GPU thread of the workgroup
// is a fiber; a fiber runs until its next cross-lane operation or barrier.  Wave = 64 lanes.
#ifndef HOSTSIM_HIP_RUNTIME_H_
#define HOSTSIM_HIP_RUNTIME_H_

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <tuple>

#define __HOSTSIM__ 1

// ---- qualifiers -------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __constant__ static
#define __align__(n) __attribute__((aligned(n)))

// ---- vector types -----------------------------------------------------------------------
struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };
struct __attribute__((aligned(8))) uint2 { uint32_t x, y; };
struct __attribute__((aligned(8))) int2 { int x, y; };
struct __attribute__((aligned(8))) float2 { float x, y; };
struct __attribute__((aligned(16))) uint4 { uint32_t x, y, z, w; };
struct __attribute__((aligned(16))) int4 { int x, y, z, w; };
struct __attribute__((aligned(16))) float4 { float x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

// ---- runtime API subset -------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2,
                     hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
struct hostsim_stream;
struct hostsim_event;
typedef hostsim_stream* hipStream_t;
typedef hostsim_event* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { int type; int device; void* devicePointer; void* hostPointer; };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

// a file-scope __device__ array is an ordinary host array here: its "device address" is its own
#define HIP_SYMBOL(x) x
template <class T> inline hipError_t hipGetSymbolAddress(void** p, const T& symbol) {
  *p = (void*)&symbol;
  return hipSuccess;
}

extern "C" {
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
struct hostsim_mempool;
typedef hostsim_mempool* hipMemPool_t;
enum hipMemPoolAttr { hipMemPoolAttrReleaseThreshold = 4 };
hipError_t hipDeviceGetDefaultMemPool(hipMemPool_t* pool, int device);
hipError_t hipMemPoolSetAttribute(hipMemPool_t pool, hipMemPoolAttr attr, void* value);
hipError_t hipMallocAsync(void** p, size_t n, hipStream_t s);
hipError_t hipFreeAsync(void* p, hipStream_t s);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);
inline hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }   // launches run to completion before they return
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, hipMemcpyKind k);
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h,
                            hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipDeviceSynchronize(void);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v);
}

// ---- execution state ----------------------------------------------------------------------
namespace hostsim {

struct Lane;
struct Block;
extern thread_local Lane* tl_lane;
extern thread_local Block* tl_block;

struct Lane {
  uint3 tid;           // threadIdx
  int lin;             // linear thread id inside the workgroup
  void* sp;            // saved stack pointer of the fiber
  bool done;
  const volatile uint64_t* waitOn;   // generation counter this lane waits on (nullptr = runnable)
  uint64_t waitVal;
  int site;
  unsigned opCount;    // cross-lane operations executed so far (selects the exchange buffer)
};

// one cross-lane exchange: every alive lane of the wave deposits a 64-bit value; returns the
// table of the 64 deposited values and the mask of lanes that took part
struct Gathered { const uint64_t* v; uint64_t mask; };
Gathered wave_gather(int site, uint64_t mine);
void block_barrier(int site);
int block_barrier_or(int site, int pred);
uint8_t* dyn_lds();
const uint3& thread_idx();
const uint3& block_idx();
const dim3& block_dim();
const dim3& grid_dim();
int lane_in_wave();

void launch(const char* name, const void* kernel, dim3 grid, dim3 block, size_t ldsBytes,
            const std::function<void()>& fn);

template <class T> inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "shuffle operand");
  uint64_t b = 0;
  memcpy(&b, &v, sizeof(T));
  return b;
}
template <class T> inline T from_bits(uint64_t b) {
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}
[[noreturn]] void fail(const char* what, int site);

template <class T> inline T shfl_from(int site, T v, int src) {
  const Gathered g = wave_gather(site, to_bits(v));
  if (src < 0 || src > 63 || !((g.mask >> src) & 1)) return v;   // inactive source: own value
  return from_bits<T>(g.v[src]);
}
template <class T> inline T shfl(int site, T v, int src, int width = 64) {
  const int lane = lane_in_wave();
  const int base = lane & ~(width - 1);
  return shfl_from(site, v, base + (src & (width - 1)));
}
template <class T> inline T shfl_up(int site, T v, unsigned d, int width = 64) {
  const int lane = lane_in_wave();
  const int base = lane & ~(width - 1);
  const int src = lane - (int)d;
  return shfl_from(site, v, src < base ? lane : src);
}
template <class T> inline T shfl_down(int site, T v, unsigned d, int width = 64) {
  const int lane = lane_in_wave();
  const int base = lane & ~(width - 1);
  const int src = lane + (int)d;
  return shfl_from(site, v, src >= base + width ? lane : src);
}
template <class T> inline T shfl_xor(int site, T v, int m, int width = 64) {
  const int lane = lane_in_wave();
  const int base = lane & ~(width - 1);
  const int src = lane ^ m;
  return shfl_from(site, v, (src < base || src >= base + width) ? lane : src);
}
inline unsigned long long ballot(int site, int pred) {
  const Gathered g = wave_gather(site, pred ? 1u : 0u);
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i)
    if (((g.mask >> i) & 1) && g.v[i]) m |= 1ull << i;
  return m;
}
template <class T> inline T readfirstlane(int site, T v) {
  const Gathered g = wave_gather(site, to_bits(v));
  return from_bits<T>(g.v[__builtin_ctzll(g.mask)]);
}
inline void wave_barrier(int site) { (void)wave_gather(site, 0); }

// v_mov_b32 with a DPP control, as __builtin_amdgcn_update_dpp gives it: the lane takes src of the
// lane the control names; a lane whose row or bank is masked out, or whose source lane does not
// exist or is inactive while bound_ctrl is clear, keeps `old`; with bound_ctrl set a missing
// source reads as 0.  Emulated controls: row_shr:1..15 (0x111..0x11f), row_bcast:15 (0x142),
// row_bcast:31 (0x143).
inline uint32_t update_dpp(int site, uint32_t old, uint32_t src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const Gathered g = wave_gather(site, src);
  const int lane = lane_in_wave(), row = lane >> 4, inrow = lane & 15;
  if (!((row_mask >> row) & 1) || !((bank_mask >> (inrow >> 2)) & 1)) return old;
  int from = -1;
  if (ctrl >= 0x111 && ctrl <= 0x11f) {
    const int n = ctrl - 0x110;
    from = inrow >= n ? lane - n : -1;
  } else if (ctrl == 0x142) {
    from = row >= 1 ? 16 * row - 1 : -1;
  } else if (ctrl == 0x143) {
    from = row >= 2 ? 31 : -1;
  } else {
    fail("update_dpp: control not emulated", site);
  }
  if (from < 0 || !((g.mask >> from) & 1)) return bound_ctrl ? 0u : old;
  return (uint32_t)g.v[from];
}

// v_mfma_i32_32x32x32_i8: D = A (32 x 32, signed bytes) * B (32 x 32) + C.  Lane l holds row l & 31
// of A and column l & 31 of B for the 16 k values 16 (l >> 5) .. + 15 (one per byte of its four
// dwords), and of C / D column l & 31, rows (i & 3) + 8 (i >> 2) + 4 (l >> 5) for register i.
typedef int sim_v4i __attribute__((ext_vector_type(4)));
typedef int sim_v16i __attribute__((ext_vector_type(16)));
inline sim_v16i mfma_i32_32x32x32_i8(int site, sim_v4i a, sim_v4i b, sim_v16i c) {
  uint64_t av[2][64], bv[2][64];
  uint64_t mask = 0;
  for (int half = 0; half < 2; ++half) {
    const uint64_t am = ((uint64_t)(uint32_t)a[2 * half + 1] << 32) | (uint32_t)a[2 * half];
    const uint64_t bm = ((uint64_t)(uint32_t)b[2 * half + 1] << 32) | (uint32_t)b[2 * half];
    const Gathered ga = wave_gather(site, am);
    for (int i = 0; i < 64; ++i) av[half][i] = ga.v[i];
    const Gathered gb = wave_gather(site, bm);
    for (int i = 0; i < 64; ++i) bv[half][i] = gb.v[i];
    mask = gb.mask;
  }
  if (mask != ~0ull) fail("mfma with inactive lanes", site);
  const int l = lane_in_wave(), n = l & 31;
  sim_v16i d = c;
  for (int i = 0; i < 16; ++i) {
    const int m = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
    int sum = 0;
    for (int hh = 0; hh < 2; ++hh)
      for (int j = 0; j < 16; ++j) {
        const int ab = (int)(int8_t)(av[j >> 3][m + 32 * hh] >> (8 * (j & 7)));
        const int bb = (int)(int8_t)(bv[j >> 3][n + 32 * hh] >> (8 * (j & 7)));
        sum += ab * bb;
      }
    d[i] += sum;
  }
  return d;
}
#define __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, x, y, z) hostsim::mfma_i32_32x32x32_i8(__LINE__, (a), (b), (c))

// v_mfma_scale_f32_32x32x64_f8f6f4 with both operands FP4 (e2m1; cbsz = blgp = 4) and both block
// scales the same power of two in every lane (E8M0 bytes, 2^(e - 127)): D = scale_a scale_b A (32 x 64) * B (64 x 32) + C in f32.  Lane l holds row l & 31 of A and column
// l & 31 of B for the 32 k values 32 (l >> 5) .. + 31, one per nibble of its first four dwords;
// C / D as for the 32 x 32 integer form.  Other formats and scales are not emulated.
typedef int sim_v8i __attribute__((ext_vector_type(8)));
typedef float sim_v16f __attribute__((ext_vector_type(16)));
inline float fp4_e2m1(unsigned c) {
  const float mag[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  return (c & 8) ? -mag[c & 7] : mag[c & 7];
}
inline sim_v16f mfma_scale_f32_32x32x64_fp4(int site, sim_v8i a, sim_v8i b, sim_v16f c, int cbsz, int blgp, int opa,
                                            int sa, int opb, int sb) {
  if (cbsz != 4 || blgp != 4 || opa != 0 || opb != 0) fail("scaled mfma: only FP4 operands are emulated", site);
  for (int k = 1; k < 4; ++k)
    if (((sa >> (8 * k)) & 0xFF) != (sa & 0xFF) || ((sb >> (8 * k)) & 0xFF) != (sb & 0xFF))
      fail("scaled mfma: the four scale bytes of a lane must be equal here", site);
  const float scale = ldexpf(1.0f, (int)(sa & 0xFF) - 127 + (int)(sb & 0xFF) - 127);
  uint64_t av[2][64], bv[2][64];
  uint64_t mask = 0;
  for (int half = 0; half < 2; ++half) {
    const uint64_t am = ((uint64_t)(uint32_t)a[2 * half + 1] << 32) | (uint32_t)a[2 * half];
    const uint64_t bm = ((uint64_t)(uint32_t)b[2 * half + 1] << 32) | (uint32_t)b[2 * half];
    const Gathered ga = wave_gather(site, am);
    for (int i = 0; i < 64; ++i) av[half][i] = ga.v[i];
    const Gathered gb = wave_gather(site, bm);
    for (int i = 0; i < 64; ++i) bv[half][i] = gb.v[i];
    mask = gb.mask;
  }
  if (mask != ~0ull) fail("mfma with inactive lanes", site);
  const int l = lane_in_wave(), n = l & 31;
  sim_v16f d = c;
  for (int i = 0; i < 16; ++i) {
    const int m = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
    float sum = 0.f;
    for (int hh = 0; hh < 2; ++hh)
      for (int j = 0; j < 32; ++j) {
        const float ab = fp4_e2m1((unsigned)(av[j >> 4][m + 32 * hh] >> (4 * (j & 15))) & 15u);
        const float bb = fp4_e2m1((unsigned)(bv[j >> 4][n + 32 * hh] >> (4 * (j & 15))) & 15u);
        sum += ab * bb;
      }
    d[i] += scale * sum;
  }
  return d;
}
#define __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, cbsz, blgp, opa, sa, opb, sb) \
  hostsim::mfma_scale_f32_32x32x64_fp4(__LINE__, (a), (b), (c), (cbsz), (blgp), (opa), (sa), (opb), (sb))

// v_alignbyte_b32: bytes [sh, sh+4) of the 8-byte value hi:lo
inline uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3)));
}
// v_perm_b32: selector byte 0..3 picks from the second operand, 4..7 from the first,
// 8..11 replicate the sign of byte 1/3/5/7, 12 gives 0x00, 13..255 give 0xff
inline uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) {
  const uint64_t src = (((uint64_t)s0) << 32) | s1;
  uint32_t out = 0;
  for (int i = 0; i < 4; ++i) {
    const uint32_t s = (sel >> (8 * i)) & 0xFF;
    uint32_t b;
    if (s <= 7) b = (uint32_t)(src >> (8 * s)) & 0xFF;
    else if (s <= 11) b = ((src >> (16 * (s - 8) + 15)) & 1) ? 0xFF : 0x00;
    else if (s == 12) b = 0x00;
    else b = 0xFF;
    out |= b << (8 * i);
  }
  return out;
}
typedef unsigned short sim_us2 __attribute__((ext_vector_type(2)));
inline uint32_t udot2(sim_us2 a, sim_us2 b, uint32_t c, bool) {
  return (uint32_t)a.x * (uint32_t)b.x + (uint32_t)a.y * (uint32_t)b.y + c;
}
inline uint32_t udot4(uint32_t a, uint32_t b, uint32_t c, bool) {
  uint32_t r = c;
  for (int i = 0; i < 4; ++i) r += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
  return r;
}
inline uint32_t mbcnt_lo(uint32_t m, uint32_t v) {
  const int lane = lane_in_wave();
  const uint32_t below = lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1u);
  return v + (uint32_t)__builtin_popcount(m & below);
}
inline uint32_t mbcnt_hi(uint32_t m, uint32_t v) {
  const int lane = lane_in_wave();
  const uint32_t below = lane <= 32 ? 0u : (lane >= 64 ? 0xFFFFFFFFu : ((1u << (lane - 32)) - 1u));
  return v + (uint32_t)__builtin_popcount(m & below);
}

}  // namespace hostsim

#define threadIdx (hostsim::thread_idx())
#define blockIdx (hostsim::block_idx())
#define blockDim (hostsim::block_dim())
#define gridDim (hostsim::grid_dim())
#define warpSize 64

// kernel arguments are evaluated once, on the host, like a real launch does
namespace hostsim {
template <class K, class... A>
inline void launch_kernel(const char* name, dim3 grid, dim3 block, size_t lds, K kern, A... args) {
  auto packed = std::make_tuple(args...);
  launch(name, (const void*)kern, grid, block, lds, [&]() { std::apply(kern, packed); });
}
}  // namespace hostsim
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
  hostsim::launch_kernel(#kern, dim3(grid), dim3(block), (size_t)(lds), kern, __VA_ARGS__)

// ---- cross-lane operations (the call site is the source line) -----------------------------
#define __shfl(...) hostsim::shfl(__LINE__, __VA_ARGS__)
#define __shfl_up(...) hostsim::shfl_up(__LINE__, __VA_ARGS__)
#define __shfl_down(...) hostsim::shfl_down(__LINE__, __VA_ARGS__)
#define __shfl_xor(...) hostsim::shfl_xor(__LINE__, __VA_ARGS__)
#define __ballot(p) hostsim::ballot(__LINE__, (p))
#define __any(p) (hostsim::ballot(__LINE__, (p)) != 0)
#define __all(p) (hostsim::ballot(__LINE__, !(p)) == 0)
#define __syncthreads() hostsim::block_barrier(__LINE__)
#define __syncthreads_or(p) hostsim::block_barrier_or(__LINE__, (p))
#define __builtin_amdgcn_readfirstlane(v) hostsim::readfirstlane(__LINE__, (v))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) \
  ((int)hostsim::update_dpp(__LINE__, (uint32_t)(old), (uint32_t)(src), (ctrl), (rm), (bm), (bc)))
#define __builtin_amdgcn_readlane(v, l) hostsim::shfl(__LINE__, (v), (l))
#define __builtin_amdgcn_wave_barrier() hostsim::wave_barrier(__LINE__)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_barrier() hostsim::block_barrier(__LINE__)

#define __builtin_amdgcn_alignbyte hostsim::alignbyte
#define __builtin_amdgcn_perm hostsim::perm
#define __builtin_amdgcn_udot2 hostsim::udot2
#define __builtin_amdgcn_udot4 hostsim::udot4
#define __builtin_amdgcn_mbcnt_lo hostsim::mbcnt_lo
#define __builtin_amdgcn_mbcnt_hi hostsim::mbcnt_hi

// ---- device math / integer intrinsics ------------------------------------------------------
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline int __float2int_rn(float v) { return (int)lrintf(v); }
static inline int __float2int_rd(float v) { return (int)floorf(v); }
static inline int __float2int_ru(float v) { return (int)ceilf(v); }
static inline int __float2int_rz(float v) { return (int)v; }
static inline int __mul24(int a, int b) {
  const int a24 = (int)((uint32_t)a << 8) >> 8, b24 = (int)((uint32_t)b << 8) >> 8;   // sign-extended low 24 bits
  return (int)((uint32_t)a24 * (uint32_t)b24);
}
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31));
}
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline float __uint_as_float(uint32_t v) { float f; memcpy(&f, &v, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t v; memcpy(&v, &f, 4); return v; }

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }

#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))

// ---- atomics (relaxed, like the device ones) ------------------------------------------------
template <class T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicXor(T* p, T v) { return __atomic_fetch_xor(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return cmp;
}
template <class T> static inline T atomicMin(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <class T> static inline T atomicMax(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}

#endif
