// Measures issue cost of individual VALU instructions on gfx950 relative to v_add_u32:
// one wave per SIMD (256 CUs x 4), 8 independent dependency chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define ITERS 4096

#define DEFINE_KERNEL(NAME, ASM)                                                          \
  __global__ __launch_bounds__(64) void NAME(uint32_t* out, uint32_t seed) {              \
    uint32_t a[CHAINS], b = seed | 1u, c = seed * 3u + 5u;                                \
    for (int i = 0; i < CHAINS; ++i) a[i] = threadIdx.x * 7u + i + seed;                   \
    for (int it = 0; it < ITERS; ++it) {                                                   \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i)                                   \
        asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));                                   \
    }                                                                                      \
    uint32_t s = 0;                                                                        \
    for (int i = 0; i < CHAINS; ++i) s ^= a[i];                                            \
    out[blockIdx.x * 64 + threadIdx.x] = s;                                                \
  }

DEFINE_KERNEL(k_add, "v_add_u32 %0, %0, %1")
DEFINE_KERNEL(k_dot2, "v_dot2_u32_u16 %0, %1, %2, %0")
DEFINE_KERNEL(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
DEFINE_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEFINE_KERNEL(k_align, "v_alignbyte_b32 %0, %0, %1, %2")
DEFINE_KERNEL(k_mulhi24, "v_mul_hi_u32_u24 %0, %0, %1")
DEFINE_KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
DEFINE_KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
DEFINE_KERNEL(k_mulhi32, "v_mul_hi_u32 %0, %0, %1")
DEFINE_KERNEL(k_mullo32, "v_mul_lo_u32 %0, %0, %1")
DEFINE_KERNEL(k_pkmin, "v_pk_min_u16 %0, %0, %1")
DEFINE_KERNEL(k_pkadd, "v_pk_add_u16 %0, %0, %1")
DEFINE_KERNEL(k_pksub, "v_pk_sub_i16 %0, %0, %1")
DEFINE_KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
DEFINE_KERNEL(k_min3, "v_min3_i32 %0, %0, %1, %2")
DEFINE_KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 1, %1")
DEFINE_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEFINE_KERNEL(k_sad, "v_sad_u8 %0, %1, %2, %0")
DEFINE_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEFINE_KERNEL(k_and, "v_and_b32 %0, %0, %1")
DEFINE_KERNEL(k_lshr, "v_lshrrev_b32 %0, 1, %0")
DEFINE_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_KERNEL(k_min, "v_min_i32 %0, %0, %1")
DEFINE_KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEFINE_KERNEL(k_mulf, "v_mul_f32 %0, %0, %1")
DEFINE_KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
DEFINE_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 8")
DEFINE_KERNEL(k_madi64, "v_mad_u32_u16 %0, %0, %1, %2")
DEFINE_KERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %2")
DEFINE_KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
DEFINE_KERNEL(k_sub, "v_sub_u32 %0, %1, %0")
DEFINE_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
DEFINE_KERNEL(k_maxu, "v_max_u32 %0, %0, %1")
DEFINE_KERNEL(k_ffbl, "v_ffbl_b32 %0, %0")
DEFINE_KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
DEFINE_KERNEL(k_max3, "v_max3_i32 %0, %0, %1, %2")
DEFINE_KERNEL(k_addlit, "v_add_u32 %0, 0x7f7f7f89, %0")
DEFINE_KERNEL(k_andlit, "v_and_b32 %0, 0x80808080, %0")
DEFINE_KERNEL(k_med3, "v_med3_i32 %0, %0, %1, %2")
DEFINE_KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
DEFINE_KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1")
DEFINE_KERNEL(k_addlshl, "v_add_lshl_u32 %0, %0, %1, 1")
DEFINE_KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
DEFINE_KERNEL(k_not, "v_not_b32 %0, %0")
DEFINE_KERNEL(k_or, "v_or_b32 %0, %0, %1")
DEFINE_KERNEL(k_mov, "v_mov_b32 %0, %1")
DEFINE_KERNEL(k_cmpcnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEFINE_KERNEL(k_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
DEFINE_KERNEL(k_dpp, "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(k_pkmaxi, "v_pk_max_i16 %0, %0, %1")
DEFINE_KERNEL(k_pklshl, "v_pk_lshlrev_b16 %0, 1, %0")

DEFINE_KERNEL(k_rndne, "v_rndne_f32 %0, %0")
DEFINE_KERNEL(k_cvti, "v_cvt_i32_f32 %0, %0")
DEFINE_KERNEL(k_cvtsdwa, "v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1")
DEFINE_KERNEL(k_cvtub, "v_cvt_f32_ubyte1 %0, %0")
DEFINE_KERNEL(k_addf, "v_add_f32 %0, %0, %1")

// the same for instructions on register pairs
#define DEFINE_KERNEL64(NAME, ASM)                                                        \
  __global__ __launch_bounds__(64) void NAME(uint32_t* out, uint32_t seed) {              \
    double a[CHAINS], b = 1.0 + seed * 1e-9, c = 0.5 + seed * 1e-9;                        \
    for (int i = 0; i < CHAINS; ++i) a[i] = threadIdx.x * 1e-3 + i + seed;                 \
    for (int it = 0; it < ITERS; ++it) {                                                   \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i)                                   \
        asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));                                   \
    }                                                                                      \
    double s = 0;                                                                          \
    for (int i = 0; i < CHAINS; ++i) s += a[i];                                            \
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(long long)s;                           \
  }
DEFINE_KERNEL64(k_pkmulf, "v_pk_mul_f32 %0, %0, %1")
DEFINE_KERNEL64(k_pkaddf, "v_pk_add_f32 %0, %0, %1")
DEFINE_KERNEL64(k_pkfmaf, "v_pk_fma_f32 %0, %0, %1, %2")
DEFINE_KERNEL64(k_addd, "v_add_f64 %0, %0, %1")
DEFINE_KERNEL64(k_muld, "v_mul_f64 %0, %0, %1")
DEFINE_KERNEL64(k_fmad, "v_fma_f64 %0, %0, %1, %2")

struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t); };

int main() {
  uint32_t* d = nullptr;
  const int waves = 256 * 4;
  hipMalloc(&d, waves * 64 * 4);
  Entry es[] = {{"v_add_u32", k_add}, {"v_dot2_u32_u16", k_dot2}, {"v_dot4_u32_u8", k_dot4},
                {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_align}, {"v_mul_hi_u32_u24", k_mulhi24},
                {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_mul_hi_u32", k_mulhi32},
                {"v_mul_lo_u32", k_mullo32}, {"v_pk_min_u16", k_pkmin}, {"v_pk_add_u16", k_pkadd},
                {"v_pk_sub_i16", k_pksub}, {"v_bcnt_u32_b32", k_bcnt}, {"v_min3_i32", k_min3},
                {"v_lshl_or_b32", k_lshlor}, {"v_add3_u32", k_add3}, {"v_sad_u8", k_sad},
                {"v_xor_b32", k_xor}, {"v_and_b32", k_and}, {"v_lshrrev_b32", k_lshr},
                {"v_cndmask_b32", k_cndmask}, {"v_min_i32", k_min}, {"v_fma_f32", k_fma},
                {"v_mul_f32", k_mulf}, {"v_cvt_f32_u32", k_cvt}, {"v_bfe_u32", k_bfe},
                {"v_mad_u32_u16", k_madi64}, {"v_and_or_b32", k_andor}, {"v_or3_b32", k_or3}, {"v_sub_u32", k_sub},
                {"v_alignbit_b32", k_alignbit}, {"v_max_u32", k_maxu}, {"v_ffbl_b32", k_ffbl}, {"v_bfi_b32", k_bfi},
                {"v_max3_i32", k_max3}, {"v_add_u32 literal", k_addlit}, {"v_and_b32 literal", k_andlit},
                {"v_med3_i32", k_med3}, {"v_xad_u32", k_xad}, {"v_lshl_add_u32", k_lshladd},
                {"v_add_lshl_u32", k_addlshl}, {"v_mbcnt_lo", k_mbcnt}, {"v_not_b32", k_not}, {"v_or_b32", k_or},
                {"v_mov_b32", k_mov}, {"v_cmp + v_cndmask (pair)", k_cmpcnd}, {"v_add_u32 sdwa", k_sdwa},
                {"v_add_u32 dpp", k_dpp}, {"v_pk_max_i16", k_pkmaxi}, {"v_pk_lshlrev_b16", k_pklshl},
                {"v_rndne_f32", k_rndne}, {"v_cvt_i32_f32", k_cvti}, {"v_cvt_f32_i32 sdwa", k_cvtsdwa},
                {"v_cvt_f32_ubyte1", k_cvtub}, {"v_add_f32", k_addf}, {"v_pk_mul_f32", k_pkmulf},
                {"v_pk_add_f32", k_pkaddf}, {"v_pk_fma_f32", k_pkfmaf}, {"v_add_f64", k_addd},
                {"v_mul_f64", k_muld}, {"v_fma_f64", k_fmad}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double base = 0;
  for (int occ = 1; occ <= 8; occ *= 2) {
    std::printf("-- %d wave(s) per SIMD --\n", occ);
    for (auto& e : es) {
      hipLaunchKernelGGL(e.fn, dim3(waves * occ), dim3(64), 0, 0, d, 1u);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(e.fn, dim3(waves * occ), dim3(64), 0, 0, d, 1u);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double ns_per_inst = ms * 1e6 / 5 / ((double)ITERS * CHAINS) / occ;
      if (e.fn == k_add) base = ns_per_inst;
      std::printf("%-20s %.3f ns per wave-instruction per SIMD  (%.2f x v_add_u32)\n", e.name, ns_per_inst, ns_per_inst / base);
    }
  }
  return 0;
}
