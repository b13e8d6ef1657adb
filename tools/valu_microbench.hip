// valu_microbench.hip -- issue cost of the VALU instruction classes the march kernels are made of,
// measured on the box it runs on (gfx950), in SHADER CYCLES read inside the kernel.  Standalone:
//   hipcc -O2 --offload-arch=gfx950 tools/valu_microbench.hip -o /tmp/valu_microbench
//   /tmp/valu_microbench [waves_per_simd = 4]
// The grid holds W waves per SIMD of the chip, each executing N wave64 instructions of one class from
// registers (8 dependency chains per wave, so a wave's own latency is covered by the SIMD's other
// waves).  Every wave brackets its loop with s_memtime (shader clock) and s_memrealtime (the 100 MHz
// constant clock) and records the SIMD it ran on (HW_REG_HW_ID, HW_REG_XCC_ID): the dispatcher does
// NOT spread the blocks evenly, so the figure is formed per SIMD from what actually ran there --
// cycles per wave-instruction = (last end - first start on the SIMD) / (waves on the SIMD x N), median
// over the SIMDs that held at least W waves.  The ratio of the two clocks is the shader clock the
// kernel actually ran at (the chip floats with power), printed per case.  Nothing is assumed about
// any instruction's cost.
// Mixed cases (two classes alternating) test whether costs add: a kernel's issue floor is priced as
// sum over classes of count x cost, which is only right if they do.
// Under `rocprofv3 --pmc SQ_INSTS_VALU_*` the same binary calibrates which hardware class counter
// each mnemonic lands in (tools/summarize_profiles.py reads the per-kernel counts).
// Output: one JSON object on stdout (tools/isa_histogram.py --costs reads "cycles").
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)

constexpr int kIters = 1024;
constexpr int kPerIter = 32; // asm statements per loop iteration (8 chains x 4)

struct WaveClock {
    unsigned long long t0, t1, ticks;
    unsigned hw_id, xcc_id; // HW_REG_HW_ID (wave / SIMD / CU / SH / SE the wave ran on), HW_REG_XCC_ID
};

#define TIMED_BEGIN                                                                      \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                          \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#define TIMED_END                                                                        \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                          \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                      \
    if ((threadIdx.x & 63) == 0) {                                                       \
        WaveClock w;                                                                     \
        w.t0 = t0;                                                                       \
        w.t1 = t1;                                                                       \
        w.ticks = r1 - r0;                                                               \
        w.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);                  \
        w.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);                \
        clk[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = w;                           \
    }

#define CLOB "vcc", "s20", "s21"

// The eight chain instructions of one group are ONE asm statement (the compiler pads separate
// statements that clobber SGPRs with s_nop).  X(r) is the instruction on chain register %r.
#define GROUP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 32-bit operand classes.  Chain registers %0..%7; %8 %9 loop-invariant VGPRs, %10 an SGPR operand.
#define KERNEL32(NAME, X)                                                                           \
    __global__ __launch_bounds__(256) void NAME(WaveClock *clk, float *out, float seed) {           \
        float a0 = seed, a1 = seed + 1.f, a2 = seed + 2.f, a3 = seed + 3.f, a4 = seed + 4.f,        \
              a5 = seed + 5.f, a6 = seed + 6.f, a7 = seed + 7.f;                                    \
        const float b = 1.0000001f, c = 1e-9f;                                                      \
        const float sc = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(seed + 0.50001f))); \
        TIMED_BEGIN                                                                                 \
        for (int i = 0; i < kIters; ++i) {                                                          \
            _Pragma("unroll") for (int u = 0; u < 4; ++u)                                           \
                asm volatile(GROUP8(X)                                                              \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                             : "v"(b), "v"(c), "s"(sc)                                              \
                             : CLOB);                                                               \
        }                                                                                           \
        TIMED_END                                                                                   \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0;             \
    }

// 64-bit operand classes (f64 and packed f32: one VGPR pair per operand); same operand numbering
#define KERNEL64(NAME, X)                                                                           \
    __global__ __launch_bounds__(256) void NAME(WaveClock *clk, float *out, float seed) {           \
        double a0 = seed, a1 = seed + 1., a2 = seed + 2., a3 = seed + 3., a4 = seed + 4.,           \
               a5 = seed + 5., a6 = seed + 6., a7 = seed + 7.;                                      \
        const double b = 1.0000001, c = 1e-9;                                                       \
        const int slo = __builtin_amdgcn_readfirstlane(__float_as_int(seed));                       \
        const double sc = __hiloint2double(0x3ff00000 + (slo & 1), slo);                            \
        TIMED_BEGIN                                                                                 \
        for (int i = 0; i < kIters; ++i) {                                                          \
            _Pragma("unroll") for (int u = 0; u < 4; ++u)                                           \
                asm volatile(GROUP8(X)                                                              \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                             : "v"(b), "v"(c), "s"(sc)                                              \
                             : CLOB);                                                               \
        }                                                                                           \
        TIMED_END                                                                                   \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[threadIdx.x] = (float)a0;       \
    }

// one 32-bit and one 64-bit chain register per instruction group: X(f, d) with f in %0..%7 (float),
// d in %8..%15 (double); %16 = loop-invariant VGPR, %17 = loop-invariant VGPR pair.  Conversions and
// the mixed-class cases.
#define GROUP8M(X) X(0, 8) X(1, 9) X(2, 10) X(3, 11) X(4, 12) X(5, 13) X(6, 14) X(7, 15)
#define KERNELMIX(NAME, X)                                                                          \
    __global__ __launch_bounds__(256) void NAME(WaveClock *clk, float *out, float seed) {           \
        float f0 = seed, f1 = seed + 1.f, f2 = seed + 2.f, f3 = seed + 3.f, f4 = seed + 4.f,        \
              f5 = seed + 5.f, f6 = seed + 6.f, f7 = seed + 7.f;                                    \
        double d0 = seed, d1 = seed + 1., d2 = seed + 2., d3 = seed + 3., d4 = seed + 4.,           \
               d5 = seed + 5., d6 = seed + 6., d7 = seed + 7.;                                      \
        const float b = 1.0000001f;                                                                 \
        const double c = 1.0000001;                                                                 \
        TIMED_BEGIN                                                                                 \
        for (int i = 0; i < kIters; ++i) {                                                          \
            _Pragma("unroll") for (int u = 0; u < 4; ++u)                                           \
                asm volatile(GROUP8M(X)                                                             \
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), \
                               "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)  \
                             : "v"(b), "v"(c)                                                       \
                             : CLOB);                                                               \
        }                                                                                           \
        TIMED_END                                                                                   \
        if (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) == 12345.678f) \
            out[threadIdx.x] = f0;                                                                  \
    }

#define R(r) "%" #r
// partner chain of r inside the same width (the second instruction of a same-width mixed case)
#define P(r) P_##r
#define P_0 "%4"
#define P_1 "%5"
#define P_2 "%6"
#define P_3 "%7"
#define P_4 "%0"
#define P_5 "%1"
#define P_6 "%2"
#define P_7 "%3"
#define I_fma_f32(r) "v_fma_f32 " R(r) ", " R(r) ", %8, %9\n"
KERNEL32(k_fma_f32, I_fma_f32)
#define I_fmac_f32(r) "v_fmac_f32 " R(r) ", %8, %9\n"
KERNEL32(k_fmac_f32, I_fmac_f32)
#define I_fma_f32_sgpr(r) "v_fma_f32 " R(r) ", " R(r) ", %10, %9\n"
KERNEL32(k_fma_f32_sgpr, I_fma_f32_sgpr)
#define I_fmaak_f32(r) "v_fmaak_f32 " R(r) ", " R(r) ", %8, 0x3f800100\n"
KERNEL32(k_fmaak_f32, I_fmaak_f32)
#define I_mul_f32(r) "v_mul_f32 " R(r) ", " R(r) ", %8\n"
KERNEL32(k_mul_f32, I_mul_f32)
#define I_mul_f32_sgpr(r) "v_mul_f32 " R(r) ", %10, " R(r) "\n"
KERNEL32(k_mul_f32_sgpr, I_mul_f32_sgpr)
#define I_mul_f32_lit(r) "v_mul_f32 " R(r) ", 0x3f800100, " R(r) "\n"
KERNEL32(k_mul_f32_lit, I_mul_f32_lit)
#define I_add_f32(r) "v_add_f32 " R(r) ", " R(r) ", %9\n"
KERNEL32(k_add_f32, I_add_f32)
#define I_sub_f32(r) "v_sub_f32 " R(r) ", " R(r) ", %9\n"
KERNEL32(k_sub_f32, I_sub_f32)
#define I_max_f32(r) "v_max_f32 " R(r) ", " R(r) ", %9\n"
KERNEL32(k_max_f32, I_max_f32)
#define I_rcp_f32(r) "v_rcp_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_rcp_f32, I_rcp_f32)
#define I_rsq_f32(r) "v_rsq_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_rsq_f32, I_rsq_f32)
#define I_sqrt_f32(r) "v_sqrt_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_sqrt_f32, I_sqrt_f32)
#define I_exp_f32(r) "v_exp_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_exp_f32, I_exp_f32)
#define I_log_f32(r) "v_log_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_log_f32, I_log_f32)
#define I_sin_f32(r) "v_sin_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_sin_f32, I_sin_f32)
#define I_cvt_i32_f32(r) "v_cvt_i32_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_cvt_i32_f32, I_cvt_i32_f32)
#define I_cvt_f32_i32(r) "v_cvt_f32_i32 " R(r) ", " R(r) "\n"
KERNEL32(k_cvt_f32_i32, I_cvt_f32_i32)
#define I_fract_f32(r) "v_fract_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_fract_f32, I_fract_f32)
#define I_floor_f32(r) "v_floor_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_floor_f32, I_floor_f32)
#define I_trunc_f32(r) "v_trunc_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_trunc_f32, I_trunc_f32)
#define I_rndne_f32(r) "v_rndne_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_rndne_f32, I_rndne_f32)
#define I_frexp_mant_f32(r) "v_frexp_mant_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_frexp_mant_f32, I_frexp_mant_f32)
#define I_frexp_exp_i32_f32(r) "v_frexp_exp_i32_f32 " R(r) ", " R(r) "\n"
KERNEL32(k_frexp_exp_i32_f32, I_frexp_exp_i32_f32)
#define I_ldexp_f32(r) "v_ldexp_f32 " R(r) ", " R(r) ", 1\n"
KERNEL32(k_ldexp_f32, I_ldexp_f32)
#define I_med3_f32(r) "v_med3_f32 " R(r) ", " R(r) ", %8, %9\n"
KERNEL32(k_med3_f32, I_med3_f32)
#define I_mov_b32(r) "v_mov_b32 " R(r) ", %8\n"
KERNEL32(k_mov_b32, I_mov_b32)
#define I_mov_b32_sgpr(r) "v_mov_b32 " R(r) ", %10\n"
KERNEL32(k_mov_b32_sgpr, I_mov_b32_sgpr)
#define I_add_u32(r) "v_add_u32 " R(r) ", " R(r) ", %8\n"
KERNEL32(k_add_u32, I_add_u32)
#define I_sub_u32(r) "v_sub_u32 " R(r) ", " R(r) ", %8\n"
KERNEL32(k_sub_u32, I_sub_u32)
#define I_and_b32(r) "v_and_b32 " R(r) ", " R(r) ", %8\n"
KERNEL32(k_and_b32, I_and_b32)
#define I_xor_b32(r) "v_xor_b32 " R(r) ", " R(r) ", %8\n"
KERNEL32(k_xor_b32, I_xor_b32)
#define I_or_b32(r) "v_or_b32 " R(r) ", " R(r) ", %8\n"
KERNEL32(k_or_b32, I_or_b32)
#define I_lshl_b32(r) "v_lshlrev_b32 " R(r) ", 1, " R(r) "\n"
KERNEL32(k_lshl_b32, I_lshl_b32)
#define I_lshr_b32(r) "v_lshrrev_b32 " R(r) ", 1, " R(r) "\n"
KERNEL32(k_lshr_b32, I_lshr_b32)
#define I_alignbit_b32(r) "v_alignbit_b32 " R(r) ", " R(r) ", %8, 3\n"
KERNEL32(k_alignbit_b32, I_alignbit_b32)
#define I_bfe_u32(r) "v_bfe_u32 " R(r) ", " R(r) ", 3, 5\n"
KERNEL32(k_bfe_u32, I_bfe_u32)
#define I_and_or_b32(r) "v_and_or_b32 " R(r) ", " R(r) ", %8, %9\n"
KERNEL32(k_and_or_b32, I_and_or_b32)
#define I_mul_lo_u32(r) "v_mul_lo_u32 " R(r) ", " R(r) ", %8\n"
KERNEL32(k_mul_lo_u32, I_mul_lo_u32)
#define I_mul_hi_u32(r) "v_mul_hi_u32 " R(r) ", " R(r) ", %8\n"
KERNEL32(k_mul_hi_u32, I_mul_hi_u32)
#define I_cndmask_b32(r) "v_cndmask_b32 " R(r) ", " R(r) ", %8, vcc\n"
KERNEL32(k_cndmask_b32, I_cndmask_b32)
#define I_cndmask_b32_sgpr(r) "v_cndmask_b32_e64 " R(r) ", " R(r) ", %8, s[20:21]\n"
KERNEL32(k_cndmask_b32_sgpr, I_cndmask_b32_sgpr)
#define I_cndmask_b32_e64vcc(r) "v_cndmask_b32_e64 " R(r) ", " R(r) ", %8, vcc\n"
KERNEL32(k_cndmask_b32_e64vcc, I_cndmask_b32_e64vcc)
#define I_cmp_cndmask(r) "v_cmp_lt_f32 vcc, " R(r) ", %8\n v_cndmask_b32 " R(r) ", " R(r) ", %9, vcc\n"
KERNEL32(k_mix_cmp_cndmask, I_cmp_cndmask)
#define I_cmp_f32(r) "v_cmp_lt_f32 vcc, " R(r) ", %8\n"
KERNEL32(k_cmp_f32, I_cmp_f32)
#define I_cmp_f32_sgpr(r) "v_cmp_lt_f32 s[20:21], " R(r) ", %8\n"
KERNEL32(k_cmp_f32_sgpr, I_cmp_f32_sgpr)
#define I_cmp_class_f32(r) "v_cmp_class_f32 s[20:21], " R(r) ", %8\n"
KERNEL32(k_cmp_class_f32, I_cmp_class_f32)
#define I_cmp_u32(r) "v_cmp_lt_u32 vcc, " R(r) ", %8\n"
KERNEL32(k_cmp_u32, I_cmp_u32)
#define I_readlane(r) "v_readlane_b32 s20, " R(r) ", 3\n"
KERNEL32(k_readlane, I_readlane)
#define I_writelane(r) "v_writelane_b32 " R(r) ", s20, 3\n"
KERNEL32(k_writelane, I_writelane)
#define I_readfirstlane(r) "v_readfirstlane_b32 s20, " R(r) "\n"
KERNEL32(k_readfirstlane, I_readfirstlane)
#define I_nop(r) "v_nop\n"
KERNEL32(k_nop, I_nop)
// ---- 64-bit operands: packed f32, f64 ----
#define I_pk_fma_f32(r) "v_pk_fma_f32 " R(r) ", " R(r) ", %8, %9\n"
KERNEL64(k_pk_fma_f32, I_pk_fma_f32)
#define I_pk_mul_f32(r) "v_pk_mul_f32 " R(r) ", " R(r) ", %8\n"
KERNEL64(k_pk_mul_f32, I_pk_mul_f32)
#define I_pk_add_f32(r) "v_pk_add_f32 " R(r) ", " R(r) ", %9\n"
KERNEL64(k_pk_add_f32, I_pk_add_f32)
#define I_pk_mov_b32(r) "v_pk_mov_b32 " R(r) ", %8, %9\n"
KERNEL64(k_pk_mov_b32, I_pk_mov_b32)
#define I_pk_mul_f32_sgpr(r) "v_pk_mul_f32 " R(r) ", " R(r) ", %10\n"
KERNEL64(k_pk_mul_f32_sgpr, I_pk_mul_f32_sgpr)
#define I_fma_f64(r) "v_fma_f64 " R(r) ", " R(r) ", %8, %9\n"
KERNEL64(k_fma_f64, I_fma_f64)
#define I_fmac_f64(r) "v_fmac_f64 " R(r) ", %8, %9\n"
KERNEL64(k_fmac_f64, I_fmac_f64)
#define I_fma_f64_sgpr(r) "v_fma_f64 " R(r) ", " R(r) ", %10, %9\n"
KERNEL64(k_fma_f64_sgpr, I_fma_f64_sgpr)
#define I_mul_f64(r) "v_mul_f64 " R(r) ", " R(r) ", %8\n"
KERNEL64(k_mul_f64, I_mul_f64)
#define I_mul_f64_sgpr(r) "v_mul_f64 " R(r) ", " R(r) ", %10\n"
KERNEL64(k_mul_f64_sgpr, I_mul_f64_sgpr)
#define I_add_f64(r) "v_add_f64 " R(r) ", " R(r) ", %9\n"
KERNEL64(k_add_f64, I_add_f64)
#define I_max_f64(r) "v_max_f64 " R(r) ", " R(r) ", %9\n"
KERNEL64(k_max_f64, I_max_f64)
#define I_rcp_f64(r) "v_rcp_f64 " R(r) ", " R(r) "\n"
KERNEL64(k_rcp_f64, I_rcp_f64)
#define I_rsq_f64(r) "v_rsq_f64 " R(r) ", " R(r) "\n"
KERNEL64(k_rsq_f64, I_rsq_f64)
#define I_sqrt_f64(r) "v_sqrt_f64 " R(r) ", " R(r) "\n"
KERNEL64(k_sqrt_f64, I_sqrt_f64)
#define I_mov_b64(r) "v_mov_b64 " R(r) ", %8\n"
KERNEL64(k_mov_b64, I_mov_b64)
#define I_mov_b64_sgpr(r) "v_mov_b64 " R(r) ", %10\n"
KERNEL64(k_mov_b64_sgpr, I_mov_b64_sgpr)
#define I_cmp_f64(r) "v_cmp_lt_f64 vcc, " R(r) ", %8\n"
KERNEL64(k_cmp_f64, I_cmp_f64)
#define I_cmp_class_f64(r) "v_cmp_class_f64 s[20:21], " R(r) ", 3\n"
KERNEL64(k_cmp_class_f64, I_cmp_class_f64)
#define I_fract_f64(r) "v_fract_f64 " R(r) ", " R(r) "\n"
KERNEL64(k_fract_f64, I_fract_f64)
#define I_rndne_f64(r) "v_rndne_f64 " R(r) ", " R(r) "\n"
KERNEL64(k_rndne_f64, I_rndne_f64)
#define I_floor_f64(r) "v_floor_f64 " R(r) ", " R(r) "\n"
KERNEL64(k_floor_f64, I_floor_f64)
#define I_ldexp_f64(r) "v_ldexp_f64 " R(r) ", " R(r) ", 1\n"
KERNEL64(k_ldexp_f64, I_ldexp_f64)
#define I_frexp_mant_f64(r) "v_frexp_mant_f64 " R(r) ", " R(r) "\n"
KERNEL64(k_frexp_mant_f64, I_frexp_mant_f64)
#define I_div_scale_f64(r) "v_div_scale_f64 " R(r) ", vcc, " R(r) ", %8, " R(r) "\n"
KERNEL64(k_div_scale_f64, I_div_scale_f64)
#define I_div_fmas_f64(r) "v_div_fmas_f64 " R(r) ", " R(r) ", %8, %9\n"
KERNEL64(k_div_fmas_f64, I_div_fmas_f64)
#define I_div_fixup_f64(r) "v_div_fixup_f64 " R(r) ", " R(r) ", %8, %9\n"
KERNEL64(k_div_fixup_f64, I_div_fixup_f64)
#define I_trig_preop_f64(r) "v_trig_preop_f64 " R(r) ", " R(r) ", 1\n"
KERNEL64(k_trig_preop_f64, I_trig_preop_f64)
#define I_lshl_b64(r) "v_lshlrev_b64 " R(r) ", 1, " R(r) "\n"
KERNEL64(k_lshl_b64, I_lshl_b64)
// ---- mixed widths and mixed classes (two instructions per group member: do the costs add?) ----
#define I_cvt_f32_f64(f, d) "v_cvt_f32_f64 " R(f) ", " R(d) "\n"
KERNELMIX(k_cvt_f32_f64, I_cvt_f32_f64)
#define I_cvt_f64_f32(f, d) "v_cvt_f64_f32 " R(d) ", " R(f) "\n"
KERNELMIX(k_cvt_f64_f32, I_cvt_f64_f32)
#define I_cvt_i32_f64(f, d) "v_cvt_i32_f64 " R(f) ", " R(d) "\n"
KERNELMIX(k_cvt_i32_f64, I_cvt_i32_f64)
#define I_cvt_f64_i32(f, d) "v_cvt_f64_i32 " R(d) ", " R(f) "\n"
KERNELMIX(k_cvt_f64_i32, I_cvt_f64_i32)
#define I_frexp_exp_i32_f64(f, d) "v_frexp_exp_i32_f64 " R(f) ", " R(d) "\n"
KERNELMIX(k_frexp_exp_i32_f64, I_frexp_exp_i32_f64)
#define I_mad_u64_u32(f, d) "v_mad_u64_u32 " R(d) ", vcc, " R(f) ", %16, " R(d) "\n"
KERNELMIX(k_mad_u64_u32, I_mad_u64_u32)
#define I_mix_fma32_rcp32(r) "v_fma_f32 " R(r) ", " R(r) ", %8, %9\n v_rcp_f32 " P(r) "" ", " P(r) "\n"
KERNEL32(k_mix_fma32_rcp32, I_mix_fma32_rcp32)
#define I_mix_fma32_mov32(r) "v_fma_f32 " R(r) ", " R(r) ", %8, %9\n v_mov_b32 " P(r) ", %8\n"
KERNEL32(k_mix_fma32_mov32, I_mix_fma32_mov32)
#define I_mix_fma32_cnd(r) "v_fma_f32 " R(r) ", " R(r) ", %8, %9\n v_cndmask_b32 " P(r) "" ", " P(r) ", %8, vcc\n"
KERNEL32(k_mix_fma32_cnd, I_mix_fma32_cnd)
#define I_mix_fma32_cvt(r) "v_fma_f32 " R(r) ", " R(r) ", %8, %9\n v_cvt_i32_f32 " P(r) "" ", " P(r) "\n"
KERNEL32(k_mix_fma32_cvt, I_mix_fma32_cvt)
#define I_mix_pkfma_fma32(f, d) "v_pk_fma_f32 " R(d) ", " R(d) ", %17, %17\n v_fma_f32 " R(f) ", " R(f) ", %16, %16\n"
KERNELMIX(k_mix_pkfma_fma32, I_mix_pkfma_fma32)
#define I_mix_pkfma_cnd(f, d) "v_pk_fma_f32 " R(d) ", " R(d) ", %17, %17\n v_cndmask_b32 " R(f) ", " R(f) ", %16, vcc\n"
KERNELMIX(k_mix_pkfma_cnd, I_mix_pkfma_cnd)
#define I_mix_pkfma_rcp32(f, d) "v_pk_fma_f32 " R(d) ", " R(d) ", %17, %17\n v_rcp_f32 " R(f) ", " R(f) "\n"
KERNELMIX(k_mix_pkfma_rcp32, I_mix_pkfma_rcp32)
#define I_mix_pkfma_mov32(f, d) "v_pk_fma_f32 " R(d) ", " R(d) ", %17, %17\n v_mov_b32 " R(f) ", %16\n"
KERNELMIX(k_mix_pkfma_mov32, I_mix_pkfma_mov32)
#define I_mix_fma64_fma32(f, d) "v_fma_f64 " R(d) ", " R(d) ", %17, %17\n v_fma_f32 " R(f) ", " R(f) ", %16, %16\n"
KERNELMIX(k_mix_fma64_fma32, I_mix_fma64_fma32)
#define I_mix_fma64_mov32(f, d) "v_fma_f64 " R(d) ", " R(d) ", %17, %17\n v_mov_b32 " R(f) ", %16\n"
KERNELMIX(k_mix_fma64_mov32, I_mix_fma64_mov32)
#define I_mix_fma64_rcp64(f, d) "v_fma_f64 " R(d) ", " R(d) ", %17, %17\n v_rcp_f64 " R(d) ", " R(d) "\n"
KERNELMIX(k_mix_fma64_rcp64, I_mix_fma64_rcp64)
#define I_mix_fma64_cnd(f, d) "v_fma_f64 " R(d) ", " R(d) ", %17, %17\n v_cndmask_b32 " R(f) ", " R(f) ", %16, vcc\n"
KERNELMIX(k_mix_fma64_cnd, I_mix_fma64_cnd)
#define I_mix_fma64_readlane(f, d) "v_fma_f64 " R(d) ", " R(d) ", %17, %17\n v_readlane_b32 s20, " R(f) ", 3\n"
KERNELMIX(k_mix_fma64_readlane, I_mix_fma64_readlane)
#define I_mix_fma64_salu(f, d) "v_fma_f64 " R(d) ", " R(d) ", %17, %17\n s_add_u32 s20, s20, 1\n"
KERNELMIX(k_mix_fma64_salu, I_mix_fma64_salu)
#define I_mix_fma32_salu(f, d) "v_fma_f32 " R(f) ", " R(f) ", %16, %16\n s_add_u32 s20, s20, 1\n"
KERNELMIX(k_mix_fma32_salu, I_mix_fma32_salu)

struct Case {
    const char *name;
    void (*fn)(WaveClock *, float *, float);
    int insts_per_stmt;
};

int main(int argc, char **argv) {
    const int waves_per_simd = argc > 1 ? std::atoi(argv[1]) : 4;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    float *out;
    CHECK(hipMalloc(&out, 4096));
    const std::vector<Case> cases = {
#define C(n) {#n, k_##n, 1}
#define C2(n) {#n, k_##n, 2}
        C(fma_f32), C(fmac_f32), C(fma_f32_sgpr), C(fmaak_f32), C(mul_f32), C(mul_f32_sgpr), C(mul_f32_lit),
        C(add_f32), C(sub_f32), C(max_f32), C(rcp_f32), C(rsq_f32), C(sqrt_f32), C(exp_f32), C(log_f32), C(sin_f32),
        C(cvt_i32_f32), C(cvt_f32_i32), C(fract_f32), C(floor_f32), C(trunc_f32), C(rndne_f32), C(frexp_mant_f32),
        C(frexp_exp_i32_f32), C(ldexp_f32), C(med3_f32), C(mov_b32), C(mov_b32_sgpr), C(add_u32), C(sub_u32),
        C(and_b32), C(xor_b32), C(or_b32), C(lshl_b32), C(lshr_b32), C(alignbit_b32), C(bfe_u32), C(and_or_b32),
        C(mul_lo_u32), C(mul_hi_u32), C(cndmask_b32), C(cndmask_b32_sgpr), C(cndmask_b32_e64vcc), C(cmp_f32), C(cmp_f32_sgpr),
        C(cmp_class_f32), C(cmp_u32), C(readlane), C(writelane), C(readfirstlane), C(nop), C(pk_fma_f32),
        C(pk_mul_f32), C(pk_add_f32), C(pk_mov_b32), C(pk_mul_f32_sgpr), C(fma_f64), C(fmac_f64), C(fma_f64_sgpr),
        C(mul_f64), C(mul_f64_sgpr), C(add_f64), C(max_f64), C(rcp_f64), C(rsq_f64), C(sqrt_f64), C(mov_b64),
        C(mov_b64_sgpr), C(cmp_f64), C(cmp_class_f64), C(fract_f64), C(rndne_f64), C(floor_f64), C(ldexp_f64),
        C(frexp_mant_f64), C(div_scale_f64), C(div_fmas_f64), C(div_fixup_f64), C(trig_preop_f64), C(lshl_b64),
        C(cvt_f32_f64), C(cvt_f64_f32), C(cvt_i32_f64), C(cvt_f64_i32), C(frexp_exp_i32_f64), C(mad_u64_u32),
        C2(mix_cmp_cndmask), C2(mix_fma32_rcp32), C2(mix_fma32_mov32), C2(mix_fma32_cnd), C2(mix_fma32_cvt), C2(mix_pkfma_fma32),
        C2(mix_pkfma_cnd), C2(mix_pkfma_rcp32), C2(mix_pkfma_mov32), C2(mix_fma64_fma32), C2(mix_fma64_mov32),
        C2(mix_fma64_rcp64), C2(mix_fma64_cnd), C2(mix_fma64_readlane), C2(mix_fma64_salu), C2(mix_fma32_salu),
#undef C
#undef C2
    };
    // one block = 4 waves = one wave per SIMD of a CU; W blocks per CU
    const int n_blocks = n_cu * waves_per_simd, n_waves = n_blocks * 4;
    WaveClock *clk;
    CHECK(hipMalloc(&clk, sizeof(WaveClock) * n_waves));
    std::vector<WaveClock> h(n_waves);
    std::vector<double> cyc(cases.size()), ghz(cases.size()), lo(cases.size()), hi(cases.size());
    int simds_seen = 0;
    double waves_per_simd_mean = 0;
    for (size_t k = 0; k < cases.size(); ++k) {
        for (int rep = 0; rep < 3; ++rep) { // the first launches warm up (clock ramp, code fetch)
            hipLaunchKernelGGL(cases[k].fn, dim3(n_blocks), dim3(256), 0, 0, clk, out, 0.5f);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), clk, sizeof(WaveClock) * n_waves, hipMemcpyDeviceToHost));
        // group the waves by the SIMD they ran on
        std::map<unsigned long long, std::vector<int>> by_simd;
        double c = 0, t = 0;
        for (int w = 0; w < n_waves; ++w) {
            // HW_ID: SIMD_ID [5:4], CU_ID [11:8], SH_ID [12], SE_ID [15:13]; XCC_ID [3:0]
            const unsigned long long key = ((unsigned long long)(h[w].xcc_id & 0xf) << 32) | (h[w].hw_id & 0xff30u);
            by_simd[key].push_back(w);
            c += (double)(h[w].t1 - h[w].t0);
            t += (double)h[w].ticks;
        }
        std::vector<double> per;
        double wsum = 0;
        for (auto &kv : by_simd) {
            unsigned long long a = ~0ull, b = 0;
            for (int w : kv.second) {
                a = std::min(a, h[w].t0);
                b = std::max(b, h[w].t1);
            }
            wsum += (double)kv.second.size();
            if ((int)kv.second.size() >= waves_per_simd)
                per.push_back((double)(b - a) / ((double)kv.second.size() * kIters * kPerIter));
        }
        if (per.empty()) per.push_back(0.0);
        std::sort(per.begin(), per.end());
        const int n = (int)per.size();
        cyc[k] = per[n / 2];
        lo[k] = per[n / 20];
        hi[k] = per[n - 1 - n / 20];
        simds_seen = (int)by_simd.size();
        waves_per_simd_mean = wsum / (double)by_simd.size();
        ghz[k] = c / (t * 10.0); // ticks of the 100 MHz clock = 10 ns
    }
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"waves_per_simd\": %d, "
                "\"simds_seen\": %d, \"waves_per_simd_mean\": %.2f, "
                "\"method\": \"s_memtime around %d statements per wave; per SIMD (HW_ID): (last end - first start) / "
                "(waves on the SIMD x statements); median over SIMDs holding >= %d waves; p5 / p95 beside it\", ",
                prop.name, n_cu, n_cu * 4, waves_per_simd, simds_seen, waves_per_simd_mean, kIters * kPerIter,
                waves_per_simd);
    std::printf("\"cycles\": {");
    bool first = true;
    for (size_t k = 0; k < cases.size(); ++k)
        if (cases[k].insts_per_stmt == 1) {
            std::printf("%s\"v_%s\": %.3f", first ? "" : ", ", cases[k].name, cyc[k]);
            first = false;
        }
    std::printf("}, \"cycles_p5_p95\": {");
    for (size_t k = 0; k < cases.size(); ++k)
        std::printf("%s\"v_%s\": [%.3f, %.3f]", k ? ", " : "", cases[k].name, lo[k], hi[k]);
    std::printf("}, \"pairs_cycles_per_statement\": {");
    first = true;
    for (size_t k = 0; k < cases.size(); ++k)
        if (cases[k].insts_per_stmt == 2) {
            std::printf("%s\"%s\": %.3f", first ? "" : ", ", cases[k].name, cyc[k]);
            first = false;
        }
    std::printf("}, \"shader_clock_ghz\": {");
    for (size_t k = 0; k < cases.size(); ++k) std::printf("%s\"v_%s\": %.3f", k ? ", " : "", cases[k].name, ghz[k]);
    std::printf("}}\n");
    return 0;
}
