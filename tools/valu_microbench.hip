// valu_microbench.hip -- issue cost of the VALU instruction classes the march kernels are made of,
// measured on the box it runs on (gfx950).  Standalone:
//   hipcc -O2 --offload-arch=gfx950 tools/valu_microbench.hip -o /tmp/valu_microbench && /tmp/valu_microbench
// Every SIMD of the chip runs W waves, each executing N independent wave64 instructions of one class
// from registers (8 dependency chains, so a lone wave's latency is covered by its SIMD's other
// waves); cycles per wave-instruction per SIMD = elapsed * clock / (W * N).  The clock is not known
// a priori (the chip floats with power): results are printed relative to v_fma_f32 == 2 cycles
// (/opt/skills/guides/MI355X_MICROARCH.md, "Per-instruction cycle constants") together with the
// clock that assumption implies, which must come out at or below 2.4 GHz.
// Output: one JSON object on stdout (tools/isa_histogram.py --costs reads it).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
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

constexpr int kIters = 2048;
constexpr int kPerIter = 32; // instructions per loop iteration (8 chains x 4)

// 32-bit operand classes
#define KERNEL32(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed) {                           \
        float a0 = seed, a1 = seed + 1.f, a2 = seed + 2.f, a3 = seed + 3.f, a4 = seed + 4.f,        \
              a5 = seed + 5.f, a6 = seed + 6.f, a7 = seed + 7.f;                                    \
        const float b = 1.0000001f, c = 1e-9f;                                                      \
        for (int i = 0; i < kIters; ++i) {                                                          \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
                asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
            }                                                                                       \
        }                                                                                           \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0;             \
    }

// 64-bit operand classes (f64 and packed f32: one VGPR pair per operand)
#define KERNEL64(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed) {                           \
        double a0 = seed, a1 = seed + 1., a2 = seed + 2., a3 = seed + 3., a4 = seed + 4.,           \
               a5 = seed + 5., a6 = seed + 6., a7 = seed + 7.;                                      \
        const double b = 1.0000001, c = 1e-9;                                                       \
        for (int i = 0; i < kIters; ++i) {                                                          \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
                asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
                asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");                                      \
            }                                                                                       \
        }                                                                                           \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[threadIdx.x] = (float)a0;       \
    }

// ---- f32 ----
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL32(k_add_f32, "v_add_f32 %0, %0, %2")
KERNEL32(k_max_f32, "v_max_f32 %0, %0, %2")
KERNEL32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL32(k_rsq_f32, "v_rsq_f32 %0, %0")
KERNEL32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL32(k_exp_f32, "v_exp_f32 %0, %0")
KERNEL32(k_log_f32, "v_log_f32 %0, %0")
KERNEL32(k_sin_f32, "v_sin_f32 %0, %0")
KERNEL32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL32(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL32(k_fract_f32, "v_fract_f32 %0, %0")
KERNEL32(k_floor_f32, "v_floor_f32 %0, %0")
KERNEL32(k_rndne_f32, "v_rndne_f32 %0, %0")
KERNEL32(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL32(k_lshl_b32, "v_lshlrev_b32 %0, 1, %0")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp_f32, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL32(k_cmpx_f32, "v_cmp_lt_f32 s[20:21], %0, %1")
KERNEL32(k_ldexp_f32, "v_ldexp_f32 %0, %0, 1")
KERNEL32(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
KERNEL32(k_readlane, "v_readlane_b32 s20, %0, 3")
KERNEL32(k_writelane, "v_writelane_b32 %0, s20, 3")
KERNEL32(k_readfirstlane, "v_readfirstlane_b32 s20, %0")
// ---- packed f32 (two f32 per lane) ----
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %2")
KERNEL64(k_pk_mov_b32, "v_pk_mov_b32 %0, %1, %2")
// ---- f64 ----
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %2")
KERNEL64(k_max_f64, "v_max_f64 %0, %0, %2")
KERNEL64(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL64(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL64(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %1")
KERNEL64(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
KERNEL64(k_fract_f64, "v_fract_f64 %0, %0")
KERNEL64(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL64(k_floor_f64, "v_floor_f64 %0, %0")
KERNEL64(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1")
KERNEL64(k_frexp_mant_f64, "v_frexp_mant_f64 %0, %0")
KERNEL64(k_div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %1, %0")
KERNEL64(k_div_fmas_f64, "v_div_fmas_f64 %0, %0, %1, %2")
KERNEL64(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
KERNEL64(k_trig_preop_f64, "v_trig_preop_f64 %0, %0, 1")
// mixed widths: %0 = 32-bit register, %1 = 64-bit pair (throughput only: no dependency chain needed)
#define KERNELMIX(NAME, ASM, OUT32)                                                                 \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed) {                           \
        float f0 = seed, f1 = seed, f2 = seed, f3 = seed, f4 = seed, f5 = seed, f6 = seed, f7 = seed; \
        double d0 = seed, d1 = seed, d2 = seed, d3 = seed, d4 = seed, d5 = seed, d6 = seed, d7 = seed; \
        for (int i = 0; i < kIters; ++i) {                                                          \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
                if (OUT32) {                                                                        \
                    asm volatile(ASM : "=v"(f0) : "v"(d0) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(f1) : "v"(d1) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(f2) : "v"(d2) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(f3) : "v"(d3) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(f4) : "v"(d4) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(f5) : "v"(d5) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(f6) : "v"(d6) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(f7) : "v"(d7) : "vcc", "s20", "s21");                                         \
                } else {                                                                            \
                    asm volatile(ASM : "=v"(d0) : "v"(f0) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(d1) : "v"(f1) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(d2) : "v"(f2) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(d3) : "v"(f3) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(d4) : "v"(f4) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(d5) : "v"(f5) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(d6) : "v"(f6) : "vcc", "s20", "s21");                                         \
                    asm volatile(ASM : "=v"(d7) : "v"(f7) : "vcc", "s20", "s21");                                         \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
        if (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) == 12345.678f) \
            out[threadIdx.x] = f0;                                                                  \
    }
KERNELMIX(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %1", true)
KERNELMIX(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1", false)
KERNELMIX(k_cvt_i32_f64, "v_cvt_i32_f64 %0, %1", true)
KERNELMIX(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1", false)

struct Case {
    const char *name;
    void (*fn)(float *, float);
};

int main(int argc, char **argv) {
    const int waves_per_simd = argc > 1 ? std::atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const int n_simd = n_cu * 4;
    float *out;
    CHECK(hipMalloc(&out, 4096));
    const std::vector<Case> cases = {
#define C(n) {#n, k_##n}
        C(fma_f32), C(mul_f32), C(add_f32), C(max_f32), C(rcp_f32), C(rsq_f32), C(sqrt_f32), C(exp_f32), C(log_f32),
        C(sin_f32), C(cvt_i32_f32), C(cvt_f32_i32), C(fract_f32), C(floor_f32), C(rndne_f32), C(mov_b32), C(add_u32),
        C(and_b32), C(lshl_b32), C(mul_lo_u32), C(cndmask_b32), C(cmp_f32), C(cmpx_f32), C(ldexp_f32), C(med3_f32),
        C(readlane), C(writelane), C(readfirstlane), C(pk_fma_f32), C(pk_mul_f32), C(pk_add_f32), C(pk_mov_b32),
        C(fma_f64), C(mul_f64), C(add_f64), C(max_f64), C(rcp_f64), C(rsq_f64), C(sqrt_f64), C(mov_b64), C(cmp_f64),
        C(fract_f64), C(rndne_f64), C(floor_f64), C(ldexp_f64), C(frexp_mant_f64), C(div_scale_f64), C(div_fmas_f64),
        C(div_fixup_f64), C(trig_preop_f64), C(cvt_f32_f64), C(cvt_f64_f32), C(cvt_i32_f64), C(cvt_f64_i32),
#undef C
    };
    // one block = 4 waves = one wave per SIMD of a CU; W blocks per CU
    const dim3 grid(n_cu * waves_per_simd), block(256);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::vector<double> ns(cases.size());
    for (size_t k = 0; k < cases.size(); ++k) {
        for (int rep = 0; rep < 2; ++rep) { // first launch warms up
            CHECK(hipEventRecord(a));
            for (int l = 0; l < 4; ++l) hipLaunchKernelGGL(cases[k].fn, grid, block, 0, 0, out, 0.5f);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, a, b));
            // wave-instructions per SIMD in the 4 launches
            const double per_simd = 4.0 * (double)waves_per_simd * kIters * kPerIter;
            ns[k] = (double)ms * 1e6 / per_simd;
        }
    }
    const double ns_fma = ns[0];
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"waves_per_simd\": %d, "
                "\"reference\": \"v_fma_f32 == 2 cycles per wave64 instruction per SIMD-32\", "
                "\"implied_clock_ghz\": %.4f, \"ns_per_wave_instruction\": {",
                prop.name, n_cu, n_simd, waves_per_simd, 2.0 / ns_fma);
    for (size_t k = 0; k < cases.size(); ++k) std::printf("%s\"v_%s\": %.5f", k ? ", " : "", cases[k].name, ns[k]);
    std::printf("}, \"cycles\": {");
    for (size_t k = 0; k < cases.size(); ++k)
        std::printf("%s\"v_%s\": %.3f", k ? ", " : "", cases[k].name, 2.0 * ns[k] / ns_fma);
    std::printf("}}\n");
    (void)n_simd;
    return 0;
}
