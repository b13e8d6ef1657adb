// shader_common.hpp -- small f32 vector helpers shared by the shader-march kernels
// (GLSL ES 3.00 built-ins: normalize(v) = v / sqrt(dot(v, v)); smoothstep, clamp, sign).
#pragma once

#include "geodesic_kernels.hpp"

namespace {

// f32 transcendental functions of the shader-order (STRICT) kernels.  GLSL / WGSL leave their
// precision to the implementation; the STRICT unit (GRV_SPECIFIED_LIBM) evaluates them as the
// written-out f64 routines of strict_libm.hpp rounded once to f32, which the checker in oracle/
// does too, so shader-order images are a pure function of their inputs.  The FAST unit only
// reaches these from code it never instantiates for FAST arithmetic.
#if defined(GRV_SPECIFIED_LIBM)
__device__ __forceinline__ float sh_sinf(float x) { return strictm::sl_sinf(x); }
__device__ __forceinline__ float sh_cosf(float x) { return strictm::sl_cosf(x); }
__device__ __forceinline__ float sh_powf(float x, float y) { return strictm::sl_powf(x, y); }
__device__ __forceinline__ float sh_expf(float x) { return strictm::sl_expf(x); }
__device__ __forceinline__ float sh_logf(float x) { return strictm::sl_logf(x); }
__device__ __forceinline__ float sh_acosf(float x) { return strictm::sl_acosf(x); }
__device__ __forceinline__ float sh_atan2f(float y, float x) { return strictm::sl_atan2f(y, x); }
#else
__device__ __forceinline__ float sh_sinf(float x) { return sinf(x); }
__device__ __forceinline__ float sh_cosf(float x) { return cosf(x); }
__device__ __forceinline__ float sh_powf(float x, float y) { return powf(x, y); }
__device__ __forceinline__ float sh_expf(float x) { return expf(x); }
__device__ __forceinline__ float sh_logf(float x) { return logf(x); }
__device__ __forceinline__ float sh_acosf(float x) { return acosf(x); }
__device__ __forceinline__ float sh_atan2f(float y, float x) { return atan2f(y, x); }
#endif

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ float clampf_d(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ float dot_f3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross_f3(F3 a, F3 b) {
    return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ F3 scale_f3(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ F3 add_f3(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float length_f3(F3 a) { return sqrtf(dot_f3(a, a)); }
__device__ __forceinline__ F3 normalize_f3(F3 a) {
    const float l = length_f3(a);
    return F3{a.x / l, a.y / l, a.z / l};
}
__device__ __forceinline__ float smoothstep_d(float e0, float e1, float x) {
    const float t = clampf_d((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
__device__ __forceinline__ float sign_d(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// block-level sum of the per-thread step counts into one atomic
__device__ __forceinline__ void add_steps(unsigned long long *total, uint32_t steps, uint32_t key) {
    __shared__ unsigned long long s_w[16];
    unsigned long long v = steps;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (uint32_t w = 0; w < (blockDim.x + 63u) / 64u; ++w) t += s_w[w];
        if (t) atomicAdd(total + (key % kStepParts) * kStepPartStride, t); // FrameStatsDev::steps_part, one line per slot
    }
}

} // namespace
