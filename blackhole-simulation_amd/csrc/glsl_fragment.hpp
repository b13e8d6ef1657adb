// glsl_fragment.hpp -- the reference's WebGL fragment shader as a HIP kernel:
// src/shaders/blackhole/fragment.glsl.ts:40-334 (the whole main()) with chunks/metric.ts,
// chunks/disk.ts, chunks/noise.ts, chunks/background.ts, chunks/blackbody.ts [SURVEY a16, a17,
// 8f-3].  ShaderManager's #defines (manager.ts:61-82) are GlslParams::features bits.
//
// Two arithmetic contracts, one per translation unit:
//   STRICT (kernels_strict.hip, -ffp-contract=off): the shader's operation order, IEEE divide /
//          sqrt, the specified f32 sin / cos / pow / exp / log / acos (shader_common.hpp: sh_*f);
//   FAST   (kernels_fast.hip, -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt):
//          the same expressions with FMA contraction, v_rcp_f32 / v_sqrt_f32 based divide and
//          sqrt, and a polynomial sin / cos for the ZAMO twist.  f32 rounding differences only.
//
// Not reproducible in the reference and therefore fixed here: the two noise
// textures are Math.random() upstream -- here they are engine-owned seeded 256x256 byte planes
// (grv_set_glsl_noise), sampled with f32 weights.
// One thread per pixel, registers only; a wave is one 8x8 pixel block.
#pragma once

#include "shader_common.hpp"

namespace {

// ===========================================================================
// GLSL fragment march
// ===========================================================================
// hot-loop math of the two contracts: STRICT = the GLSL built-ins as written (IEEE divide and
// sqrt); FAST = v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 (1 ulp each, no denormal scaling)
template <int ARITH> __device__ __forceinline__ float length_t(F3 a) {
    if constexpr (ARITH == GRV_ARITH_FAST) return __builtin_amdgcn_sqrtf(dot_f3(a, a));
    else return length_f3(a);
}
template <int ARITH> __device__ __forceinline__ F3 normalize_t(F3 a) {
    if constexpr (ARITH == GRV_ARITH_FAST) return scale_f3(a, __builtin_amdgcn_rsqf(dot_f3(a, a)));
    else return normalize_f3(a);
}
// divide / sqrt of the sampling branches: FAST = x * v_rcp_f32(y), v_sqrt_f32, v_rsq_f32 (1 ulp each, no
// denormal rescue -- every operand here is a radius, a temperature or a density of moderate size); the
// compiler's own FAST-mode `/` and sqrtf() wrap the same instructions in frexp / ldexp scaling, 6-7
// instructions per quotient, and the disk block has twenty of them
#ifndef GRV_GLSL_RAW_DIV
#define GRV_GLSL_RAW_DIV 1
#endif
template <int ARITH> __device__ __forceinline__ float div_t(float a, float b) {
    if constexpr (ARITH == GRV_ARITH_FAST && GRV_GLSL_RAW_DIV) return a * __builtin_amdgcn_rcpf(b);
    else return a / b;
}
template <int ARITH> __device__ __forceinline__ float sqrt_t(float a) {
    if constexpr (ARITH == GRV_ARITH_FAST && GRV_GLSL_RAW_DIV) return __builtin_amdgcn_sqrtf(a);
    else return sqrtf(a);
}
template <int ARITH> __device__ __forceinline__ float rsqrt_t(float a) { // 1 / sqrt(a)
    if constexpr (ARITH == GRV_ARITH_FAST && GRV_GLSL_RAW_DIV) return __builtin_amdgcn_rsqf(a);
    else return 1.0f / sqrtf(a);
}
template <int ARITH> __device__ __forceinline__ float smoothstep_t(float e0, float e1, float x) {
    if constexpr (ARITH == GRV_ARITH_FAST) {
        // (1 / (e1 - e0) folds at compile time for the literal edges of the march; v_rcp_f32 otherwise)
        const float t = clampf_d((x - e0) * (1.0f / (e1 - e0)), 0.0f, 1.0f);
        return t * t * (3.0f - 2.0f * t);
    } else {
        return smoothstep_d(e0, e1, x);
    }
}

// chunks/metric.ts:96-149 in the FAST contract: the same expressions with the reciprocals and
// roots taken once (rsq of |p|^2 and of r_k^2, rcp of Sigma and of r_k^3 + a^2 r_k), split into the
// part that depends on the position alone and the part that needs the direction.  The Verlet step
// evaluates the acceleration at the new position with the old direction and, one iteration later,
// at the same position with the new direction: the march forms the position part once per position
// and carries it across the iteration boundary (with |p| for the step-size logic), which removes
// five of the six quarter-rate instructions and about half of the second evaluation.
struct GlslGeomFast {
    float rho2;   // |p|^2
    float r2_inv; // 1 / r_k^2
    float k_pull; // M r_k^-2 (r_k^2 / Sigma)
    float rs;     // 1 / |p|
    float drag;   // 2 M a / (r_k^3 + a^2 r_k) = omega
};
__device__ __forceinline__ GlslGeomFast glsl_geom_fast(F3 p, float M, float a) {
    const float a2 = a * a;
    GlslGeomFast g;
    g.rho2 = dot_f3(p, p);
    const float diff = g.rho2 - a2;
    const float py2 = p.y * p.y;
    const float disc = fmaf(diff, diff, 4.0f * a2 * py2);
    // (disc = diff^2 + 4 a^2 y^2 and L2_eff below are sums of squares closed by an fma: never negative,
    // so the shader's max(0, .) on them is the identity and is not issued)
    const float r2 = 0.5f * (diff + __builtin_amdgcn_sqrtf(disc));
    const float r2c = fmaxf(1e-8f, r2);
    const float inv_rk = __builtin_amdgcn_rsqf(r2c); // 1 / r_k
    const float r_k = r2c * inv_rk;
    g.r2_inv = inv_rk * inv_rk;
    const float sigma = fmaf(a2, py2 * g.r2_inv, r2);
    const float sigma_ratio = r2 * __builtin_amdgcn_rcpf(fmaxf(1e-8f, sigma));
    g.k_pull = M * g.r2_inv * sigma_ratio;
    g.rs = __builtin_amdgcn_rsqf(g.rho2);
    g.drag = 2.0f * M * a * __builtin_amdgcn_rcpf(fmaxf(1e-8f, r_k * (r2 + a2)));
    return g;
}
__device__ __forceinline__ F3 glsl_accel_from_geom(const GlslGeomFast &g, F3 p, F3 v, float a) {
    const F3 L = cross_f3(p, v);
    const float Ly_eff = L.y - a;
    const float L2_eff = fmaf(Ly_eff, Ly_eff, fmaf(L.x, L.x, L.z * L.z));
    // M r^-2 S + 3 M max(0, L^2) r^-4 S, along -normalize(p)
    const float s = -(g.k_pull * fmaf(3.0f * L2_eff, g.r2_inv, 1.0f)) * g.rs;
    // cross((0,1,0), v) = (v.z, 0, -v.x)
    return F3{fmaf(p.x, s, v.z * g.drag), p.y * s, fmaf(p.z, s, -v.x * g.drag)};
}
__device__ __forceinline__ F3 glsl_kerr_accel_fast(F3 p, F3 v, float M, float a, float &omega) {
    const GlslGeomFast g = glsl_geom_fast(p, M, a);
    omega = g.drag;
    return glsl_accel_from_geom(g, p, v, a);
}
template <int ARITH>
__device__ __forceinline__ F3 glsl_accel(F3 p, F3 v, float M, float a, float &omega);

// chunks/metric.ts:96-149
__device__ __forceinline__ F3 glsl_kerr_accel(F3 p, F3 v, float M, float a, float &omega) {
    const float a2 = a * a;
    const float rho2 = dot_f3(p, p);
    const float diff = rho2 - a2;
    const float disc = diff * diff + 4.0f * a2 * p.y * p.y;
    const float r2 = 0.5f * (diff + sqrtf(fmaxf(0.0f, disc)));
    const float r_k = sqrtf(fmaxf(1e-8f, r2));
    const float sigma = r2 + a2 * (p.y * p.y / fmaxf(1e-8f, r2));
    const F3 L = cross_f3(p, v);
    const float Ly = L.y;
    const float Ly_eff = Ly - a;
    const float L2_eff = Ly_eff * Ly_eff + (dot_f3(L, L) - Ly * Ly);
    const float r_inv = 1.0f / r_k;
    const float r2_inv = r_inv * r_inv;
    const float r4_inv = r2_inv * r2_inv;
    const float sigma_ratio = r2 / fmaxf(1e-8f, sigma);
    const F3 n = normalize_f3(p);
    const F3 r_hat{-n.x, -n.y, -n.z};
    F3 acc = scale_f3(r_hat, M * r2_inv * sigma_ratio + 3.0f * M * fmaxf(0.0f, L2_eff) * r4_inv * sigma_ratio);
    const float r3_p_a2r = r_k * r2 + a2 * r_k;
    const float drag = 2.0f * M * a / fmaxf(1e-8f, r3_p_a2r);
    acc = add_f3(acc, scale_f3(cross_f3(F3{0.0f, 1.0f, 0.0f}, v), drag));
    omega = 2.0f * M * a / fmaxf(1e-8f, r3_p_a2r);
    return acc;
}
template <int ARITH>
__device__ __forceinline__ F3 glsl_accel(F3 p, F3 v, float M, float a, float &omega) {
    if constexpr (ARITH == GRV_ARITH_FAST) return glsl_kerr_accel_fast(p, v, M, a, omega);
    else return glsl_kerr_accel(p, v, M, a, omega);
}

// sin / cos of the FAST contract: two-term Cody-Waite reduction by pi/2 + cephes minimax
// polynomials on [-pi/4, pi/4] (~1 ulp f32 for the O(1) angles of the march)
__device__ __forceinline__ void glsl_fast_sincos(float ang, float &s, float &c) {
    if (__ballot(!(fabsf(ang) < 2.44140625e-4f)) == 0ull) {
        // |angle| < 2^-12 on the whole wave (the twist of a step beyond r ~ 30 M): 1 - z/2 rounds to 1.0f
        // (z/2 < 2^-25: at most the tie, which goes to the even 1.0f) and the cubic term of the sine stays
        // under a quarter ulp -- the polynomials below return exactly (angle, 1.0f)
        s = ang;
        c = 1.0f;
        return;
    }
    if (__ballot(!(fabsf(ang) <= 0.0625f)) == 0ull) {
        // the whole wave twists by less than 1/16 rad (every march step beyond r ~ 3 M does): the next
        // terms of both series, x^7 / 5040 and x^6 / 720, stay below 2^-36 and 2^-33 -- far under half an
        // ulp of the results -- so two fma fewer give the same roundings as the full polynomials
        const float z = ang * ang;
        s = fmaf(ang * z, fmaf(z, 8.3333333333e-3f, -1.6666666667e-1f), ang);
        c = fmaf(z * z, 4.1666666667e-2f, fmaf(z, -0.5f, 1.0f));
        return;
    }
    if (__ballot(!(fabsf(ang) <= 0.78539816f)) == 0ull) {
        // the whole wave is inside [-pi/4, pi/4] (the ZAMO twist omega * dt of a march step always is):
        // j = 0, the reduction returns the argument and the quadrant logic is the identity -- same bits
        const float z = ang * ang;
        float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
        ps = fmaf(z, ps, -1.6666654611e-1f);
        s = fmaf(ang * z, ps, ang);
        float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
        pc = fmaf(z, pc, 4.166664568298827e-2f);
        c = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
        return;
    }
    const float j = rintf(ang * 0.636619772367581343f);
    float x = fmaf(-j, 1.57079637050628662109375f, ang);
    x = fmaf(-j, -4.37113900018624283e-8f, x);
    const float z = x * x;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    const float sr = fmaf(x * z, ps, x);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    const float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    const int q = (int)j;
    const float s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// pow / exp / log: the specified f32 forms (sh_*f) in the STRICT contract; v_log_f32 / v_exp_f32 (base 2, ~1 ulp each)
// in the FAST contract.  pow(0, y > 0) = exp2(-inf) = 0 in both.
template <int ARITH> __device__ __forceinline__ float pow_d(float x, float y) {
    if constexpr (ARITH == GRV_ARITH_FAST) return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
    else return sh_powf(x, y);
}
template <int ARITH> __device__ __forceinline__ float exp_d(float x) {
    if constexpr (ARITH == GRV_ARITH_FAST) return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
    else return sh_expf(x);
}
template <int ARITH> __device__ __forceinline__ float log_d(float x) {
    if constexpr (ARITH == GRV_ARITH_FAST) return __builtin_amdgcn_logf(x) * 0.693147180559945309f;
    else return sh_logf(x);
}

// chunks/common.ts:44-47 : `v.ab *= rot(ang)`
template <int ARITH>
__device__ __forceinline__ void glsl_rot(float ang, float &x, float &y) {
    float s, c;
    if constexpr (ARITH == GRV_ARITH_FAST) {
        glsl_fast_sincos(ang, s, c);
    } else {
        s = sh_sinf(ang);
        c = sh_cosf(ang);
    }
    const float nx = x * c + y * (-s);
    const float ny = x * s + y * c;
    x = nx;
    y = ny;
}

// chunks/blackbody.ts:9-34
template <int ARITH>
__device__ __forceinline__ void glsl_blackbody(float temp, float rgb[3]) {
    const float t = (ARITH == GRV_ARITH_FAST && GRV_GLSL_RAW_DIV) ? fmaxf(temp, 1.0f) * 0.01f : fmaxf(temp, 1.0f) / 100.0f;
    float r, g, b;
    if (t <= 66.0f) {
        r = 255.0f;
        g = 99.4708025861f * log_d<ARITH>(t) - 161.1195681661f;
        b = (t <= 19.0f) ? 0.0f : 138.5177312231f * log_d<ARITH>(t - 10.0f) - 305.0447927307f;
    } else {
        r = 329.698727446f * pow_d<ARITH>(t - 60.0f, -0.1332047592f);
        g = 288.1221695283f * pow_d<ARITH>(t - 60.0f, -0.0755148492f);
        b = 255.0f;
    }
    constexpr float k255 = (ARITH == GRV_ARITH_FAST && GRV_GLSL_RAW_DIV) ? 1.0f / 255.0f : 0.0f;
    if constexpr (ARITH == GRV_ARITH_FAST && GRV_GLSL_RAW_DIV) {
        rgb[0] = pow_d<ARITH>(fmaxf(r * k255, 0.0f), 2.2f);
        rgb[1] = pow_d<ARITH>(fmaxf(g * k255, 0.0f), 2.2f);
        rgb[2] = pow_d<ARITH>(fmaxf(b * k255, 0.0f), 2.2f);
    } else {
        rgb[0] = pow_d<ARITH>(fmaxf(r / 255.0f, 0.0f), 2.2f);
        rgb[1] = pow_d<ARITH>(fmaxf(g / 255.0f, 0.0f), 2.2f);
        rgb[2] = pow_d<ARITH>(fmaxf(b / 255.0f, 0.0f), 2.2f);
    }
}

// ---- chunks/noise.ts: the 256x256 R channels live in HBM/L2 (64 KiB each) ----
__device__ __forceinline__ float glsl_texel(const uint8_t *__restrict__ t, int x, int y) {
    return (float)t[(uint32_t)(y & 255) * 256u + (uint32_t)(x & 255)] / 255.0f; // REPEAT, UNORM8
}
// texture(u_noiseTex, (uv + 0.5) / 256.0).r with LINEAR + REPEAT, f32 weights
__device__ __forceinline__ float glsl_hash_uv(const uint8_t *__restrict__ T, float uvx, float uvy) {
    const float s = (uvx + 0.5f) / 256.0f, t = (uvy + 0.5f) / 256.0f;
    const float u = s * 256.0f - 0.5f, v = t * 256.0f - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    // REPEAT wrap of the texel index: (int)fu & 255 equals (int)fmod(fu, 256) & 255 whenever fu fits
    // an int32 (two's complement), which every lattice coordinate of the shader does.
    // beyond it: fu - 256 floor(fu / 256) is exact in f32 (power-of-two scalings, exact floor and
    // difference) and congruent to fmod(fu, 256) mod 256, so the masked index is the same -- without
    // fmodf's division loop inlined at every noise() corner
    const bool small = fabsf(fu) < 1.0e9f && fabsf(fv) < 1.0e9f;
    const int i0 = small ? (int)fu : (int)(fu - 256.0f * floorf(fu * 0.00390625f));
    const int j0 = small ? (int)fv : (int)(fv - 256.0f * floorf(fv * 0.00390625f));
    // integer lattice points (every noise() corner) land on a texel centre: weights (1, 0, 0, 0),
    // and 0 * texel is exactly 0 for UNORM8 data -- one fetch gives the bitwise same value
    if (a == 0.0f && b == 0.0f) return glsl_texel(T, i0, j0);
    const float t00 = glsl_texel(T, i0, j0), t10 = glsl_texel(T, i0 + 1, j0);
    const float t01 = glsl_texel(T, i0, j0 + 1), t11 = glsl_texel(T, i0 + 1, j0 + 1);
    return (1.0f - a) * (1.0f - b) * t00 + a * (1.0f - b) * t10 + (1.0f - a) * b * t01 + a * b * t11;
}
__device__ __forceinline__ float glsl_hash(const uint8_t *__restrict__ T, F3 p) { // noise.ts:3-9
    return glsl_hash_uv(T, p.x + p.z * 37.0f, p.y + p.z * 37.0f);
}
#ifndef GRV_GLSL_NOISE_INLINE
#define GRV_GLSL_NOISE_INLINE
#endif
__device__ __forceinline__ float mix_d(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ float fract_d(float x) { return x - floorf(x); }
__device__ GRV_GLSL_NOISE_INLINE float glsl_noise(const uint8_t *__restrict__ T, F3 p) { // noise.ts:11-21
    const F3 i{floorf(p.x), floorf(p.y), floorf(p.z)};
    F3 f{fract_d(p.x), fract_d(p.y), fract_d(p.z)};
    f.x = f.x * f.x * (3.0f - 2.0f * f.x);
    f.y = f.y * f.y * (3.0f - 2.0f * f.y);
    f.z = f.z * f.z * (3.0f - 2.0f * f.z);
    auto H = [&](float dx, float dy, float dz) { return glsl_hash(T, F3{i.x + dx, i.y + dy, i.z + dz}); };
    return mix_d(mix_d(mix_d(H(0, 0, 0), H(1, 0, 0), f.x), mix_d(H(0, 1, 0), H(1, 1, 0), f.x), f.y),
                 mix_d(mix_d(H(0, 0, 1), H(1, 0, 1), f.x), mix_d(H(0, 1, 1), H(1, 1, 1), f.x), f.y), f.z);
}
// FAST contract: noise() on the integer lattice.  Every corner noise() hashes is an integer point, so
// hash()'s texture coordinate (p.xy + p.z * 37 + 0.5) / 256 is a texel centre: LINEAR filtering
// returns that one texel (weights 1, 0, 0, 0 -- glsl_hash_uv's a == b == 0 case), and with REPEAT
// wrapping the texel index is ((x + 37 z) mod 256, (y + 37 z) mod 256) of the corner.  That is
// integer arithmetic on the cell's base corner reduced mod 256 (i - 256 floor(i / 256): exact for every
// float): one add / mask per corner instead of the float round trip through uv space (~30
// instructions per corner), and the texel comes from an f32 copy of the plane (byte / 255.0f, the IEEE
// quotient, formed once on the host) instead of load + convert + divide.  For |coordinates| < 2^17
// every f32 operation of the shader's hash() is exact too, so the value equals the shader-order
// noise() bit for bit; beyond that the shader's own sums round (p.z * 37 leaves the 24-bit mantissa)
// and this form keeps the exact index.
#ifndef GRV_GLSL_LATTICE_NOISE
#define GRV_GLSL_LATTICE_NOISE 1
#endif
#ifndef GRV_GLSL_JET_PREFILTER
#define GRV_GLSL_JET_PREFILTER 1
#endif
__device__ __forceinline__ float glsl_noise_lattice(const GlslParams &U, F3 p) {
    const F3 i{floorf(p.x), floorf(p.y), floorf(p.z)};
    F3 f{p.x - i.x, p.y - i.y, p.z - i.z};
    f.x = f.x * f.x * (3.0f - 2.0f * f.x);
    f.y = f.y * f.y * (3.0f - 2.0f * f.y);
    f.z = f.z * f.z * (3.0f - 2.0f * f.z);
    const int rx = (int)fmaf(-256.0f, floorf(i.x * 0.00390625f), i.x);
    const int ry = (int)fmaf(-256.0f, floorf(i.y * 0.00390625f), i.y);
    const int rz = (int)fmaf(-256.0f, floorf(i.z * 0.00390625f), i.z);
    const int ux = rx + 37 * rz, uy = ry + 37 * rz;
    const float *__restrict__ Tf = U.noise_f;
    auto H = [&](int dx, int dy, int dz) {
        return Tf[(((uint32_t)(uy + dy + 37 * dz) & 255u) << 8) | ((uint32_t)(ux + dx + 37 * dz) & 255u)];
    };
    return mix_d(mix_d(mix_d(H(0, 0, 0), H(1, 0, 0), f.x), mix_d(H(0, 1, 0), H(1, 1, 0), f.x), f.y),
                 mix_d(mix_d(H(0, 0, 1), H(1, 0, 1), f.x), mix_d(H(0, 1, 1), H(1, 1, 1), f.x), f.y), f.z);
}
template <int ARITH> __device__ __forceinline__ float glsl_noise_t(const GlslParams &U, F3 p) {
    if constexpr (ARITH == GRV_ARITH_FAST && GRV_GLSL_LATTICE_NOISE) return glsl_noise_lattice(U, p);
    else return glsl_noise(U.noise_r, p);
}
template <int ARITH>
__device__ __forceinline__ float glsl_fbm(const GlslParams &U, F3 p) { // noise.ts:23-33
    float f = 0.0f, amp = 0.5f;
    for (int i = 0; i < 4; ++i) {
        f += amp * glsl_noise_t<ARITH>(U, p);
        p = scale_f3(p, 2.0f);
        amp *= 0.5f;
    }
    return f;
}

// chunks/blackbody.ts:36-46
__device__ __forceinline__ void glsl_star_color(float bv, float c[3]) {
    const float t = clampf_d(bv, -0.4f, 2.0f);
    if (t < 0.0f) { c[0] = 0.6f; c[1] = 0.7f; c[2] = 1.0f; }
    else if (t < 0.3f) { c[0] = 0.85f; c[1] = 0.88f; c[2] = 1.0f; }
    else if (t < 0.6f) { c[0] = 1.0f; c[1] = 0.96f; c[2] = 0.9f; }
    else if (t < 1.0f) { c[0] = 1.0f; c[1] = 0.85f; c[2] = 0.6f; }
    else { c[0] = 1.0f; c[1] = 0.6f; c[2] = 0.4f; }
}

// chunks/background.ts:3-30
template <int ARITH>
__device__ void glsl_starfield(const GlslParams &U, F3 dir, float stars[3]) {
    const uint8_t *T = U.noise_r;
    stars[0] = stars[1] = stars[2] = 0.0f;
    F3 cell{floorf(dir.x * 200.0f), floorf(dir.y * 200.0f), floorf(dir.z * 200.0f)};
    float starNoise = glsl_hash(T, cell);
    if (starNoise > 0.998f) {
        const float brightness = pow_d<ARITH>(starNoise, 10.0f) * 2.0f;
        const float bv = glsl_hash(T, F3{cell.x + 127.1f, cell.y + 127.1f, cell.z + 127.1f}) * 2.4f - 0.4f;
        const float tw_arg = U.time * (3.0f + glsl_hash(T, F3{cell.x + 73.7f, cell.y + 73.7f, cell.z + 73.7f}) * 2.0f);
        float tw_s, tw_c;
        if constexpr (ARITH == GRV_ARITH_FAST) {
            glsl_fast_sincos(tw_arg, tw_s, tw_c);
        } else {
            tw_s = sh_sinf(tw_arg);
        }
        const float twinkle = 0.85f + 0.15f * tw_s;
        float sc[3];
        glsl_star_color(bv, sc);
        for (int c = 0; c < 3; ++c) stars[c] = sc[c] * brightness * twinkle;
    }
    cell = F3{floorf(dir.x * 500.0f), floorf(dir.y * 500.0f), floorf(dir.z * 500.0f)};
    starNoise = glsl_hash(T, cell);
    if (starNoise > 0.996f) {
        const float brightness = pow_d<ARITH>(starNoise, 20.0f) * 1.5f;
        const float bv = glsl_hash(T, F3{cell.x + 217.3f, cell.y + 217.3f, cell.z + 217.3f}) * 2.4f - 0.4f;
        float sc[3];
        glsl_star_color(bv, sc);
        for (int c = 0; c < 3; ++c) stars[c] += sc[c] * brightness;
    }
    const float tt = U.time * 0.01f;
    const float nebula = glsl_fbm<ARITH>(U, F3{dir.x * 2.0f + tt, dir.y * 2.0f + tt, dir.z * 2.0f + tt}) * 0.03f;
    const float ln = fabsf(nebula);
    stars[0] += nebula * 0.2f + 0.05f * ln;
    stars[1] += nebula * 0.3f + 0.02f * ln;
    stars[2] += nebula * 0.5f + 0.05f * ln;
}

// chunks/disk.ts:16-115
// (r_p = |p|, which the march has already: the FAST contract takes the sample radius from it when
// the step did not cross the plane and the sample point is p itself)
template <int ARITH>
__device__ __forceinline__ void glsl_sample_disk(const GlslParams &U, F3 p, F3 p_prev, F3 v,
                                                 float isco, float M, float a, float dt,
                                                 float col[3], float &alpha, float r_p) {
    if (!(U.show_redshift < 0.5f)) return;
    const bool crossed = (p_prev.y * p.y < 0.0f);
    F3 sp = p;
    float sampleR = r_p;
    if (crossed) {
        const float t = div_t<ARITH>(fabsf(p_prev.y), fmaxf(0.0001f, fabsf(p_prev.y) + fabsf(p.y)));
        sp.x = p_prev.x * (1.0f - t) + p.x * t;
        sp.y = p_prev.y * (1.0f - t) + p.y * t;
        sp.z = p_prev.z * (1.0f - t) + p.z * t;
        if constexpr (ARITH == GRV_ARITH_FAST) sampleR = length_t<ARITH>(sp);
    }
    if constexpr (ARITH != GRV_ARITH_FAST) sampleR = length_t<ARITH>(sp);
    const float effH = fminf(U.disk_scale_height, 0.45f);
    const float diskHeight = sampleR * effH;
    const float diskInner = isco;
    const float diskOuter = fmaxf(M * U.disk_size, diskInner * 1.1f);
    if (!((fabsf(sp.y) < diskHeight || crossed) && sampleR > diskInner && sampleR < diskOuter)) return;
    float turbulence = U.turbulence;
    if (turbulence < 0.0f) { // disk.ts:43-55: Keplerian phase rotation of the noise field
        const float sqrt_Mp = sqrt_t<ARITH>(M);
        const float signSpinPhase = sign_d(U.spin + 1e-8f);
        const float OmegaPhase = div_t<ARITH>(signSpinPhase * sqrt_Mp, sampleR * sqrt_t<ARITH>(sampleR) + a * sqrt_Mp);
        const float rotAngle = OmegaPhase * U.time * 0.12f * 10.0f;
        float cs, sn;
        if constexpr (ARITH == GRV_ARITH_FAST) {
            glsl_fast_sincos(rotAngle, sn, cs);
        } else {
            cs = sh_cosf(rotAngle);
            sn = sh_sinf(rotAngle);
        }
        F3 np{sp.x * cs + sp.z * (-sn), sp.y, sp.x * sn + sp.z * cs};
        np = scale_f3(np, 0.75f);
        turbulence = glsl_noise_t<ARITH>(U, np) * 0.5f + glsl_noise_t<ARITH>(U, scale_f3(np, 2.5f)) * 0.25f;
    }
    const float heightFalloff = exp_d<ARITH>(div_t<ARITH>(-fabsf(sp.y), fmaxf(0.001f, (sampleR * effH) * 0.25f)));
    float radialFalloff;
    if constexpr (ARITH == GRV_ARITH_FAST && GRV_GLSL_RAW_DIV) {
        const float ts = clampf_d((sampleR - diskOuter) * __builtin_amdgcn_rcpf(diskInner - diskOuter), 0.0f, 1.0f);
        radialFalloff = ts * ts * (3.0f - 2.0f * ts);
    } else {
        radialFalloff = smoothstep_d(diskOuter, diskInner, sampleR);
    }
    const float baseDensity = turbulence * heightFalloff * radialFalloff;
    if (!(baseDensity > 0.001f)) return;

    const float r2 = sampleR * sampleR;
    const float sqrt_M = sqrt_t<ARITH>(M);
    const float signSpin = sign_d(U.spin + 1e-8f);
    const float Omega = div_t<ARITH>(signSpin * sqrt_M, sampleR * sqrt_t<ARITH>(sampleR) + a * sqrt_M);
    float g_tt, g_tphi, g_phiphi, isco_r;
    if constexpr (ARITH == GRV_ARITH_FAST && GRV_GLSL_RAW_DIV) { // the four quotients by sampleR share one v_rcp_f32
        const float inv_r = __builtin_amdgcn_rcpf(sampleR);
        g_tt = -(1.0f - 2.0f * M * inv_r);
        g_tphi = -2.0f * M * a * inv_r;
        g_phiphi = r2 + a * a + 2.0f * M * a * a * inv_r;
        isco_r = clampf_d(isco * inv_r, 0.0f, 1.0f);
    } else {
        g_tt = -(1.0f - 2.0f * M / sampleR);
        g_tphi = -2.0f * M * a / sampleR;
        g_phiphi = r2 + a * a + 2.0f * M * a * a / sampleR;
        isco_r = clampf_d(isco / sampleR, 0.0f, 1.0f);
    }
    const float u_t_sq = -(g_tt + 2.0f * Omega * g_tphi + Omega * Omega * g_phiphi);
    const float u_t = rsqrt_t<ARITH>(fmaxf(1e-6f, u_t_sq));
    const float L_photon = p.z * v.x - p.x * v.z;
    const float delta = div_t<ARITH>(1.0f, fmaxf(0.01f, u_t * (1.0f - Omega * L_photon)));
    const float beaming = (U.features & GRV_GLSL_DOPPLER) ? fmaxf(0.01f, pow_d<ARITH>(delta, 3.5f)) : 1.0f;
    const float nt_factor = fmaxf(0.0f, 1.0f - sqrt_t<ARITH>(isco_r));
    const float grad = pow_d<ARITH>(isco_r, 0.75f) * pow_d<ARITH>(nt_factor, 0.25f);
    const float temperature = U.disk_temp * grad * delta;
    float bb[3];
    glsl_blackbody<ARITH>(temperature, bb);
    const float density = baseDensity * U.disk_density * 0.12f * dt;
#pragma unroll
    for (int c = 0; c < 3; ++c) col[c] += bb[c] * beaming * density * (1.0f - alpha);
    alpha += density;
}

// chunks/disk.ts:117-155
template <int ARITH>
__device__ __forceinline__ void glsl_sample_jets(const GlslParams &U, F3 p, F3 v, float rh, float dt,
                                                 float col[3], float &alpha, float r_p = 0.0f) {
    const float jetVerticalPos = fabsf(p.y);
    if constexpr (ARITH == GRV_ARITH_FAST) {
        // a ray is inside the jets only within 2 (1 + 0.15 |y|) of the axis, and its axial distance
        // sqrt(|p|^2 - y^2) is at least |p| - |y|: beyond |p| = 2.01 + 1.31 |y| (the shader's bound with a
        // margin of 0.01 (1 + |y|), a thousand times the rounding of |p|) nothing below can pass, and one
        // fma and one compare replace the nine operations of the tests -- which stay as they are
        if (!(r_p < fmaf(jetVerticalPos, 1.31f, 2.01f))) return;
    }
    if (!(jetVerticalPos > rh * 1.8f && jetVerticalPos < 10000.0f * 0.8f)) return;
    const float jetWidth = 1.0f + jetVerticalPos * 0.15f;
    float radialFalloff;
    if constexpr (ARITH == GRV_ARITH_FAST) {
        // the shader only ever uses the axial distance squared: compared against the squared width and
        // fed to the falloff as it is, no root (most steps outside the slab reach this test and fail it)
        const float rad2 = fmaf(p.x, p.x, p.z * p.z), w2 = jetWidth * 2.0f;
        if (!(rad2 < w2 * w2)) return;
        const float q = div_t<ARITH>(-rad2, jetWidth * 0.5f);   // the radial falloff's exponent, as the shader forms it
        if constexpr (GRV_GLSL_JET_PREFILTER) {
            // The density below is exp(q) exp(-0.05 |y|) max(0, noise - 0.2) with noise <= 1 (both octaves are trilinear
            // mixes of texels in [0, 1]), cut at 0.001.  Where the two exponents sum to -6.9 or less the falloffs' product
            // is <= 0.00101 (v_exp_f32 is good to 1e-6 relative, this sum to 1e-6 absolute) and the density <= 0.00081:
            // the cut is taken whatever the noise says -- and neither the noise's sixteen texel loads nor the two
            // exponentials are issued.  (Rays between -6.9 and the exact boundary ln(0.00125) = -6.68 take the shader's
            // path.)  The shader's own cone, axial distance < 2 w, is far wider than what can pass (distance^2 < 3.4 w:
            // 85 % of the cone's section at w = 1, 21 % at |y| = 20): a view down the axis spent most of its steps in that
            // dead shell -- c2 at theta = 5 deg: 162 -> 448 G ray-steps/s at r0 = 60 M, 237 -> 478 G at 200 M, every pixel
            // and step count unchanged (profiles/r06_ab_glsl_jet_prefilter.jsonl, profiles/EXPERIMENTS.md W).
            if (!(fmaf(jetVerticalPos, -0.05f, q) > -6.9f)) return;
        }
        radialFalloff = exp_d<ARITH>(q);
    } else {
        const float jetRadialDist = sqrt_t<ARITH>(p.x * p.x + p.z * p.z);
        if (!(jetRadialDist < jetWidth * 2.0f)) return;
        radialFalloff = exp_d<ARITH>(div_t<ARITH>(-(jetRadialDist * jetRadialDist), jetWidth * 0.5f));
    }
    const float lengthFalloff = exp_d<ARITH>(-jetVerticalPos * 0.05f);
    const float flow = p.y * 2.0f - U.time * 8.0f;
    const F3 uvJet{p.x, flow, p.z};
    const float noiseVal = glsl_noise_t<ARITH>(U, scale_f3(uvJet, 0.5f)) * 0.6f +
                           glsl_noise_t<ARITH>(U, scale_f3(uvJet, 1.5f)) * 0.4f;
    const float jetDensity = radialFalloff * lengthFalloff * fmaxf(0.0f, noiseVal - 0.2f);
    if (!(jetDensity > 0.001f)) return;
    const float jetVel = 0.92f * sign_d(p.y);
    const F3 nv = normalize_f3(F3{0.0f, jetVel, 0.0f});
    const float cosThetaJet = dot_f3(nv, F3{-v.x, -v.y, -v.z});
    const float betaJet = fabsf(jetVel);
    const float gammaJet = rsqrt_t<ARITH>(1.0f - betaJet * betaJet);
    const float deltaJet = div_t<ARITH>(1.0f, gammaJet * (1.0f - betaJet * cosThetaJet));
    const float beamingJet = pow_d<ARITH>(deltaJet, 3.5f);
    const float base[3] = {0.4f, 0.7f, 1.0f};
#pragma unroll
    for (int c = 0; c < 3; ++c) col[c] += base[c] * jetDensity * 0.05f * beamingJet * dt * (1.0f - alpha);
    alpha += jetDensity * 0.05f * dt;
}

__device__ __forceinline__ float aces_d(float c) {
    return clampf_d((c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f), 0.0f, 1.0f);
}

__device__ __forceinline__ F3 glsl_qrot(const float q[4], F3 v) { // common.ts:74-76
    const F3 qv{q[0], q[1], q[2]};
    const F3 t = add_f3(cross_f3(qv, v), scale_f3(v, q[3]));
    return add_f3(v, scale_f3(cross_f3(qv, t), 2.0f));
}

// distance from (px, py) to the segment a-b (fragment.glsl.ts:303-308)
__device__ __forceinline__ float glsl_seg_dist(float px, float py, float ax, float ay, float bx, float by) {
    const float pax = px - ax, pay = py - ay, bax = bx - ax, bay = by - ay;
    const float h = clampf_d((pax * bax + pay * bay) / (bax * bax + bay * bay), 0.0f, 1.0f);
    const float dx = pax - bax * h, dy = pay - bay * h;
    return sqrtf(dx * dx + dy * dy);
}

// fragment.glsl.ts:40-334 -- the whole main().  Returns the march steps taken.
template <int ARITH>
__device__ uint32_t glsl_fragment(const GlslParams &U, uint32_t width, uint32_t height, uint32_t X,
                                  uint32_t Y, float o[3]) {
    const uint32_t F = U.features;
    const float PI = 3.14159265359f;
    const float resx = (float)width, resy = (float)height;
    const float minRes = fminf(resx, resy);
    const float fcx = (float)X + 0.5f, fcy = (float)(height - 1u - Y) + 0.5f;
    const float uvx = (fcx - 0.5f * resx) / minRes, uvy = (fcy - 0.5f * resy) / minRes;
    if (U.debug > 0.5f) {
        o[0] = uvx + 0.5f;
        o[1] = uvy + 0.5f;
        o[2] = 0.0f;
        return 0;
    }
    F3 ro, rd;
    if (length_f3(F3{U.cam_pos[0], U.cam_pos[1], U.cam_pos[2]}) > 0.001f) {
        ro = F3{U.cam_pos[0], U.cam_pos[1], U.cam_pos[2]};
        rd = glsl_qrot(U.cam_quat, normalize_f3(F3{uvx, uvy, 1.2f}));
    } else {
        ro = F3{0.0f, 0.0f, -U.zoom};
        rd = normalize_f3(F3{uvx, uvy, 1.5f});
        const float ax = (U.mouse[1] - 0.5f) * PI, ay = (U.mouse[0] - 0.5f) * PI * 2.0f;
        glsl_rot<ARITH>(ax, ro.y, ro.z);
        glsl_rot<ARITH>(ax, rd.y, rd.z);
        glsl_rot<ARITH>(ay, ro.x, ro.z);
        glsl_rot<ARITH>(ay, rd.x, rd.z);
    }

    const float M = U.mass;
    const float rs = M * 2.0f;
    const float a = U.spin * M;
    const float rh = M + sqrtf(fmaxf(0.0f, M * M - a * a)); // metric.ts:13-15
    // metric.ts:32-37
    const float a_star = clampf_d(a / M, -0.9999f, 0.9999f);
    const float rph = 2.0f * M * (1.0f + sh_cosf((2.0f / 3.0f) * sh_acosf(clampf_d(-a_star, -1.0f, 1.0f))));
    // metric.ts:18-29
    const float absS = fabsf(clampf_d(a / M, -0.9999f, 0.9999f));
    const float z1 = 1.0f + pow_d<ARITH>(1.0f - absS * absS, 1.0f / 3.0f) *
                                (pow_d<ARITH>(1.0f + absS, 1.0f / 3.0f) + pow_d<ARITH>(1.0f - absS, 1.0f / 3.0f));
    const float z2 = sqrtf(3.0f * absS * absS + z1 * z1);
    float sgnA = sign_d(a);
    if (sgnA == 0.0f) sgnA = 1.0f;
    const float isco = M * (3.0f + z2 - sgnA * sqrtf((3.0f - z1) * (3.0f + z1 + 2.0f * z2)));
    const float absA = fabsf(U.spin);

    if (U.quality == 0) { // RAY_QUALITY_LOW / OFF indicator path, fragment.glsl.ts:76-88
        float bg[3];
        glsl_starfield<ARITH>(U, rd, bg);
        const float d = length_f3(cross_f3(ro, rd));
        const float shadow = smoothstep_d(rh * 1.2f, rh * 0.9f, d);
        const float glow = exp_d<ARITH>(-fabsf(d - rph) * 12.0f) * 0.8f;
        const float glowCol[3] = {0.3f * glow, 0.6f * glow, 1.0f * glow};
        const float diskMask =
            smoothstep_d(isco * 2.0f, isco * 1.0f, d) * (1.0f - smoothstep_d(isco * 1.0f, isco * 0.8f, d));
        const float diskCol[3] = {1.0f * diskMask * 0.6f, 0.7f * diskMask * 0.6f, 0.3f * diskMask * 0.6f};
        for (int c = 0; c < 3; ++c) o[c] = pow_d<ARITH>(bg[c] * (1.0f - shadow) + glowCol[c] + diskCol[c], 0.4545f);
        return 0;
    }

    F3 p = ro, v = rd;
    if (length_f3(ro) < rh * 1.5f) {
        ro = scale_f3(scale_f3(normalize_f3(ro), rh), 1.5f);
        p = ro;
    }
    float col[3] = {0.0f, 0.0f, 0.0f};
    float alpha = 0.0f;
    bool hitHorizon = false;
    float maxRedshift = 0.0f;
    float bNoise = 0.0f; // fragment.glsl.ts:104-108, NEAREST + REPEAT
    if (F & GRV_GLSL_DITHER) bNoise = glsl_texel(U.blue_r, (int)floorf(fcx), (int)floorf(fcy));
    p = add_f3(p, scale_f3(scale_f3(v, bNoise), 0.01f));

    int photonCrossings = 0;
    float prevY = p.y;
    const float impactParam = length_f3(cross_f3(ro, rd));
    bool redshiftInit = false;
    const int maxSteps = (int)fminf((float)U.max_ray_steps, 500.0f);
    F3 p_prev = p;
    if (impactParam < rh * 0.9f) hitHorizon = true;
    uint32_t steps = 0;
    const bool lensing = (F & GRV_GLSL_LENSING) != 0u, disk = (F & GRV_GLSL_DISK) != 0u;
    const bool jets = disk && (F & GRV_GLSL_JETS) != 0u;

    // FAST: position part of the acceleration and |p| of the current position, carried from the
    // previous iteration's second evaluation (see GlslGeomFast)
    GlslGeomFast geom{};
    float r_cur = 0.0f;
    if constexpr (ARITH == GRV_ARITH_FAST) {
        if (lensing) {
            geom = glsl_geom_fast(p, M, a);
            r_cur = geom.rho2 * geom.rs; // |p| from the reciprocal root the pull needs anyway
        } else {
            r_cur = length_t<ARITH>(p);
        }
    }

    // FAST: the same march with ONE loop exit.  The shader's loop leaves from three places (horizon and
    // far tests at the top, the opaque disk in the middle); compiled as written, every exit keeps its own
    // copies of the loop-carried state alive and the loop head spends ~25 moves and ~30 exec-mask
    // operations per iteration on merging them.  Here the three conditions are evaluated where the shader
    // evaluates them and tested together at the top of the next iteration; nothing else runs in between
    // (the jets of an iteration whose disk sample went opaque are skipped, as the shader's break skips them).
    if constexpr (ARITH == GRV_ARITH_FAST) {
        int i = 0;
        bool opaque = false, hz = false;
        const float slabH = fminf(U.disk_scale_height, 0.45f); // sample_disk's effH
        const float r_far = fmaxf(64.0f, rph + 21.0f);
        // Two Verlet steps per loop pass on two position registers sets (old -> new, new -> old): the loop keeps
        // ONE exit (the second step is a plain divergent `if`; a ray that must leave before it skips it and
        // leaves at the next top test), and the three moves `p_prev = p` of every step are gone.  Same
        // operations step for step: pixels and step counts bit for bit (tools/ab_glsl_identical.py).
        F3 qa = p, qb = p;
        auto verlet = [&](const F3 &pc, F3 &pn) __attribute__((always_inline)) {
            const float r = r_cur;
            // Far field: when every ray of the wave is beyond r_far = max(64, r_ph + 21) and at least 0.2 off
            // the equatorial plane, the step size formed below is 3.0f for each of them, exactly -- the
            // un-clamped step 0.1 (r - r_h)(1 + 0.05 r) >= 0.013 r (1 + 0.05 r) > 3.5 and its upper clamp
            // 1.2 (1 + 0.05 r) > 5 there, the photon-sphere limit 0.01 + 0.15 |r - r_ph| > 3.16, all cut by the
            // min with 3.0f, and the plane refinement is the factor 1.0f -- so the seventeen operations that
            // would find that out are skipped (every f32 radius: tests/test_glsl_fast_identities.py).  Most
            // steps of a frame are such steps.
            float dt, cdt;
            if (__ballot(!(r > r_far && fabsf(pc.y) >= 0.2f)) == 0ull) {
                dt = cdt = 3.0f;
            } else {
                const float distFactor = 1.0f + r * 0.05f;
                // (v_med3_f32 is the clamp for lo <= hi: 1.2 distFactor >= 1.2)
                dt = __builtin_amdgcn_fmed3f((r - rh) * 0.1f * distFactor, 0.01f, 1.2f * distFactor);
                // The shader's far-field block (fragment.glsl.ts:152-156) IS `dt = min(dt, 3.0f)` on every
                // radius the march can hold (1.15 r_h <= r: the horizon test above leaves first).  r > 30: the
                // un-clamped step exceeds the far boost 0.01 + 0.08 (r - 30) by >= 0.67 there (a quadratic in r
                // without real roots), so max(dt, boost) = dt unless dt sits at its upper clamp 1.2f distFactor
                // -- and distFactor = fma(r, 0.05f, 1) >= 2.5f makes that >= 3.0f (1.2f x 2.5f is the exact tie
                // between 3.0f and its successor: round-to-even gives 3.0f), where the shader's
                // min(., 1.2 * 2.5) returns 3.0f either way.  r <= 30: distFactor <= 2.5f, dt <= 3.0f, a min
                // changes nothing -- as the shader, which skips the block.  The 3.0f joins the photon-sphere
                // limit, min(min(dt, 3), lim) = min(dt, min(lim, 3)): one v_min3_f32 for a compare, four
                // operations, a select and a min.
                const float sphereProx = fabsf(r - rph);
                const float lim = fminf(0.01f + sphereProx * 0.15f, 3.0f);
                dt = fminf(dt, lim);
                // the refinement near the equatorial plane: smoothstep(0.2, 0, |y|) is exactly 0 for |y| >= 0.2
                // and the step then stays dt (dt * (1 - 0 * 0.7) = dt); skipped when no ray of the wave is
                // that close
                cdt = dt;
                if (__ballot(fabsf(pc.y) < 0.2f) != 0ull) {
                    const float hRefinement = smoothstep_t<ARITH>(0.2f, 0.0f, fabsf(pc.y));
                    cdt = dt * (1.0f - hRefinement * 0.7f);
                }
            }
            F3 accel{0.0f, 0.0f, 0.0f};
            if (lensing) {
                accel = glsl_accel_from_geom(geom, pc, v, a);
                glsl_rot<ARITH>(geom.drag * cdt, v.x, v.z);
            }
            const float k2 = 0.5f * cdt * cdt * U.lensing_strength;
            pn = F3{fmaf(accel.x, k2, fmaf(v.x, cdt, pc.x)), fmaf(accel.y, k2, fmaf(v.y, cdt, pc.y)),
                    fmaf(accel.z, k2, fmaf(v.z, cdt, pc.z))};
            float r_new;
            if (lensing) {
                geom = glsl_geom_fast(pn, M, a);
                r_new = geom.rho2 * geom.rs;
            } else {
                r_new = length_t<ARITH>(pn);
            }
            r_cur = r_new;
            if (lensing && alpha < 0.95f) {
                const F3 accel_new = glsl_accel_from_geom(geom, pn, v, a);
                const float kv = 0.5f * cdt * U.lensing_strength;
                v = F3{fmaf(accel.x + accel_new.x, kv, v.x), fmaf(accel.y + accel_new.y, kv, v.y),
                       fmaf(accel.z + accel_new.z, kv, v.z)};
            }
            v = normalize_t<ARITH>(v);
            ++i;
            if (U.show_redshift > 0.5f) {
                const float potential = sqrtf(fmaxf(0.0f, 1.0f - rs / r_new));
                maxRedshift = redshiftInit ? fminf(maxRedshift, potential) : potential;
                redshiftInit = true;
            }
            // the plane-crossing counter and the disk sample only matter on the few steps that cross the
            // equatorial plane or land inside the slab |y| < r effH (sample_disk returns at once otherwise:
            // without a crossing its sample point is pn and its radius r_new): two products and two compares
            // decide that, and everything else sits behind ONE branch (the empty asm keeps the optimiser
            // from turning the block back into selects issued on every step)
            const bool crossed = pc.y * pn.y < 0.0f;
            const bool in_slab = fabsf(pn.y) < r_new * slabH;
            if (crossed || (disk && in_slab)) {
                asm volatile("" ::: "memory");
                if (crossed && r_new < rph * 2.0f && r_new > rh)
                    photonCrossings = photonCrossings + 1 < 3 ? photonCrossings + 1 : 3;
                if (disk) glsl_sample_disk<ARITH>(U, pn, pc, v, isco, M, a, cdt, col, alpha, r_new);
            }
            if (disk) opaque = alpha > 0.99f;
            if (jets && !opaque) glsl_sample_jets<ARITH>(U, pn, v, rh, dt, col, alpha, r_new); // un-refined dt (fragment.glsl.ts:219)
        };
        for (;;) {
            hz = r_cur < rh * 1.15f;
            if (!(i < maxSteps) || opaque || hz || r_cur > 10000.0f) break;
            verlet(qa, qb);
            const bool hz2 = r_cur < rh * 1.15f;
            if (i < maxSteps && !opaque && !hz2 && !(r_cur > 10000.0f)) verlet(qb, qa);
        }
        p = (i & 1) ? qb : qa;
        steps = (uint32_t)i;
        hitHorizon = hitHorizon || (hz && !opaque && i < maxSteps);
        (void)prevY;
    } else {
        // shader order: the operations of fragment.glsl.ts:129-221 in their order, in the same one-exit
        // loop form (control flow only: every arithmetic result is the shader's, bit for bit)
        int i = 0;
        bool opaque = false, hz = false;
        for (;;) {
            const float r = length_t<ARITH>(p);
            hz = r < rh * 1.15f;
            if (!(i < maxSteps) || opaque || hz || r > 10000.0f) break;
            p_prev = p;
            const float distFactor = 1.0f + r * 0.05f;
            float dt = clampf_d((r - rh) * 0.1f * distFactor, 0.01f, 1.2f * distFactor);
            if (r > 30.0f) {
                const float farBoost = (r - 30.0f) * 0.08f;
                dt = fmaxf(dt, 0.01f + farBoost);
                dt = fminf(dt, 1.2f * 2.5f);
            }
            const float sphereProx = fabsf(r - rph);
            dt = fminf(dt, 0.01f + sphereProx * 0.15f);
            const float hRefinement = smoothstep_t<ARITH>(0.2f, 0.0f, fabsf(p.y));
            const float cdt = dt * (1.0f - hRefinement * 0.7f);

            F3 accel{0.0f, 0.0f, 0.0f};
            if (lensing) {
                float omega;
                accel = scale_f3(glsl_accel<ARITH>(p, v, M, a, omega), U.lensing_strength);
                glsl_rot<ARITH>(omega * cdt, v.x, v.z);
            }
            p = add_f3(p, add_f3(scale_f3(v, cdt), scale_f3(scale_f3(scale_f3(accel, 0.5f), cdt), cdt)));
            const float r_new = length_t<ARITH>(p);
            if (lensing && alpha < 0.95f) {
                float om2;
                const F3 accel_new = scale_f3(glsl_accel<ARITH>(p, v, M, a, om2), U.lensing_strength);
                v = add_f3(v, scale_f3(scale_f3(add_f3(accel, accel_new), 0.5f), cdt));
            }
            v = normalize_t<ARITH>(v);
            ++i;
            if (p_prev.y * p.y < 0.0f && r_new < rph * 2.0f && r_new > rh)
                photonCrossings = photonCrossings + 1 < 3 ? photonCrossings + 1 : 3;
            if (U.show_redshift > 0.5f) {
                const float potential = sqrtf(fmaxf(0.0f, 1.0f - rs / r_new));
                maxRedshift = redshiftInit ? fminf(maxRedshift, potential) : potential;
                redshiftInit = true;
            }
            if (disk) {
                glsl_sample_disk<ARITH>(U, p, p_prev, v, isco, M, a, cdt, col, alpha, r_new);
                opaque = alpha > 0.99f;
            }
            if (jets && !opaque) glsl_sample_jets<ARITH>(U, p, v, rh, dt, col, alpha); // un-refined dt (fragment.glsl.ts:219)
        }
        steps = (uint32_t)i;
        hitHorizon = hitHorizon || (hz && !opaque && i < maxSteps);
        (void)prevY;
    }

    if ((F & GRV_GLSL_REDSHIFT) && U.show_redshift > 0.5f) { // fragment.glsl.ts:224-237
        const float val = hitHorizon ? 0.0f : maxRedshift;
        const float c1[3] = {1.0f, 0.0f, 0.0f}, c2[3] = {1.0f, 1.0f, 0.0f}, c3[3] = {0.0f, 0.0f, 1.0f};
        const float t1 = smoothstep_d(0.0f, 0.3f, val), t2 = smoothstep_d(0.3f, 0.7f, val),
                    t3 = smoothstep_d(0.7f, 1.0f, val);
        for (int c = 0; c < 3; ++c) {
            float h = mix_d(0.0f, c1[c], t1);
            h = mix_d(h, c2[c], t2);
            o[c] = mix_d(h, c3[c], t3);
        }
        return steps;
    }

    float background[3] = {0.0f, 0.0f, 0.0f};
    if (F & GRV_GLSL_STARS) glsl_starfield<ARITH>(U, v, background);

    float photonColor = 0.0f;
    if ((F & GRV_GLSL_PHOTON_GLOW) && !hitHorizon) { // fragment.glsl.ts:246-258
        const float dring = fabsf(length_f3(p) - rph);
        const float directRing = exp_d<ARITH>(-dring * 40.0f) * 1.8f * U.lensing_strength;
        float higher = 0.0f;
        if (photonCrossings > 0) {
            const float sharp = 60.0f + (float)photonCrossings * 30.0f;
            const float bright = exp_d<ARITH>(-(float)photonCrossings * 1.0f) * 1.2f;
            higher = exp_d<ARITH>(-dring * sharp) * bright * U.lensing_strength;
        }
        photonColor = 1.0f * (directRing + higher);
    }
    float ergo[3] = {0.0f, 0.0f, 0.0f};
    if (absA > 0.1f && !hitHorizon) { // fragment.glsl.ts:261-268
        const float rFinal = length_f3(p);
        const float cosTheta = p.y / fmaxf(rFinal, 0.001f);
        const float r_ergo = M + sqrtf(fmaxf(0.0f, M * M - a * a * cosTheta * cosTheta));
        const float g = exp_d<ARITH>(-fabsf(rFinal - r_ergo) * 20.0f) * 0.35f * absA;
        ergo[0] = 0.3f * g;
        ergo[1] = 0.35f * g;
        ergo[2] = 0.9f * g;
    }
    if (hitHorizon) background[0] = background[1] = background[2] = 0.0f;
    float fin[3];
    for (int c = 0; c < 3; ++c)
        fin[c] = background[c] * (1.0f - alpha) + col[c] + photonColor * (1.0f - alpha) + ergo[c] * (1.0f - alpha);

    if (U.show_kerr_shadow > 0.5f) { // fragment.glsl.ts:279-324
        const F3 cam_dir = normalize_f3(ro);
        const F3 sky_right = normalize_f3(cross_f3(F3{0.0f, 1.0f, 0.0f}, cam_dir));
        const F3 sky_up = cross_f3(cam_dir, sky_right);
        const F3 impact = scale_f3(cross_f3(cam_dir, rd), length_f3(ro));
        const float sa = -dot_f3(impact, sky_up), sb = dot_f3(impact, sky_right);
        float minDist = 1e10f;
        const int count = (int)U.shadow_count;
        for (int j = 0; j < 63; ++j) {
            if (j >= count - 1) break;
            minDist = fminf(minDist, glsl_seg_dist(sa, sb, U.shadow_curve[j][0], U.shadow_curve[j][1],
                                                   U.shadow_curve[j + 1][0], U.shadow_curve[j + 1][1]));
        }
        if (count > 2)
            minDist = fminf(minDist, glsl_seg_dist(sa, sb, U.shadow_curve[count - 1][0],
                                                   U.shadow_curve[count - 1][1], U.shadow_curve[0][0],
                                                   U.shadow_curve[0][1]));
        const float thickness = M * 0.045f;
        if (minDist < thickness) {
            const float edge = smoothstep_d(thickness, thickness * 0.5f, minDist);
            const float green[3] = {0.0f, 1.0f, 0.0f};
            for (int c = 0; c < 3; ++c) fin[c] = mix_d(fin[c], green[c], 1.0f * edge);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = U.tone_map ? pow_d<ARITH>(fmaxf(aces_d(fin[c]), 0.0f), 0.4545f) : fin[c];
    return steps;
}

// one thread per pixel of this rank's tile set; a wave is one 8x8 pixel block
// FAST form compiled for eight waves per SIMD (64 VGPRs; ~50 B of scratch, all inside the disk / jet
// sampling branches, none on the far-field step).  A gfx950 SIMD issues two plain 32-bit VALU operations
// of two different waves in one quad-cycle (tools/valu_microbench.hip), and how often it finds a partner
// grows with the waves it can pick from: 5 -> 6 -> 7 -> 8 waves measured +2.3 / +5.3 / +6.3 % on the
// 1080p default preset (profiles/r04_ab_glsl_waves.jsonl)
#ifndef GRV_GLSL_FAST_WAVES
#define GRV_GLSL_FAST_WAVES 8
#endif

// Measurement instrument, compiled only into an A/B library (-DGRV_MARCH_TIMELINE; tools/march_timeline.py):
// every wave of the FAST march records {start, end} on the constant-rate clock, its hardware id and its
// step count, so that the launch's ramp, steady state and tail can be drawn wave by wave.
#ifdef GRV_MARCH_TIMELINE
__device__ unsigned long long *g_march_timeline = nullptr; // [waves][4]
#endif
template <int ARITH>
__global__ __launch_bounds__(kMarchBlock) __attribute__((amdgpu_waves_per_eu(ARITH == GRV_ARITH_FAST ? GRV_GLSL_FAST_WAVES : 1)))
void glsl_fragment_kernel(FrameGeom G, GlslParams U,
                                                               float4 *__restrict__ out_rgba,
                                                               uint32_t *__restrict__ out_steps,
                                                               unsigned long long *total_steps,
                                                               uint32_t n_slots, MarchSched sched) {
#ifdef GRV_MARCH_TIMELINE
    const unsigned long long tl0 = wall_clock64();
#endif
    // measured-cost dispatch order (engine_types.hpp MarchSched); the shader-order form is dispatched in
    // natural order: its launcher passes nulls, and the compiler sees so
    constexpr bool kSched = ARITH == GRV_ARITH_FAST;
    const unsigned long long sched_t0 = (kSched && sched.cost) ? wall_clock64() : 0ull;
    const uint32_t block = (kSched && sched.order) ? sched.order[blockIdx.x] : blockIdx.x;
    const uint32_t slot = block * kMarchBlock + threadIdx.x;
    uint32_t X = 0, Y = 0, oi = 0;
    const bool valid = slot < n_slots && slot_to_pixel(G, slot, X, Y, oi);
    uint32_t steps = 0;
    if (valid) {
        float o[3];
        steps = glsl_fragment<ARITH>(U, G.width, G.height, X, Y, o);
        if (out_rgba) out_rgba[oi] = make_float4(o[0], o[1], o[2], 1.0f);
        if (out_steps) out_steps[oi] = steps;
    }
    add_steps(total_steps, steps, block);
    if (kSched && sched.cost && threadIdx.x == 0) sched.cost[block] = (uint32_t)(wall_clock64() - sched_t0);
#ifdef GRV_MARCH_TIMELINE
    if (ARITH == GRV_ARITH_FAST && g_march_timeline) {
        unsigned long long v = steps;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63u) == 0u) {
            unsigned long long *t = g_march_timeline + 4ull * blockIdx.x;
            t[0] = tl0;
            t[1] = wall_clock64();
            t[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | // HW_ID
                   ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); // XCC_ID
            t[3] = v;
        }
    }
#endif
}

} // namespace
