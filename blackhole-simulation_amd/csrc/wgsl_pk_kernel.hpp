// wgsl_pk_kernel.hpp -- the FAST WGSL compute march with TWO rays per lane, so that the
// arithmetic runs on CDNA's packed-f32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two
// f32 results per lane per issue slot).  Same equations and the same FAST arithmetic contract as
// wgsl_symplectic_fast_kernel (wgsl_fast_kernel.hpp; src/shaders/compute.wgsl.ts:42-258); each
// lane carries a horizontal pair of slots as 2-vectors, every expression is written once on the
// pair, and only the ops without a packed form (rounding, compares/selects, v_rcp_f32, max) issue
// per component.  A ray that has finished is frozen by select while its partner runs on.
#pragma once

#include "wgsl_fast_kernel.hpp"

#ifndef GRV_PK_WAVES
#define GRV_PK_WAVES 4 // waves per SIMD the packed march is compiled for (5 spills: measured slower)
#endif
#ifndef GRV_PK_FREEZE_BY_STEP
#define GRV_PK_FREEZE_BY_STEP 1
#endif

namespace {

typedef float f2_t __attribute__((ext_vector_type(2)));
typedef int i2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2_t pk_splat(float x) { return f2_t{x, x}; }
__device__ __forceinline__ f2_t pk_fma(f2_t a, f2_t b, f2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2_t pk_sel(i2_t m, f2_t a, f2_t b) { // m != 0 ? a : b, per component
    return f2_t{m.x ? a.x : b.x, m.y ? a.y : b.y};
}
__device__ __forceinline__ f2_t pk_rcp(f2_t x) {
    return f2_t{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
}
__device__ __forceinline__ f2_t pk_max(f2_t a, float b) { return f2_t{fmaxf(a.x, b), fmaxf(a.y, b)}; }

struct PkDeriv {
    f2_t dr, dth, dpr, dpth;
};
struct PkConsts { // per-pair products of the constants of motion (p_t = -1 for every ray)
    f2_t pph, pph2, a_pph, two_a_pph;
};

// right-hand side on a pair of rays; same algebra as wf32_rhs (W bracket, one reciprocal)
__device__ __forceinline__ PkDeriv pk_rhs(const Wf32Hole &bh, const PkConsts &c, f2_t r, f2_t theta,
                                          f2_t p_r, f2_t p_th) {
#if GRV_TRIG_BITS
    // packed round-to-nearest-even by the 1.5 2^23 sum (wgsl_fast_kernel.hpp): one v_pk_fma + one
    // v_pk_add instead of a v_pk_mul and two v_rndne, and the parity bit is already in an integer lane
    const f2_t tq = pk_fma(theta, pk_splat(0.636619772367581343f), pk_splat(12582912.0f));
    const f2_t jf = tq - 12582912.0f;
#else
    const f2_t jf = __builtin_elementwise_rint(theta * 0.636619772367581343f);
#endif
    f2_t x = pk_fma(jf, pk_splat(-1.57079637050628662109375f), theta);
    x = pk_fma(jf, pk_splat(4.37113900018624283e-8f), x);
    const f2_t z = x * x;
    f2_t ps = pk_fma(z, pk_splat(-1.9515295891e-4f), pk_splat(8.3321608736e-3f));
    ps = pk_fma(z, ps, pk_splat(-1.6666654611e-1f));
    const f2_t sr = pk_fma(x * z, ps, x);
    f2_t pc = pk_fma(z, pk_splat(2.443315711809948e-5f), pk_splat(-1.388731625493765e-3f));
    pc = pk_fma(z, pc, pk_splat(4.166664568298827e-2f));
    const f2_t cr = pk_fma(z * z, pc, pk_fma(z, pk_splat(-0.5f), pk_splat(1.0f)));
    const f2_t prod = sr * cr;
#if GRV_TRIG_BITS
    const int tx = __float_as_int(tq.x), ty = __float_as_int(tq.y);
    const f2_t sn = f2_t{__int_as_float(bits_select(bits_sext_bit0(tx), __float_as_int(cr.x), __float_as_int(sr.x))),
                         __int_as_float(bits_select(bits_sext_bit0(ty), __float_as_int(cr.y), __float_as_int(sr.y)))};
    const f2_t sc = f2_t{__int_as_float(bits_shl_add(tx, 31, __float_as_int(prod.x))),
                         __int_as_float(bits_shl_add(ty, 31, __float_as_int(prod.y)))};
#else
    const i2_t odd = i2_t{(int)jf.x & 1, (int)jf.y & 1};
    const f2_t sn = pk_sel(odd, cr, sr);
    const f2_t sc = pk_sel(odd, -prod, prod);
#endif
    const f2_t sin2 = pk_max(sn * sn, 1e-12f);

    const f2_t r2a2 = pk_fma(r, r, pk_splat(bh.a2));
    const f2_t sigma = pk_fma(pk_splat(-bh.a2), sin2, r2a2);
    const f2_t delta = pk_fma(pk_splat(-bh.two_m), r, r2a2);
    const f2_t inv_ss = pk_rcp(sigma * sin2);
    const f2_t isig = inv_ss * sin2, isin2 = inv_ss * sigma;
    const f2_t two_mr = r * bh.two_m;
    // p_t = -1: p_t^2 - 2 p_t p_r = 1 + 2 p_r
    const f2_t pt_mix = pk_fma(p_r, pk_splat(2.0f), pk_splat(1.0f));

    PkDeriv d;
    d.dr = isig * (pk_fma(delta, p_r, c.a_pph) - two_mr);
    d.dth = isig * p_th;
    const f2_t q = c.pph2 * isin2;
    const f2_t pr2 = p_r * p_r;
    f2_t w = pk_fma(c.two_a_pph, p_r, -(two_mr * pt_mix));
    w = w + q;
    w = pk_fma(p_th, p_th, w);
    w = pk_fma(delta, pr2, w);
    const f2_t ar_half = pk_fma(-r, w, sigma * pk_fma(r - bh.M, pr2, -(pt_mix * bh.M)));
    const f2_t ath_half = sc * pk_fma(pk_splat(bh.a2), w, -(sigma * (q * isin2)));
    const f2_t isig2 = isig * isig;
    d.dpr = -(isig2 * ar_half);
    d.dpth = -(isig2 * ath_half);
    return d;
}

// pixel -> initial state of one slot (compute.wgsl.ts:153-187); false if the slot holds no pixel
__device__ __forceinline__ bool pk_init_slot(const FrameGeom &G, const WgslParams &P, uint32_t slot,
                                             uint32_t n_slots, float r0, float st, float ct, float sp,
                                             float cp, float &p_r, float &p_th, float &p_ph, uint32_t &oi) {
    uint32_t X = 0, Y = 0;
    oi = 0;
    p_r = -1.0f;
    p_th = 0.0f;
    p_ph = 0.0f;
    if (!(slot < n_slots && slot_to_pixel(G, slot, X, Y, oi))) return false;
    const float fw = (float)G.width, fh = (float)G.height;
    const float ndcx = ((float)X / fw + P.jitter[0] / fw) * 2.0f - 1.0f;
    const float ndcy = ((float)Y / fh + P.jitter[1] / fh) * 2.0f - 1.0f;
    float vt[4], w4[4];
    wf32_m4v4(P.inv_proj, ndcx, -ndcy, 1.0f, 1.0f, vt);
    float vx = vt[0] / vt[3], vy = vt[1] / vt[3], vz = vt[2] / vt[3];
    float len = sqrtf(vx * vx + vy * vy + vz * vz);
    vx /= len;
    vy /= len;
    vz /= len;
    wf32_m4v4(P.inv_view, vx, vy, vz, 0.0f, w4);
    len = sqrtf(w4[0] * w4[0] + w4[1] * w4[1] + w4[2] * w4[2]);
    const float wx = w4[0] / len, wy = w4[1] / len, wz = w4[2] / len;
    const float pr_far = wx * (st * cp) + wy * ct + wz * (st * sp);
    const float pth_far = (wx * (ct * cp) + wy * (-st) + wz * (ct * sp)) / r0;
    const float pph_far = (wx * (-sp) + wz * cp) / (r0 * fmaxf(st, 1e-4f));
    p_r = pr_far;
    p_th = pth_far * r0 * r0;
    p_ph = pph_far * r0 * r0 * st * st;
    return true;
}

// thin-disk crossing shading of one ray (compute.wgsl.ts:216-246), pre-step radius rb
__device__ __forceinline__ void pk_shade(float rb, float M, float a, float isco, float p_ph, float col[3],
                                         float &alpha) {
    const float p_t = -1.0f;
    const float Omega = 1.0f / (powf(rb, 1.5f) + a);
    const float u_t = 1.0f / sqrtf(fmaxf(1.0f - 2.0f * M / rb - Omega * Omega * (rb * rb + a * a), 1e-4f));
    const float u_phi = Omega * u_t;
    const float g_factor = -p_t / fmaxf(-(u_t * p_t + u_phi * p_ph), 1e-4f);
    const float artistic_T = (1.0f / powf(fmaxf(rb / isco, 1.0f), 0.75f)) * g_factor;
    const float base[3] = {1.0f, 0.5f, 0.1f}, blue[3] = {0.5f, 0.7f, 1.0f}, red[3] = {1.0f, 0.2f, 0.0f};
    const float bs = fmaxf(g_factor - 1.0f, 0.0f), rs = fmaxf(1.0f - g_factor, 0.0f) * 0.5f;
    const float target_opacity = 0.6f * artistic_T;
    const float g4 = powf(g_factor, 4.0f);
    const float mri_sat = 1.0f + 0.0001f * sinf(rb * 100.0f * powf(rb, -1.5f));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float target = (base[k] + blue[k] * bs - red[k] * rs) * artistic_T * 4.0f;
        const float I_em = target * target_opacity / fmaxf(g4, 1e-5f);
        col[k] += g4 * (I_em * mri_sat) * (1.0f - alpha);
    }
    alpha += target_opacity * mri_sat;
}

// One-wave blocks (kMarchBlock): a finished wave's slot goes back at once.  On long marches, where a few waves
// of a block would run on after the others, that is worth 10 % (8K x 1024 steps: 285 -> 314 G ray-steps/s,
// profiles/r03_ab_march_block.jsonl).  Short marches (1080p x 512 steps) used to prefer four-wave blocks
// (230 G against 213 G in natural order); with the measured-cost dispatch order the finer blocks are the
// better schedule there as well (+2.0 % / +1.5 %, profiles/r05_ab_pk_short_one_wave.jsonl), and the four-wave
// form is gone.  (The body stays written straight into the kernel: a shared __device__ function taking the
// uniform blocks by reference compiled to 128 VGPRs and ran 5 % slower: 122 VGPRs.)
__global__ __launch_bounds__(kMarchBlock) __attribute__((amdgpu_waves_per_eu(GRV_PK_WAVES, GRV_PK_WAVES)))
void wgsl_symplectic_pk_kernel(FrameGeom G, WgslParams P, float4 *__restrict__ out_rgba,
                               uint32_t *__restrict__ out_steps, unsigned long long *total_steps, uint32_t n_slots,
                               MarchSched sched) {
    const unsigned long long sched_t0 = sched.cost ? wall_clock64() : 0ull;
    const uint32_t pk_block = sched.order ? sched.order[blockIdx.x] : dispatch_block(blockIdx.x, gridDim.x);
    const uint32_t pair = pk_block * kMarchBlock + threadIdx.x;
    const uint32_t slot0 = 2u * pair;
    const float PI = 3.14159265f;
    // camera terms shared by every ray (compute.wgsl.ts:170-178)
    const float cx = P.position[0], cy = P.position[1], cz = P.position[2];
    const float r0 = sqrtf(cx * cx + cy * cy + cz * cz);
    const float theta0 = acosf(fminf(fmaxf(cy / r0, -1.0f), 1.0f));
    const float phi0 = atan2f(cz, cx);
    const float st = sinf(theta0), ct = cosf(theta0), sp = sinf(phi0), cp = cosf(phi0);

    float pr0, pth0, pph0, pr1, pth1, pph1;
    uint32_t oi0, oi1;
    const bool v0 = pk_init_slot(G, P, slot0, n_slots, r0, st, ct, sp, cp, pr0, pth0, pph0, oi0);
    const bool v1 = pk_init_slot(G, P, slot0 + 1u, n_slots, r0, st, ct, sp, cp, pr1, pth1, pph1, oi1);

    const float M = P.mass;
    const float a = P.spin * M;
    const Wf32Hole bh{M, a, a * a, 2.0f * M};
    const float disc = M * M - a * a;
    const float rh = disc < 0.0f ? M : M + sqrtf(disc);
    const float absS = fabsf(fminf(fmaxf(a / M, -0.999f), 0.999f));
    const float z1 = 1.0f + powf(1.0f - absS * absS, 1.0f / 3.0f) *
                                (powf(1.0f + absS, 1.0f / 3.0f) + powf(1.0f - absS, 1.0f / 3.0f));
    const float z2 = sqrtf(3.0f * absS * absS + z1 * z1);
    const float isco = M * (3.0f + z2 - sqrtf((3.0f - z1) * (3.0f + z1 + 2.0f * z2)));
    const float r_stop = rh * 1.001f;

    f2_t r = pk_splat(r0), th = pk_splat(theta0);
    f2_t p_r = f2_t{pr0, pr1}, p_th = f2_t{pth0, pth1};
    const f2_t p_ph = f2_t{pph0, pph1};
    const PkConsts c{p_ph, p_ph * p_ph, p_ph * a, p_ph * (2.0f * a)};
    float col[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    float alpha[2] = {0.0f, 0.0f};
    uint32_t steps[2] = {0u, 0u};
    bool live[2] = {v0, v1};

    for (int i = 0; i < P.max_steps; ++i) {
        // loop-top exits of the shader, per ray
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float rk = k ? r.y : r.x;
            // !(r >= r_stop), not r < r_stop: a ray that crosses the polar axis blows up in the shader's
            // spherical coordinates (sin^2 clamped at 1e-12, compute.wgsl.ts:49) in every arithmetic; the
            // shader order and a double evaluation land on a huge negative r there and leave through
            // this test, the FAST forms land on NaN, which must leave through it too -- or the ray marches
            // its whole budget as NaN (profiles/r04_c4_budget_rays.json: 3 200 such rays at 8K)
            if (live[k] && (GRV_F32_BELOW(rk, r_stop) || alpha[k] > 0.99f)) live[k] = false;
            if (live[k] && rk > 100.0f) { // star hash of the escape branch, compute.wgsl.ts:199-206
                if (P.stars) {
                    const float sx = k ? p_r.y : p_r.x, sy = (k ? p_th.y : p_th.x) / rk,
                                sz = (k ? p_ph.y : p_ph.x) / (rk * fmaxf(st, 1e-4f));
                    const float inv = __builtin_amdgcn_rsqf(sx * sx + sy * sy + sz * sz);
                    const float sn = sinf((sx * 12.9898f + sy * 78.233f + sz * 45.164f) * inv) * 43758.5453f;
                    if (sn - floorf(sn) > 0.999f)
                        for (int q = 0; q < 3; ++q) col[k][q] += 1.0f * (1.0f - alpha[k]);
                }
                live[k] = false;
            }
        }
        if (__ballot(live[0] || live[1]) == 0ull) break;
        if (live[0] || live[1]) {
            const f2_t r_before = r, th_before = th;
            f2_t h = (r - rh) * 0.15f;
            h = f2_t{fminf(fmaxf(h.x, 0.05f), 1.0f), fminf(fmaxf(h.y, 0.05f), 1.0f)};
            const f2_t hh = h * 0.5f;
            PkDeriv d = pk_rhs(bh, c, r, th, p_r, p_th);
            f2_t mr = pk_fma(d.dr, hh, r), mth = pk_fma(d.dth, hh, th);
            f2_t mpr = pk_fma(d.dpr, hh, p_r), mpth = pk_fma(d.dpth, hh, p_th);
            d = pk_rhs(bh, c, mr, mth, mpr, mpth);
            mr = pk_fma(d.dr, hh, r);
            mth = pk_fma(d.dth, hh, th);
            mpr = pk_fma(d.dpr, hh, p_r);
            mpth = pk_fma(d.dpth, hh, p_th);
            d = pk_rhs(bh, c, mr, mth, mpr, mpth);
#if GRV_PK_FREEZE_BY_STEP
            // a finished ray stays where it ended: its step is 0 (two selects per pair instead of eight;
            // x + 0 d = x for every finite derivative, and a frozen ray's state is read by nothing but
            // the guarded loop-top tests)
            const f2_t hs = f2_t{live[0] ? h.x : 0.0f, live[1] ? h.y : 0.0f};
            r = pk_fma(d.dr, hs, r);
            th = pk_fma(d.dth, hs, th);
            p_r = pk_fma(d.dpr, hs, p_r);
            p_th = pk_fma(d.dpth, hs, p_th);
#else
            const i2_t m = i2_t{live[0] ? 1 : 0, live[1] ? 1 : 0};
            r = pk_sel(m, pk_fma(d.dr, h, r), r); // a finished ray stays where it ended
            th = pk_sel(m, pk_fma(d.dth, h, th), th);
            p_r = pk_sel(m, pk_fma(d.dpr, h, p_r), p_r);
            p_th = pk_sel(m, pk_fma(d.dpth, h, p_th), p_th);
#endif
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!live[k]) continue;
                ++steps[k];
                const float rb = k ? r_before.y : r_before.x;
                const float tb = k ? th_before.y : th_before.x, tn = k ? th.y : th.x;
                if ((tb - PI * 0.5f) * (tn - PI * 0.5f) <= 0.0f && rb > isco && rb < 30.0f)
                    pk_shade(rb, M, a, isco, k ? p_ph.y : p_ph.x, col[k], alpha[k]);
            }
        }
    }
    if (v0) {
        if (out_rgba) out_rgba[oi0] = make_float4(col[0][0], col[0][1], col[0][2], 1.0f);
        if (out_steps) out_steps[oi0] = steps[0];
    }
    if (v1) {
        if (out_rgba) out_rgba[oi1] = make_float4(col[1][0], col[1][1], col[1][2], 1.0f);
        if (out_steps) out_steps[oi1] = steps[1];
    }
    __shared__ unsigned long long s_w[kMarchBlock / 64];
    unsigned long long v = (unsigned long long)steps[0] + steps[1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
#pragma unroll
        for (int w = 0; w < kMarchBlock / 64; ++w) tot += s_w[w];
        // (slot by the block's own index, which is live for the cost store below: by blockIdx.x the 8K march measured -0.45 %)
        if (tot) atomicAdd(total_steps + (pk_block % kStepParts) * kStepPartStride, tot); // FrameStatsDev::steps_part
        if (sched.cost) sched.cost[pk_block] = (uint32_t)(wall_clock64() - sched_t0);
    }
}

} // namespace
