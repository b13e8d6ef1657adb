// wgsl_fast_kernel.hpp -- FAST arithmetic contract of the WGSL compute march
// (src/shaders/compute.wgsl.ts:147-258), compiled with -ffp-contract=fast.
//
// Same equations as wgsl_symplectic_kernel (shader_kernels.hpp, shader operation order):
// Kerr-Schild Hamiltonian right-hand side (compute.wgsl.ts:42-120), implicit midpoint with two
// fixed-point sweeps (:122-133), h = clamp((r - r+) * 0.15, 0.05, 1) (:214), thin-disk crossing
// shading (:216-246).  What changes is the arithmetic, exactly as for the f64 FAST kernel
// (kerr_device.hpp: rhs_ks_geom):
//   * one v_rcp_f32 per right-hand side (1 / (Sigma sin^2)); 1/Sigma, 1/sin^2 are products;
//   * both force terms share the bracket W = 2 Sigma H + Sigma p_t^2;
//   * sin^2 and sin cos from one two-term Cody-Waite reduction + cephes minimax polynomials
//     (|x| <= pi/4, ~1 ulp f32), one swap and one sign instead of quadrant logic;
//   * t and phi never feed back: only the final midpoint evaluation forms dt, dphi;
//   * FMA contraction throughout.
// Results differ from the shader-order kernel by f32 rounding only; the parity tests hold both
// to the same statistical tolerance against the shader oracle (tests/test_shader_kernels.py).
#pragma once

#include "geodesic_kernels.hpp"

// the horizon exit of the FAST f32 marches: !(r >= r_stop), so that a non-finite state leaves as well
// (wgsl_pk_kernel.hpp); GRV_F32_NAN_EXIT=0 compiles the shader's literal r < r_stop for A/B runs
#ifndef GRV_F32_NAN_EXIT
#define GRV_F32_NAN_EXIT 1
#endif
#if GRV_F32_NAN_EXIT
#define GRV_F32_BELOW(r, lim) (!((r) >= (lim)))
#else
#define GRV_F32_BELOW(r, lim) ((r) < (lim))
#endif

namespace {

struct Wf32Hole {
    float M, a, a2, two_m;
};
struct Wf32Consts { // products of the constants of motion
    float pt, pph, pt2, m2pt, pph2, a_pph, two_a_pph;
};
struct Wf32Deriv {
    float dt, dr, dth, dph, dpr, dpth;
};

// geometry at (r, theta) + right-hand side; WANT_TPHI: also dt, dphi
template <bool WANT_TPHI>
__device__ __forceinline__ Wf32Deriv wf32_rhs(const Wf32Hole &bh, const Wf32Consts &c, float r,
                                              float theta, float p_r, float p_th) {
    // sin^2(theta), sin(theta) cos(theta)
#if GRV_TRIG_BITS
    // round-to-nearest-even(theta 2/pi) by adding 1.5 2^23 (|theta| < 2^21): the integer stays in the
    // sum's low mantissa bits, bit 0 is the quadrant's parity (kerr_device.hpp ks_geom, f64 twin)
    const float tq = fmaf(theta, 0.636619772367581343f, 12582912.0f);
    const float j = tq - 12582912.0f;
#else
    const float j = rintf(theta * 0.636619772367581343f); // 2/pi
#endif
    float x = fmaf(-j, 1.57079637050628662109375f, theta);
    x = fmaf(-j, -4.37113900018624283e-8f, x); // pi/2 = hi + lo
    const float z = x * x;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    const float sr = fmaf(x * z, ps, x);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    const float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    const float prod = sr * cr;
#if GRV_TRIG_BITS
    const int tl = __float_as_int(tq);
    const float sn = __int_as_float(bits_select(bits_sext_bit0(tl), __float_as_int(cr), __float_as_int(sr)));
    const float sc = __int_as_float(bits_shl_add(tl, 31, __float_as_int(prod)));
#else
    const bool odd = ((int)j & 1) != 0;
    const float sn = odd ? cr : sr;
    const float sc = odd ? -prod : prod;
#endif
    const float sin2 = fmaxf(sn * sn, 1e-12f); // compute.wgsl.ts:49

    const float r2a2 = fmaf(r, r, bh.a2);
    const float sigma = fmaf(-bh.a2, sin2, r2a2);
    const float delta = fmaf(-bh.two_m, r, r2a2);
    const float inv_ss = __builtin_amdgcn_rcpf(sigma * sin2);
    const float isig = inv_ss * sin2, isin2 = inv_ss * sigma;
    const float two_mr = bh.two_m * r;
    const float pt_mix = fmaf(c.m2pt, p_r, c.pt2);

    Wf32Deriv d;
    if (WANT_TPHI) {
        d.dt = fmaf(two_mr * isig, p_r - c.pt, -c.pt);
        d.dph = isig * fmaf(c.pph, isin2, bh.a * p_r);
    } else {
        d.dt = 0.0f;
        d.dph = 0.0f;
    }
    d.dr = isig * fmaf(two_mr, c.pt, fmaf(delta, p_r, c.a_pph));
    d.dth = isig * p_th;
    const float q = c.pph2 * isin2;
    const float pr2 = p_r * p_r;
    float w = fmaf(c.two_a_pph, p_r, -(two_mr * pt_mix));
    w = w + q;
    w = fmaf(p_th, p_th, w);
    w = fmaf(delta, pr2, w);
    const float ar_half = fmaf(-r, w, sigma * fmaf(r - bh.M, pr2, -(bh.M * pt_mix)));
    const float ath_half = sc * fmaf(bh.a2, w, -(sigma * (q * isin2)));
    const float isig2 = isig * isig;
    d.dpr = -(isig2 * ar_half);
    d.dpth = -(isig2 * ath_half); // the shader has no polar special case (compute.wgsl.ts:96-117)
    return d;
}

__device__ __forceinline__ void wf32_m4v4(const float *m, float x, float y, float z, float w, float o[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = m[0 + r] * x + m[4 + r] * y + m[8 + r] * z + m[12 + r] * w;
}

// seven waves per SIMD (72 VGPRs, no scratch): +2.6-3.3 % over six on the 4K and 1080p marches (cross-wave
// pairing of plain f32 ops, as for the GLSL march); eight spills 20 B into the loop and loses 2 %
// (profiles/r04_ab_wgsl_one_ray_waves.txt)
#ifndef GRV_WF32_WAVES
#define GRV_WF32_WAVES 7
#endif
__global__ __launch_bounds__(kMarchBlock) __attribute__((amdgpu_waves_per_eu(GRV_WF32_WAVES, GRV_WF32_WAVES))) void wgsl_symplectic_fast_kernel(FrameGeom G, WgslParams P,
                                                                      float4 *__restrict__ out_rgba,
                                                                      uint32_t *__restrict__ out_steps,
                                                                      unsigned long long *total_steps,
                                                                      uint32_t n_slots) {
    const uint32_t slot = dispatch_block(blockIdx.x, gridDim.x) * kMarchBlock + threadIdx.x;
    uint32_t X = 0, Y = 0, oi = 0;
    const bool valid = slot < n_slots && slot_to_pixel(G, slot, X, Y, oi);
    uint32_t steps = 0;
    if (valid) {
        // pixel -> state, compute.wgsl.ts:153-187
        const float PI = 3.14159265f;
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
        const float cx = P.position[0], cy = P.position[1], cz = P.position[2];
        const float r0 = sqrtf(cx * cx + cy * cy + cz * cz);
        const float theta0 = acosf(fminf(fmaxf(cy / r0, -1.0f), 1.0f));
        const float phi0 = atan2f(cz, cx);
        const float st = sinf(theta0), ct = cosf(theta0), sp = sinf(phi0), cp = cosf(phi0);
        const float pr_far = wx * (st * cp) + wy * ct + wz * (st * sp);
        const float pth_far = (wx * (ct * cp) + wy * (-st) + wz * (ct * sp)) / r0;
        const float pph_far = (wx * (-sp) + wz * cp) / (r0 * fmaxf(st, 1e-4f));

        float t = 0.0f, r = r0, th = theta0, ph = phi0;
        float p_r = pr_far, p_th = pth_far * r0 * r0;
        const float p_t = -1.0f, p_ph = pph_far * r0 * r0 * st * st;

        const float M = P.mass;
        const float a = P.spin * M;
        const Wf32Hole bh{M, a, a * a, 2.0f * M};
        const Wf32Consts c{p_t, p_ph, p_t * p_t, -2.0f * p_t, p_ph * p_ph, a * p_ph, 2.0f * a * p_ph};
        // compute.wgsl.ts:28-40
        const float disc = M * M - a * a;
        const float rh = disc < 0.0f ? M : M + sqrtf(disc);
        const float absS = fabsf(fminf(fmaxf(a / M, -0.999f), 0.999f));
        const float z1 = 1.0f + powf(1.0f - absS * absS, 1.0f / 3.0f) *
                                    (powf(1.0f + absS, 1.0f / 3.0f) + powf(1.0f - absS, 1.0f / 3.0f));
        const float z2 = sqrtf(3.0f * absS * absS + z1 * z1);
        const float isco = M * (3.0f + z2 - sqrtf((3.0f - z1) * (3.0f + z1 + 2.0f * z2)));
        const float r_stop = rh * 1.001f;

        float col[3] = {0.0f, 0.0f, 0.0f};
        float alpha = 0.0f;
        // one loop exit (as glsl_fragment.hpp's FAST march): the shader leaves from three places -- horizon
        // and far tests at the top, the opaque disk at the bottom; tested together at the top of the next
        // iteration the loop-carried state is merged once instead of once per exit.  The star hash of
        // the escape branch runs after the loop, on the rays that left through r > 100.
        int i = 0;
        bool opaque = false, below = false, far = false;
        for (;;) {
            below = GRV_F32_BELOW(r, r_stop); // NaN leaves here too: see wgsl_pk_kernel.hpp
            far = r > 100.0f;
            if (!(i < P.max_steps) || opaque || below || far) break;
            const float r_before = r, th_before = th;
            const float h = fminf(fmaxf((r - rh) * 0.15f, 0.05f), 1.0f);
            const float hh = 0.5f * h;
            // implicit midpoint, two fixed-point sweeps then the update (compute.wgsl.ts:122-133)
            Wf32Deriv d = wf32_rhs<false>(bh, c, r, th, p_r, p_th);
            float mr = fmaf(d.dr, hh, r), mth = fmaf(d.dth, hh, th);
            float mpr = fmaf(d.dpr, hh, p_r), mpth = fmaf(d.dpth, hh, p_th);
            d = wf32_rhs<false>(bh, c, mr, mth, mpr, mpth);
            mr = fmaf(d.dr, hh, r);
            mth = fmaf(d.dth, hh, th);
            mpr = fmaf(d.dpr, hh, p_r);
            mpth = fmaf(d.dpth, hh, p_th);
            d = wf32_rhs<true>(bh, c, mr, mth, mpr, mpth);
            t = fmaf(d.dt, h, t);
            r = fmaf(d.dr, h, r);
            th = fmaf(d.dth, h, th);
            ph = fmaf(d.dph, h, ph);
            p_r = fmaf(d.dpr, h, p_r);
            p_th = fmaf(d.dpth, h, p_th);
            ++i;
            // thin-disk plane crossing, shaded with the pre-step radius (compute.wgsl.ts:216-246)
            const float rb = r_before;
            if ((th_before - PI * 0.5f) * (th - PI * 0.5f) <= 0.0f && rb > isco && rb < 30.0f) {
                const float Omega = 1.0f / (powf(rb, 1.5f) + a);
                const float u_t =
                    1.0f / sqrtf(fmaxf(1.0f - 2.0f * M / rb - Omega * Omega * (rb * rb + a * a), 1e-4f));
                const float u_phi = Omega * u_t;
                const float g_factor = -p_t / fmaxf(-(u_t * p_t + u_phi * p_ph), 1e-4f);
                const float artistic_T = (1.0f / powf(fmaxf(rb / isco, 1.0f), 0.75f)) * g_factor;
                const float base[3] = {1.0f, 0.5f, 0.1f}, blue[3] = {0.5f, 0.7f, 1.0f},
                            red[3] = {1.0f, 0.2f, 0.0f};
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
            opaque = alpha > 0.99f;
        }
        steps = (uint32_t)i;
        // the escape branch (compute.wgsl.ts:199-206): taken by a ray that was neither opaque nor below the
        // horizon limit when the budget still had room
        if (P.stars && far && !below && !opaque && i < P.max_steps) {
            const float sx = p_r, sy = p_th / r, sz = p_ph / (r * fmaxf(st, 1e-4f));
            const float inv = __builtin_amdgcn_rsqf(sx * sx + sy * sy + sz * sz);
            const float sn = sinf((sx * 12.9898f + sy * 78.233f + sz * 45.164f) * inv) * 43758.5453f;
            if (sn - floorf(sn) > 0.999f)
                for (int k = 0; k < 3; ++k) col[k] += 1.0f * (1.0f - alpha);
        }
        (void)t;
        (void)ph;
        if (out_rgba) out_rgba[oi] = make_float4(col[0], col[1], col[2], 1.0f);
        if (out_steps) out_steps[oi] = steps;
    }
    // one atomic per block for the frame's step total
    __shared__ unsigned long long s_w[kMarchBlock / 64];
    unsigned long long v = steps;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
#pragma unroll
        for (int w = 0; w < kMarchBlock / 64; ++w) tot += s_w[w];
        if (tot) atomicAdd(total_steps + (blockIdx.x % kStepParts) * kStepPartStride, tot); // FrameStatsDev::steps_part
    }
}

} // namespace
