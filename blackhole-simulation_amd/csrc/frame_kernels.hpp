// frame_kernels.hpp -- finalize / shading / LUT kernels (compiled once, in the
// -ffp-contract=off translation unit).
//
// Reference behaviour restated:
//   kerr_g_factor             gravitas-core/src/physics/redshift.rs:65-95
//   generate_blackbody_lut    gravitas-core/src/physics/spectrum.rs:76-102
//   planck_law / integrate_planck_xyz / cie_1931 / xyz_to_linear_rgb  spectrum.rs:12-70
//   disk temperature profile  src/shaders/blackhole/chunks/disk.ts:100-102
//   Trajectory fields         gravitas-core/src/geodesic/mod.rs:149-161
#pragma once

#include <hip/hip_fp16.h>
#include "geodesic_kernels.hpp"

namespace {

// ---------------------------------------------------------------------------
// wave / block reductions for the frame statistics
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// per-thread partial statistics, reduced once per block
struct StatAcc {
    unsigned long long steps = 0, tries = 0, cross = 0, rays = 0;
    unsigned long long tc[5] = {0, 0, 0, 0, 0};
    double drift = 0.0;
    __device__ __forceinline__ void add(uint32_t s, uint32_t t, uint32_t term, uint32_t nc, double d) {
        steps += s;
        tries += t;
        cross += nc;
        rays += 1;
#pragma unroll
        for (int k = 0; k < 5; ++k) tc[k] += (term == (uint32_t)k) ? 1ull : 0ull;
        drift = fmax(drift, d);
    }
};

// all threads of the block must call this (wave shuffles + LDS + one atomic set per block)
__device__ __forceinline__ void flush_stats(FrameStatsDev *st, const StatAcc &a) {
    __shared__ unsigned long long s_part[16][9];
    __shared__ double s_drift[16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_waves = (blockDim.x + 63u) >> 6;
    unsigned long long v[9] = {a.steps, a.tries, a.cross, a.rays, a.tc[0], a.tc[1], a.tc[2], a.tc[3], a.tc[4]};
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = wave_sum_u64(v[k]);
    const double d = wave_max_f64(a.drift);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s_part[wave][k] = v[k];
        s_drift[wave] = d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double dm = 0.0;
        for (uint32_t w = 0; w < n_waves; ++w) {
#pragma unroll
            for (int k = 0; k < 9; ++k) t[k] += s_part[w][k];
            dm = fmax(dm, s_drift[w]);
        }
        if (t[3] != 0ull) {
            atomicAdd(&st->accepted_steps, t[0]);
            atomicAdd(&st->rkf_tries, t[1]);
            atomicAdd(&st->crossings, t[2]);
            atomicAdd(&st->rays, t[3]);
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (t[4 + k]) atomicAdd(&st->term_count[k], t[4 + k]);
            // non-negative doubles order like their bit patterns
            atomicMax(&st->max_drift_bits, (unsigned long long)__double_as_longlong(dm));
        }
    }
}

// ---------------------------------------------------------------------------
// batch finalize: SoA workspace -> AoS GeodesicState + Trajectory scalars
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void finalize_batch_kernel(
    RayWorkspace ws, double *__restrict__ out_states, uint32_t *__restrict__ out_steps,
    uint8_t *__restrict__ out_term, double *__restrict__ out_drift, FrameStatsDev *st) {
    StatAcc acc;
    // (block size is a launch property: one-wave blocks for worker-sized batches on the control stream)
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ws.n; i += stride) {
        const uint32_t steps = ws.steps[i];
        const uint32_t tries = ws.tries[i];
        const uint32_t flags = ws.flags[i];
        const double drift = ws.drift[i];
        if (out_states) {
            double2 *dst = reinterpret_cast<double2 *>(out_states + (size_t)i * 8);
            dst[0] = make_double2(ws.t[i], ws.r[i]);
            dst[1] = make_double2(ws.th[i], ws.ph[i]);
            dst[2] = make_double2(ws.pt[i], ws.pr[i]);
            dst[3] = make_double2(ws.pth[i], ws.pph[i]);
        }
        if (out_steps) out_steps[i] = steps;
        if (out_term) out_term[i] = (uint8_t)(flags & kFlagTermMask);
        if (out_drift) out_drift[i] = drift;
        acc.add(steps, tries, flags & kFlagTermMask, (flags & kFlagCrossMask) >> kFlagCrossShift, drift);
    }
    if (st) flush_stats(st, acc);
}

// ---------------------------------------------------------------------------
// shading
// ---------------------------------------------------------------------------
__device__ __forceinline__ double kerr_g_factor_dev(double r, double mass, double spin,
                                                    double lambda) {
    const double a = spin * mass;
    const double r2 = r * r;
    const double a2 = a * a;
    const double m = mass;
    const double omega = sqrt(m) / (pow_rs(r, 1.5) + a * sqrt(m));
    const double sigma = r2;
    const double g_tt = -(1.0 - 2.0 * m * r / sigma);
    const double g_tphi = -(2.0 * m * r * a) / sigma;
    const double g_phiphi = r2 + a2 + 2.0 * m * r * a2 / sigma;
    const double ut_denom = -g_tt - 2.0 * omega * g_tphi - omega * omega * g_phiphi;
    if (ut_denom <= 0.0) return 0.0;
    const double ut = 1.0 / sqrt(ut_denom);
    const double factor = 1.0 - lambda * omega;
    if (fabs(factor) < 1e-30) return 0.0;
    return 1.0 / (ut * factor);
}

__device__ __forceinline__ double disk_temp_profile_dev(double r, double disk_inner) {
    double isco_r = disk_inner / r;
    isco_r = isco_r < 0.0 ? 0.0 : (isco_r > 1.0 ? 1.0 : isco_r);
    const double nt = fmax(0.0, 1.0 - sqrt(isco_r));
    return pow_rs(isco_r, 0.75) * pow_rs(nt, 0.25);
}

// Page-Thorne profile: the 512-entry table of generate_temperature_lut (physics/disk.rs:175-201),
// entry i at r = rin + (i / 511)(rout - rin), normalised to its maximum; read as the LINEAR +
// CLAMP_TO_EDGE texture the renderer uploads it as (src/rendering/webgl/renderer.ts:436-446), i.e.
// linear interpolation between the two entries around the continuous index, clamped at both ends.
__device__ __forceinline__ double disk_lut_profile_dev(const float *lut_s, double r, double rin,
                                                       double rout) {
    const double last = (double)(kDiskLutWidth - 1u);
    double x = (r - rin) / (rout - rin) * last;
    if (!(x > 0.0)) x = 0.0;
    if (x > last) x = last;
    const uint32_t i0 = (uint32_t)x;
    const uint32_t i1 = (i0 + 1u < kDiskLutWidth) ? i0 + 1u : i0;
    const double t0 = lut_s[i0], t1 = lut_s[i1];
    return t0 + (t1 - t0) * (x - (double)i0);
}

// texel fetch: LDS for the staged rows [row0, row0+rows), HBM/L2 otherwise
__device__ __forceinline__ float4 lut_texel(const float4 *__restrict__ lut_g, const float4 *lut_s,
                                            const ShadeParams &S, uint32_t x, uint32_t y) {
    const uint32_t rel = y - S.lds_row0;
    if (rel < S.lds_rows) return lut_s[rel * S.lut_w + x];
    return lut_g[(size_t)y * S.lut_w + x];
}

// frame finalize + shade.  Persistent blocks: each block stages its LUT rows in
// LDS once, then grid-strides over the slots.
__global__ __launch_bounds__(1024) void finalize_frame_kernel(
    RayWorkspace ws, FrameGeom G, ShadeParams S, int shading, const float4 *__restrict__ lut,
    const float *__restrict__ disk_lut, float4 *__restrict__ out_rgba, double *__restrict__ out_states,
    uint32_t *__restrict__ out_steps, uint8_t *__restrict__ out_term,
    double *__restrict__ out_drift, FrameStatsDev *st, uint32_t *__restrict__ wave_cost) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4 *lut_s = reinterpret_cast<float4 *>(smem_raw);
    // the Page-Thorne table (2 KB) sits behind the staged Planck rows
    float *disk_s = reinterpret_cast<float *>(lut_s + (size_t)S.lds_rows * S.lut_w);
    const bool page_thorne = shading && lut && disk_lut && S.disk_profile == GRV_DISK_PROFILE_PAGE_THORNE;
    if (shading && lut) {
        const uint32_t n4 = S.lds_rows * S.lut_w;
        const float4 *src = lut + (size_t)S.lds_row0 * S.lut_w;
        for (uint32_t k = threadIdx.x; k < n4; k += blockDim.x) lut_s[k] = src[k];
        if (page_thorne)
            for (uint32_t k = threadIdx.x; k < kDiskLutWidth; k += blockDim.x) disk_s[k] = disk_lut[k];
    }
    __syncthreads();

    StatAcc acc;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < ws.n; slot += stride) {
        uint32_t X, Y, oi;
        bool valid = false;
        uint32_t steps = 0, tries = 0, flags = 0;
        double drift = 0.0;
        valid = slot_to_pixel(G, slot, X, Y, oi);
        if (valid) {
            steps = ws.steps[slot];
            tries = ws.tries[slot];
            flags = ws.flags[slot];
            drift = ws.drift[slot];
            const double pt = ws.pt[slot], pph = ws.pph[slot];
            if (out_states) {
                double2 *dst = reinterpret_cast<double2 *>(out_states + (size_t)oi * 8);
                dst[0] = make_double2(ws.t[slot], ws.r[slot]);
                dst[1] = make_double2(ws.th[slot], ws.ph[slot]);
                dst[2] = make_double2(pt, ws.pr[slot]);
                dst[3] = make_double2(ws.pth[slot], pph);
            }
            if (out_steps) out_steps[oi] = steps;
            if (out_term) out_term[oi] = (uint8_t)(flags & kFlagTermMask);
            if (out_drift) out_drift[oi] = drift;
            if (out_rgba) {
                double col[3] = {0.0, 0.0, 0.0};
                if (shading && lut) {
                    uint32_t nc = (flags & kFlagCrossMask) >> kFlagCrossShift;
                    nc = nc < (uint32_t)kMaxCrossRec ? nc : (uint32_t)kMaxCrossRec;
                    double alpha = 0.0;
                    const double lambda = pph / (-pt);
                    for (uint32_t c = 0; c < nc; ++c) {
                        const double r_c = ws.rc[(size_t)c * ws.n + slot];
                        const double g = kerr_g_factor_dev(r_c, S.M, S.spin, lambda);
                        const double temp =
                            S.disk_temp * (page_thorne ? disk_lut_profile_dev(disk_s, r_c, S.pt_rin, S.pt_rout)
                                                       : disk_temp_profile_dev(r_c, S.disk_inner));
                        // inverse LUT axes (spectrum.rs:82,85)
                        const double u = pow_rs(fmax(temp, 0.0) / S.lut_max_temp, 1.0 / 2.5);
                        double fx = u * (double)(S.lut_w > 1 ? S.lut_w - 1 : 1);
                        double fy = (g - 0.05) / (5.0 - 0.05) * (double)(S.lut_h > 1 ? S.lut_h - 1 : 1);
                        if (!(fx > 0.0)) fx = 0.0;
                        if (!(fy > 0.0)) fy = 0.0;
                        if (fx > (double)(S.lut_w - 1)) fx = (double)(S.lut_w - 1);
                        if (fy > (double)(S.lut_h - 1)) fy = (double)(S.lut_h - 1);
                        const uint32_t x0 = (uint32_t)fx, y0 = (uint32_t)fy;
                        const uint32_t x1 = (x0 + 1 < S.lut_w) ? x0 + 1 : x0;
                        const uint32_t y1 = (y0 + 1 < S.lut_h) ? y0 + 1 : y0;
                        const double tx = fx - (double)x0, ty = fy - (double)y0;
                        const float4 t00 = lut_texel(lut, lut_s, S, x0, y0);
                        const float4 t10 = lut_texel(lut, lut_s, S, x1, y0);
                        const float4 t01 = lut_texel(lut, lut_s, S, x0, y1);
                        const float4 t11 = lut_texel(lut, lut_s, S, x1, y1);
                        const double a00[3] = {t00.x, t00.y, t00.z}, a10[3] = {t10.x, t10.y, t10.z};
                        const double a01[3] = {t01.x, t01.y, t01.z}, a11[3] = {t11.x, t11.y, t11.z};
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            const double top = a00[ch] + (a10[ch] - a00[ch]) * tx;
                            const double bot = a01[ch] + (a11[ch] - a01[ch]) * tx;
                            const double v = top + (bot - top) * ty;
                            col[ch] += v * S.exposure * (1.0 - alpha);
                        }
                        alpha += S.disk_opacity;
                    }
                }
                out_rgba[oi] = make_float4((float)col[0], (float)col[1], (float)col[2], 1.0f);
            }
        }
        if (valid)
            acc.add(steps, tries, flags & kFlagTermMask, (flags & kFlagCrossMask) >> kFlagCrossShift, drift);
        if (wave_cost) {
            // 64 consecutive slots = one wave of the segment kernel: it ran as long as its slowest ray.  The
            // next frame's one-launch dispatch starts the longest waves first (SegmentParams.order).
            uint32_t m = tries;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)m, off);
                m = o > m ? o : m;
            }
            if ((threadIdx.x & 63u) == 0u) wave_cost[slot >> 6] = (m < 2047u ? m : 2047u) << kWaveCostShift;
        }
    }
    if (st) flush_stats(st, acc);
}

// ---------------------------------------------------------------------------
// rank-0 de-interleave after the tile gather: packed [tile_local][64][64] pixels of
// `G.tile_rank` -> row-major image.  One thread per 4-byte word of a pixel.
// (tile grid as physics-engine/_legacy_src/tiling.rs:38-56)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void unpack_tiles_kernel(FrameGeom G,
                                                              const uint32_t *__restrict__ packed,
                                                              uint32_t *__restrict__ image,
                                                              uint32_t words_per_pixel) {
    const size_t total = (size_t)G.n_tiles_local * 4096u * words_per_pixel;
    for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < total;
         k += (size_t)gridDim.x * kBlock) {
        const uint32_t w = (uint32_t)(k % words_per_pixel);
        const size_t pix = k / words_per_pixel;
        const uint32_t tile_local = (uint32_t)(pix >> 12);
        const uint32_t within = (uint32_t)(pix & 4095u);
        const uint32_t px = within & 63u, py = within >> 6;
        const uint32_t tile = tile_local * G.tile_world + G.tile_rank;
        const uint32_t X = (tile % G.tiles_x) * 64u + px, Y = (tile / G.tiles_x) * 64u + py;
        if (X < G.width && Y < G.height)
            image[((size_t)Y * G.width + X) * words_per_pixel + w] = packed[k];
    }
}

// 16-byte pixels (RGBA f32, the frame path's format): one thread moves one pixel, a wave one 1-KiB tile
// row -- a coalesced read and a coalesced write, index arithmetic in shifts.
__global__ __launch_bounds__(kBlock) void unpack_tiles16_kernel(FrameGeom G, const uint4 *__restrict__ packed,
                                                                uint4 *__restrict__ image) {
    const size_t total = (size_t)G.n_tiles_local * 4096u;
    for (size_t pix = (size_t)blockIdx.x * kBlock + threadIdx.x; pix < total; pix += (size_t)gridDim.x * kBlock) {
        const uint32_t tile_local = (uint32_t)(pix >> 12);
        const uint32_t within = (uint32_t)(pix & 4095u);
        const uint32_t px = within & 63u, py = within >> 6;
        const uint32_t tile = tile_local * G.tile_world + G.tile_rank;
        const uint32_t X = (tile % G.tiles_x) * 64u + px, Y = (tile / G.tiles_x) * 64u + py;
        if (X < G.width && Y < G.height) image[(size_t)Y * G.width + X] = packed[pix];
    }
}

// The exchange in the reference's own output format (rgba16float storage texture of the compute pass,
// src/rendering/webgpu/renderer.ts:163-176): a rank's RGBA f32 share -> four binary16 channels per
// pixel (round to nearest even, the conversion the post chain's post_store applies), 8 B instead of 16
// on the wire; rank 0 widens them again while it de-interleaves the tiles.
__device__ __forceinline__ uint2 rgba_to_half4(float4 c) {
    const uint32_t x = __half_as_ushort(__float2half_rn(c.x)), y = __half_as_ushort(__float2half_rn(c.y));
    const uint32_t z = __half_as_ushort(__float2half_rn(c.z)), w = __half_as_ushort(__float2half_rn(c.w));
    return make_uint2(x | (y << 16), z | (w << 16));
}
__device__ __forceinline__ float4 half4_to_rgba(uint2 h) {
    return make_float4(__half2float(__ushort_as_half((unsigned short)(h.x & 0xFFFFu))),
                       __half2float(__ushort_as_half((unsigned short)(h.x >> 16))),
                       __half2float(__ushort_as_half((unsigned short)(h.y & 0xFFFFu))),
                       __half2float(__ushort_as_half((unsigned short)(h.y >> 16))));
}
__global__ __launch_bounds__(kBlock) void pack_half_kernel(const float4 *__restrict__ src, uint2 *__restrict__ dst,
                                                          size_t n_px) {
    for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < n_px; k += (size_t)gridDim.x * kBlock)
        dst[k] = rgba_to_half4(src[k]);
}
__global__ __launch_bounds__(kBlock) void widen_half_kernel(const uint2 *__restrict__ src, float4 *__restrict__ dst,
                                                           size_t n_px) {
    for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < n_px; k += (size_t)gridDim.x * kBlock)
        dst[k] = half4_to_rgba(src[k]);
}
__global__ __launch_bounds__(kBlock) void unpack_tiles_half_kernel(FrameGeom G, const uint2 *__restrict__ packed,
                                                                  float4 *__restrict__ image) {
    const size_t total = (size_t)G.n_tiles_local * 4096u;
    for (size_t pix = (size_t)blockIdx.x * kBlock + threadIdx.x; pix < total; pix += (size_t)gridDim.x * kBlock) {
        const uint32_t tile_local = (uint32_t)(pix >> 12);
        const uint32_t within = (uint32_t)(pix & 4095u);
        const uint32_t px = within & 63u, py = within >> 6;
        const uint32_t tile = tile_local * G.tile_world + G.tile_rank;
        const uint32_t X = (tile % G.tiles_x) * 64u + px, Y = (tile / G.tiles_x) * 64u + py;
        if (X < G.width && Y < G.height) image[(size_t)Y * G.width + X] = half4_to_rgba(packed[pix]);
    }
}

// ---------------------------------------------------------------------------
// Planck (T x g) LUT: one thread per texel, 201-sample CIE integration
// ---------------------------------------------------------------------------
__device__ __forceinline__ double cie_lobe_dev(double l_nm, double mean, double sd) {
    const double x = (l_nm - mean) / sd;
    return exp_rs(-0.5 * x * x);
}

__global__ __launch_bounds__(kBlock) void spectrum_lut_kernel(float4 *__restrict__ out,
                                                              uint32_t width, uint32_t height,
                                                              double max_temp) {
    const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= width * height) return;
    const uint32_t x = idx % width, y = idx / width;
    const uint32_t hden = (height > 1) ? height - 1 : 1;
    const uint32_t wden = (width > 1) ? width - 1 : 1;
    const double g = 0.05 + (5.0 - 0.05) * ((double)y / (double)hden);
    const double t = pow_rs((double)x / (double)wden, 2.5) * max_temp;
    const double t_eff = t * g;

    constexpr double SI_C = 299792458.0, SI_KB = 1.380649e-23, HP = 6.62607015e-34;
    constexpr double C1 = 2.0 * HP * SI_C * SI_C;
    constexpr double C2 = HP * SI_C / SI_KB;

    double X = 0.0, Y = 0.0, Z = 0.0;
    if (!(t_eff < 100.0)) {
        double lambda = 380.0e-9;
        const double end = 780.0e-9, step = 2.0e-9;
        while (lambda <= end) {
            const double exponent = C2 / (lambda * t_eff);
            double intensity = 0.0;
            if (!(exponent > 100.0)) {
                const double l2 = lambda * lambda;
                const double l5 = lambda * (l2 * l2); // powi(5) = a * (a^2)^2
                intensity = (C1 / l5) / (exp_rs(exponent) - 1.0);
            }
            const double l_nm = lambda * 1e9;
            const double cx = fmax(1.056 * cie_lobe_dev(l_nm, 599.0, 37.9) +
                                       0.362 * cie_lobe_dev(l_nm, 442.0, 16.0) -
                                       0.065 * cie_lobe_dev(l_nm, 501.0, 20.4),
                                   0.0);
            const double cy = fmax(0.821 * cie_lobe_dev(l_nm, 568.0, 46.9) +
                                       0.286 * cie_lobe_dev(l_nm, 530.0, 22.1),
                                   0.0);
            const double cz = fmax(1.217 * cie_lobe_dev(l_nm, 437.0, 11.8) +
                                       0.681 * cie_lobe_dev(l_nm, 459.0, 26.0),
                                   0.0);
            X += intensity * cx * step;
            Y += intensity * cy * step;
            Z += intensity * cz * step;
            lambda += step;
        }
    }
    const double r = 3.2404542 * X - 1.5371385 * Y - 0.4985314 * Z;
    const double gg = -0.9692660 * X + 1.8760108 * Y + 0.0415560 * Z;
    const double b = 0.0556434 * X - 0.2040259 * Y + 1.0572252 * Z;
    const double g2 = g * g;
    const double g4 = g2 * g2;
    const float scale = (float)(1.0e-14 * g4);
    out[idx] = make_float4((float)fmax(r, 0.0) * scale, (float)fmax(gg, 0.0) * scale,
                           (float)fmax(b, 0.0) * scale, 1.0f);
}

} // namespace
