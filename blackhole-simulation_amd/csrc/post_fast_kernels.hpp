// post_fast_kernels.hpp -- FAST-contract forms of the post chain (included by kernels_fast.hip only).
//
// The FAST contract of the post kernels (post_kernels.hpp) is: the shaders' equations with FMA
// contraction, reciprocal-based divide / sqrt, v_log / v_exp gamma -- and a tap that sits exactly on
// a texel centre, or exactly half-way between texels, is the texel (or the plain average of the two /
// four texels) instead of an f32 bilinear evaluation whose weights come out as 1 - 1e-7 / 0.5 + 1e-7.
// That leaves every kernel free to pick its own tiling and summation order; these are the forms the
// FAST launchers use (the STRICT launchers keep the shader-order kernels of post_kernels.hpp):
//   taa / ataa resolve : 64 x 16 output tile per 256-thread block, the current frame's 66 x 18 window
//                        staged in LDS once (1.16 fetches per pixel; the 64 x 4 tile of the shader-order
//                        kernel needs 2.1), one thread resolves four consecutive rows of its column
//                        with the 3 x 3 moments built from shared row sums (18 LDS reads per 4 pixels
//                        instead of 36);
//   bloom              : three launches as in post_kernels.hpp, each with one stage less:
//                        bright + H blur reads its 4 x 4 scene block per quarter-size texel straight
//                        into registers (16 B/px of scene traffic, no LDS bright window),
//                        (V, H) pairs and V + combine blur straight out of global memory (texel reads
//                        on a quarter-size image that lives in L2) into one LDS stage.
// Tolerances against the oracle: tests/test_post_chain.py::test_fast_post_chain_matches_oracle.
#pragma once

#include "post_kernels.hpp"

namespace {

constexpr int kFtW = 64, kFtH = 16, kFtRows = 4; // output tile, rows per thread (256 threads = 64 x 4)
struct FastTile {
    float4 t[kFtH + 2][kFtW + 2]; // t[0][0] is image texel (blockIdx.x * 64 - 1, blockIdx.y * 16 - 1), edge-clamped
};
__device__ __forceinline__ void fast_tile_load(FastTile &T, const float4 *__restrict__ img, uint32_t w, uint32_t h) {
    const int tid = threadIdx.y * kFtW + threadIdx.x;
    const int bx0 = (int)(blockIdx.x * kFtW) - 1, by0 = (int)(blockIdx.y * kFtH) - 1;
    for (int k = tid; k < (kFtH + 2) * (kFtW + 2); k += kFtW * kFtRows) {
        const int ly = k / (kFtW + 2), lx = k - ly * (kFtW + 2);
        const int gx = post_clampi(bx0 + lx, 0, (int)w - 1), gy = post_clampi(by0 + ly, 0, (int)h - 1);
        T.t[ly][lx] = img[(size_t)gy * w + gx];
    }
    __syncthreads();
}
// mean and standard deviation of the 3 x 3 YCoCg neighbourhoods of the thread's four pixels
// (column threadIdx.x, tile rows 4 threadIdx.y ... + 3), and the four centre texels
struct FastNeighbourhood {
    float4 centre[kFtRows];
    float mean[kFtRows][3], sd[kFtRows][3];
};
__device__ __forceinline__ void fast_neighbourhood(const FastTile &T, FastNeighbourhood &N) {
    const int lx = (int)threadIdx.x + 1, r0 = (int)threadIdx.y * kFtRows;
    float s1[kFtRows + 2][3], s2[kFtRows + 2][3];
#pragma unroll
    for (int j = 0; j < kFtRows + 2; ++j) {
        const float4 a = T.t[r0 + j][lx - 1], b = T.t[r0 + j][lx], c = T.t[r0 + j][lx + 1];
        const YCC ya = to_ycocg(a.x, a.y, a.z), yb = to_ycocg(b.x, b.y, b.z), yc = to_ycocg(c.x, c.y, c.z);
        s1[j][0] = ya.y + yb.y + yc.y;
        s1[j][1] = ya.co + yb.co + yc.co;
        s1[j][2] = ya.cg + yb.cg + yc.cg;
        s2[j][0] = ya.y * ya.y + yb.y * yb.y + yc.y * yc.y;
        s2[j][1] = ya.co * ya.co + yb.co * yb.co + yc.co * yc.co;
        s2[j][2] = ya.cg * ya.cg + yb.cg * yb.cg + yc.cg * yc.cg;
        if (j >= 1 && j <= kFtRows) N.centre[j - 1] = b;
    }
#pragma unroll
    for (int i = 0; i < kFtRows; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float m = (s1[i][c] + s1[i + 1][c] + s1[i + 2][c]) * (1.0f / 9.0f);
            N.mean[i][c] = m;
            N.sd[i][c] = sqrtf(fmaxf((s2[i][c] + s2[i + 1][c] + s2[i + 2][c]) * (1.0f / 9.0f) - m * m, 0.0f));
        }
}

// reprojection.glsl.ts:44-116 (taa_resolve_kernel of post_kernels.hpp), FAST contract
__global__ __launch_bounds__(256) void taa_resolve_fast_kernel(uint32_t w, uint32_t h,
                                                               const float4 *__restrict__ current,
                                                               const float4 *__restrict__ history,
                                                               float blend_factor, int camera_moving,
                                                               int half_storage, float4 *__restrict__ out) {
    __shared__ FastTile tile;
    fast_tile_load(tile, current, w, h);
    const uint32_t px = blockIdx.x * kFtW + threadIdx.x;
    if (px >= w) return;
    FastNeighbourhood N;
    fast_neighbourhood(tile, N);
#pragma unroll
    for (int i = 0; i < kFtRows; ++i) {
        const uint32_t py = blockIdx.y * kFtH + threadIdx.y * kFtRows + (uint32_t)i;
        if (py >= h) break;
        const float4 h4 = history[(size_t)py * w + px]; // the history tap sits on the pixel centre
        YCC hy = to_ycocg(h4.x, h4.y, h4.z);
        hy.y = post_clamp(hy.y, N.mean[i][0] - 1.5f * N.sd[i][0], N.mean[i][0] + 1.5f * N.sd[i][0]);
        hy.co = post_clamp(hy.co, N.mean[i][1] - 1.5f * N.sd[i][1], N.mean[i][1] + 1.5f * N.sd[i][1]);
        hy.cg = post_clamp(hy.cg, N.mean[i][2] - 1.5f * N.sd[i][2], N.mean[i][2] + 1.5f * N.sd[i][2]);
        const float varianceWeight = 1.0f - post_clamp(N.sd[i][0] * 4.0f, 0.0f, 0.55f);
        const float alpha = camera_moving ? 0.0f : blend_factor * varianceWeight;
        const YCC cy = to_ycocg(N.centre[i].x, N.centre[i].y, N.centre[i].z);
        out[(size_t)py * w + px] = from_ycocg(
            YCC{post_mix(cy.y, hy.y, alpha), post_mix(cy.co, hy.co, alpha), post_mix(cy.cg, hy.cg, alpha)},
            half_storage);
    }
}

// ataa.wgsl.ts:29-86 (ataa_resolve_kernel of post_kernels.hpp), FAST contract.
// The reprojection chain of the shader -- ndc -> inv_proj -> perspective divide -> normalise ->
// inv_view -> position + 12 dir -> prev_view_proj -> perspective divide -> uv -> texel coordinates --
// is three 4x4 products, two divisions and a square root per pixel (about 75 VALU instructions, more
// than the resolve itself).  Everything in it except the normalisation is linear in the pixel
// coordinates, so the launcher folds the matrices once, in f64 (ataa_reproj_fold):
//   vt      = inv_proj (ndc.x, -ndc.y, 1, 1)                          = Vx px + Vy py + V0
//   clip    = K + s (B (ndc.x, -ndc.y, 1, 1)),  s = 12 sign(vt.w) / |vt.xyz|
//             K = prev_view_proj (position, 1),  B = prev_view_proj[:, :3] inv_view[:3, :3] inv_proj[:3, :]
//   texel x = (0.5 w clip.x + (0.5 w - 0.5) clip.w) / clip.w   (row combination folded into K and B)
// which leaves seven fma per pixel row, a dot product, one v_rsq_f32, one v_rcp_f32 and five more
// multiply-adds (19 instructions).  The tap position agrees with the shader-order chain to the
// latter's own f32 rounding (the fold is exact to f64): tests/test_post_chain.py holds it to the same
// tap-position bound as before.
struct AtaaReproj {
    // per quantity q in (vt.x, vt.y, vt.z, vt.w, clipx', clipy', clipw): q = c[q][0] px + c[q][1] py + c[q][2]
    // (clip rows: the part multiplied by s), k = the constant part of the three clip rows
    float c[7][3];
    float k[3];
};

// bilinear LINEAR + CLAMP_TO_EDGE tap at texel coordinates (x, y) = (u w - 0.5, v h - 0.5)
__device__ __forceinline__ float4 fast_sample_xy(const float4 *__restrict__ tex, uint32_t w, uint32_t h, float x,
                                                 float y) {
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    const int i0 = post_clampi((int)fx, 0, (int)w - 1), i1 = post_clampi((int)fx + 1, 0, (int)w - 1);
    const int j0 = post_clampi((int)fy, 0, (int)h - 1), j1 = post_clampi((int)fy + 1, 0, (int)h - 1);
    const float4 t00 = tex[(size_t)j0 * w + i0], t10 = tex[(size_t)j0 * w + i1];
    const float4 t01 = tex[(size_t)j1 * w + i0], t11 = tex[(size_t)j1 * w + i1];
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return make_float4(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x,
                       w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
                       w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z,
                       w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w);
}

__global__ __launch_bounds__(256) void ataa_resolve_fast_kernel(uint32_t w, uint32_t h, AtaaReproj rp,
                                                                const float4 *__restrict__ current,
                                                                const float4 *__restrict__ history,
                                                                int half_storage, float4 *__restrict__ out) {
    __shared__ FastTile tile;
    fast_tile_load(tile, current, w, h);
    const uint32_t px = blockIdx.x * kFtW + threadIdx.x;
    if (px >= w) return;
    FastNeighbourhood N;
    fast_neighbourhood(tile, N);
    float col[7]; // the part of each quantity that depends on the column only
#pragma unroll
    for (int q = 0; q < 7; ++q) col[q] = fmaf((float)px, rp.c[q][0], rp.c[q][2]);
#pragma unroll
    for (int i = 0; i < kFtRows; ++i) {
        const uint32_t py = blockIdx.y * kFtH + threadIdx.y * kFtRows + (uint32_t)i;
        if (py >= h) break;
        float q[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) q[k] = fmaf((float)py, rp.c[k][1], col[k]);
        const float s = copysignf(12.0f * __builtin_amdgcn_rsqf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), q[3]);
        const float ipw = __builtin_amdgcn_rcpf(fmaf(s, q[6], rp.k[2]));
        const float4 h4 = fast_sample_xy(history, w, h, fmaf(s, q[4], rp.k[0]) * ipw, fmaf(s, q[5], rp.k[1]) * ipw);
        YCC hy = to_ycocg(h4.x, h4.y, h4.z);
        hy.y = post_clamp(hy.y, N.mean[i][0] - 2.0f * N.sd[i][0], N.mean[i][0] + 2.0f * N.sd[i][0]);
        hy.co = post_clamp(hy.co, N.mean[i][1] - 2.0f * N.sd[i][1], N.mean[i][1] + 2.0f * N.sd[i][1]);
        hy.cg = post_clamp(hy.cg, N.mean[i][2] - 2.0f * N.sd[i][2], N.mean[i][2] + 2.0f * N.sd[i][2]);
        const YCC center = to_ycocg(N.centre[i].x, N.centre[i].y, N.centre[i].z);
        out[(size_t)py * w + px] = from_ycocg(YCC{post_mix(center.y, hy.y, 0.92f), post_mix(center.co, hy.co, 0.92f),
                                                  post_mix(center.cg, hy.cg, 0.92f)},
                                              half_storage);
    }
}

// ---- bloom, nested sizes (w = 4 bw, h = 4 bh, hw = 2 bw, hh = 2 bh) ----------------------------
#define GRV_BLUR_WEIGHTS const float kBlurW[5] = {0.227027f, 0.1945946f, 0.1216216f, 0.054054f, 0.016216f}
constexpr int kBlurOutW = kTileW - 8; // a 64-wide stage with 4 texels of halo either side feeds 56 outputs

// one texel of the half-size bright target (bloom_bright_kernel): the tap sits on the corner shared by
// scene texels (2i, 2j) ... (2i + 1, 2j + 1)
__device__ __forceinline__ float4 fast_bright(const float4 a, const float4 b, const float4 c, const float4 d,
                                              float threshold, int half_storage) {
    const float4 s = make_float4(0.25f * (a.x + b.x + c.x + d.x), 0.25f * (a.y + b.y + c.y + d.y),
                                 0.25f * (a.z + b.z + c.z + d.z), 0.25f * (a.w + b.w + c.w + d.w));
    const float lum = s.x * 0.299f + s.y * 0.587f + s.z * 0.114f;
    return lum > threshold ? make_float4(post_store(s.x, half_storage), post_store(s.y, half_storage),
                                         post_store(s.z, half_storage), post_store(s.w, half_storage))
                           : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
// nine taps along x out of a 64-wide LDS row stage; lx = the centre's column in the stage
__device__ __forceinline__ float4 fast_blur9_row(const float4 *row, int lx, int half_storage) {
    GRV_BLUR_WEIGHTS;
    const float4 c = row[lx];
    float r = c.x * kBlurW[0], g = c.y * kBlurW[0], b = c.z * kBlurW[0];
#pragma unroll
    for (int i = 1; i < 5; ++i) {
        const float4 p = row[lx + i], m = row[lx - i];
        r += (p.x + m.x) * kBlurW[i];
        g += (p.y + m.y) * kBlurW[i];
        b += (p.z + m.z) * kBlurW[i];
    }
    return make_float4(post_store(r, half_storage), post_store(g, half_storage), post_store(b, half_storage), 1.0f);
}
// nine taps along y of a bw x bh image in global memory (rows clamp to the edge)
__device__ __forceinline__ float4 fast_blur9_col(const float4 *__restrict__ src, uint32_t bw, uint32_t bh, int x, int y,
                                                 int half_storage) {
    GRV_BLUR_WEIGHTS;
    const float4 c = src[(size_t)y * bw + x];
    float r = c.x * kBlurW[0], g = c.y * kBlurW[0], b = c.z * kBlurW[0];
#pragma unroll
    for (int i = 1; i < 5; ++i) {
        const float4 p = src[(size_t)post_clampi(y + i, 0, (int)bh - 1) * bw + x];
        const float4 m = src[(size_t)post_clampi(y - i, 0, (int)bh - 1) * bw + x];
        r += (p.x + m.x) * kBlurW[i];
        g += (p.y + m.y) * kBlurW[i];
        b += (p.z + m.z) * kBlurW[i];
    }
    return make_float4(post_store(r, half_storage), post_store(g, half_storage), post_store(b, half_storage), 1.0f);
}

// bright pass + first horizontal blur.  Block = 64 x 4 threads, 56 x 4 outputs of the quarter-size
// target.  Stage: thread (tx, ty) holds column x = 56 bx - 4 + tx: the H blur's tap on the half-size
// bright target sits on the corner of bright texels (2x, 2y) ... (2x + 1, 2y + 1) (columns clamped to
// the edge), each of which is the thresholded corner average of a 2 x 2 scene block: a 4 x 4 scene
// block per thread, read once, as four 64-byte runs.
__global__ __launch_bounds__(256) void bloom_bright_hblur_fast_kernel(uint32_t w, const float4 *__restrict__ scene,
                                                                      uint32_t bw, uint32_t bh, float threshold,
                                                                      int half_storage, float4 *__restrict__ dst) {
    __shared__ float4 q_s[kTileH][kTileW];
    const int x = (int)(blockIdx.x * kBlurOutW) - 4 + (int)threadIdx.x;
    const uint32_t y = blockIdx.y * kTileH + threadIdx.y;
    if (y < bh) {
        const int hw = 2 * (int)bw;
        const int i0 = post_clampi(2 * x, 0, hw - 1), i1 = post_clampi(2 * x + 1, 0, hw - 1);
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int j = 0; j < 2; ++j) { // bright rows 2y, 2y + 1 = scene rows 4y + 2j, 4y + 2j + 1
            const float4 *r0 = scene + (size_t)(4u * y + 2u * (uint32_t)j) * w, *r1 = r0 + w;
            const float4 b0 = fast_bright(r0[2 * i0], r0[2 * i0 + 1], r1[2 * i0], r1[2 * i0 + 1], threshold, half_storage);
            const float4 b1 = fast_bright(r0[2 * i1], r0[2 * i1 + 1], r1[2 * i1], r1[2 * i1 + 1], threshold, half_storage);
            acc.x += b0.x + b1.x;
            acc.y += b0.y + b1.y;
            acc.z += b0.z + b1.z;
        }
        q_s[threadIdx.y][threadIdx.x] = make_float4(0.25f * acc.x, 0.25f * acc.y, 0.25f * acc.z, 0.0f);
    }
    __syncthreads();
    if (threadIdx.x < 4 || threadIdx.x >= 4 + kBlurOutW || x >= (int)bw || y >= bh) return;
    dst[(size_t)y * bw + x] = fast_blur9_row(q_s[threadIdx.y], (int)threadIdx.x, half_storage);
}

// a middle (V, H) pair: thread (tx, ty) blurs column x = 56 bx - 4 + tx (clamped to the edge, as the H
// taps clamp) vertically straight out of global memory into the stage, 56 x 4 threads blur the stage
// horizontally.
__global__ __launch_bounds__(256) void bloom_vh_blur_fast_kernel(uint32_t bw, uint32_t bh,
                                                                 const float4 *__restrict__ src, int half_storage,
                                                                 float4 *__restrict__ dst) {
    __shared__ float4 v_s[kTileH][kTileW];
    const int x = (int)(blockIdx.x * kBlurOutW) - 4 + (int)threadIdx.x;
    const uint32_t y = blockIdx.y * kTileH + threadIdx.y;
    if (y < bh)
        v_s[threadIdx.y][threadIdx.x] = fast_blur9_col(src, bw, bh, post_clampi(x, 0, (int)bw - 1), (int)y, half_storage);
    __syncthreads();
    if (threadIdx.x < 4 || threadIdx.x >= 4 + kBlurOutW || x >= (int)bw || y >= bh) return;
    dst[(size_t)y * bw + x] = fast_blur9_row(v_s[threadIdx.y], (int)threadIdx.x, half_storage);
}

// last vertical blur + combine.  Block = 64 x 16 output pixels; their bilinear taps on the quarter-size
// bloom touch texels qx0 - 1 ... qx0 + 16, qy0 - 1 ... qy0 + 4 (18 x 6, clamped to the edge), each a
// vertical blur read straight out of global memory by one of 108 threads.
constexpr int kFcW = kTileW / 4 + 2, kFcH = 4 + 2;
__global__ __launch_bounds__(256) void bloom_vblur_combine_fast_kernel(uint32_t w, uint32_t h,
                                                                       const float4 *__restrict__ scene, uint32_t bw,
                                                                       uint32_t bh, const float4 *__restrict__ hblur,
                                                                       float intensity, int half_storage,
                                                                       float4 *__restrict__ out) {
    __shared__ float4 vb_s[kFcH][kFcW];
    const int tid = threadIdx.y * kTileW + threadIdx.x;
    const int qx0 = (int)(blockIdx.x * (kTileW / 4)) - 1, qy0 = (int)(blockIdx.y * 4) - 1;
    if (tid < kFcW * kFcH) {
        const int ly = tid / kFcW, lx = tid - ly * kFcW;
        vb_s[ly][lx] = fast_blur9_col(hblur, bw, bh, post_clampi(qx0 + lx, 0, (int)bw - 1),
                                      post_clampi(qy0 + ly, 0, (int)bh - 1), half_storage);
    }
    __syncthreads();
    const uint32_t px = blockIdx.x * kTileW + threadIdx.x;
    if (px >= w) return;
    // pixel px sits at quarter-size coordinate (px + 0.5) / 4 - 0.5: texels (px + 2) / 4 - 1 and + 1 with
    // weight ((px + 2) mod 4) / 4 + 1 / 8 on the second (exact in f32)
    const int cx = (int)((px + 2u) >> 2) - 1 - qx0;
    const float ax = (float)((px + 2u) & 3u) * 0.25f + 0.125f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t py = blockIdx.y * 16u + (uint32_t)r * 4u + threadIdx.y;
        if (py >= h) continue;
        const int cy = (int)((py + 2u) >> 2) - 1 - qy0;
        const float ay = (float)((py + 2u) & 3u) * 0.25f + 0.125f;
        const float4 t00 = vb_s[cy][cx], t10 = vb_s[cy][cx + 1], t01 = vb_s[cy + 1][cx], t11 = vb_s[cy + 1][cx + 1];
        const float w00 = (1.0f - ax) * (1.0f - ay), w10 = ax * (1.0f - ay), w01 = (1.0f - ax) * ay, w11 = ax * ay;
        const float bx_ = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
        const float by_ = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
        const float bz_ = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
        const float4 s = scene[(size_t)py * w + px];
        out[(size_t)py * w + px] =
            make_float4(post_pow<GRV_ARITH_FAST>(post_aces(s.x + bx_ * intensity), 0.4545f),
                        post_pow<GRV_ARITH_FAST>(post_aces(s.y + by_ * intensity), 0.4545f),
                        post_pow<GRV_ARITH_FAST>(post_aces(s.z + bz_ * intensity), 0.4545f), 1.0f);
    }
}

} // namespace
