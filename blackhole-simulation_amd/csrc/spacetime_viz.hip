// spacetime_viz.hip -- see spacetime_viz.hpp.  Compiled with -ffp-contract=off so the
// closed forms keep the reference's operation order on host and device alike.
#include "spacetime_viz.hpp"
#include "strict_libm.hpp"

#include <cmath>

namespace grvhip {

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kHalfPi = 1.57079632679489661923;

// the five non-zero covariant Boyer-Lindquist components (metric/kerr.rs:241-264)
struct CovBL {
    double tt, rr, thth, phph, tph;
};
__host__ __device__ inline CovBL covariant_bl(const VizHole &bh, double r, double theta) {
    const double m = bh.mass, a = bh.a_bl;
    const double rr = r * r, aa = a * a;
    const double st = strictm::sl_sin(theta), ct = strictm::sl_cos(theta);
    const double s2 = st * st, c2 = ct * ct;
    const double sigma = rr + aa * c2;
    const double delta = rr - 2.0 * m * r + aa;
    CovBL g;
    g.tt = -(1.0 - (2.0 * m * r) / sigma);
    g.rr = sigma / delta;
    g.thth = sigma;
    g.phph = (rr + aa + (2.0 * m * r * aa * s2) / sigma) * s2;
    g.tph = -(2.0 * m * r * a * s2) / sigma;
    return g;
}

__host__ __device__ inline double kretschner_at(const VizHole &bh, double r, double theta) {
    // raw spin: the reference hands (mass, spin) to a free function, no Kerr::new clamp
    const double a = bh.spin_raw * bh.mass;
    const double r2 = r * r, a2 = a * a;
    const double c = strictm::sl_cos(theta);
    const double c2 = c * c, c4 = c2 * c2, c6 = c4 * c2;
    const double r4 = r2 * r2, r6 = r4 * r2;
    const double a4 = a2 * a2, a6 = a4 * a2;
    const double sigma = r2 + a2 * c2;
    const double sg2 = sigma * sigma;
    const double sigma6 = sg2 * (sg2 * sg2); // powi(6): x^2 * (x^2)^2
    if (sigma6 < 1e-30) return INFINITY;
    const double num = r6 - 15.0 * r4 * a2 * c2 + 15.0 * r2 * a4 * c4 - a6 * c6;
    return 48.0 * bh.mass * bh.mass * num / sigma6;
}

__host__ __device__ inline double tilt_at(const VizHole &bh, double r, double theta) {
    const CovBL g = covariant_bl(bh, r, theta);
    // Boyer-Lindquist: g_tr == 0, so only the diagonal branch of lightcone.rs:26-33 is live
    if (g.tt >= 0.0) return kHalfPi;
    const double ratio = fmax(-g.tt / g.rr, 0.0);
    return strictm::sl_atan(sqrt(ratio));
}

__host__ __device__ inline double omega_at(const VizHole &bh, double r, double theta) {
    const CovBL g = covariant_bl(bh, r, theta);
    return fabs(g.phph) < 1e-30 ? 0.0 : -g.tph / g.phph;
}

__host__ __device__ inline double ergosphere_at(const VizHole &bh, double theta) {
    const double c = strictm::sl_cos(theta);
    const double disc = bh.mass * bh.mass - bh.a_bl * bh.a_bl * c * c;
    return disc < 0.0 ? bh.mass : bh.mass + sqrt(disc);
}

__host__ __device__ inline double flamm_at(double r, double mass) {
    const double rs = 2.0 * mass;
    return r <= rs ? 0.0 : 2.0 * sqrt(rs * (r - rs));
}

// midpoint rule over the equatorial g_rr; EMBED: sqrt|g_rr - 1| (embedding.rs:31-46),
// else sqrt|g_rr| (embedding.rs:51-65)
template <bool EMBED>
__host__ __device__ inline double radial_midpoint_sum(const VizHole &bh, double r_from, double r_to,
                                                      size_t n_steps) {
    const double dr = (r_to - r_from) / (double)n_steps;
    double acc = 0.0;
    for (size_t i = 0; i < n_steps; ++i) {
        const double ri = r_from + ((double)i + 0.5) * dr;
        const double grr = covariant_bl(bh, ri, kHalfPi).rr;
        acc += sqrt(fabs(EMBED ? grr - 1.0 : grr)) * dr;
    }
    return acc;
}

__global__ __launch_bounds__(256) void viz_field_kernel(int field, VizHole bh, double r_min,
                                                        double r_max, uint32_t n_radial,
                                                        uint32_t n_polar, float *__restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_radial * n_polar) return;
    const uint32_t i = k / n_polar, j = k % n_polar;
    const double r = r_min + (r_max - r_min) * (double)i / (double)(n_radial - 1u);
    const double theta = 0.1 + (kPi - 0.2) * (double)j / (double)(n_polar - 1u);
    double v;
    if (field == kVizKretschner) v = kretschner_at(bh, r, theta);
    else if (field == kVizLightConeTilt) v = tilt_at(bh, r, theta);
    else v = omega_at(bh, r, theta);
    out[3 * k + 0] = (float)r;
    out[3 * k + 1] = (float)theta;
    out[3 * k + 2] = (float)v;
}

__global__ __launch_bounds__(256) void embedding_mesh_kernel(VizHole bh, double r_min, double r_max,
                                                             uint32_t n_radial, uint32_t n_angular,
                                                             float *__restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_radial * n_angular) return;
    const uint32_t i = k / n_angular, j = k % n_angular;
    const double t = (double)i / (double)(n_radial - 1u);
    const double r = r_min + t * (r_max - r_min);
    const double height = fabs(bh.spin_raw) < 1e-6 ? flamm_at(r, bh.mass)
                                                   : radial_midpoint_sum<true>(bh, r, r_max, 100);
    const double phi = 2.0 * kPi * (double)j / (double)n_angular;
    out[3 * k + 0] = (float)(r * strictm::sl_cos(phi));
    out[3 * k + 1] = (float)(-height);
    out[3 * k + 2] = (float)(r * strictm::sl_sin(phi));
}

__global__ __launch_bounds__(256) void ergosphere_mesh_kernel(VizHole bh, uint32_t n_polar,
                                                              uint32_t n_azimuthal,
                                                              float *__restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_polar * n_azimuthal) return;
    const uint32_t i = k / n_azimuthal, j = k % n_azimuthal;
    const double theta = kPi * (double)i / (double)(n_polar - 1u);
    const double re = ergosphere_at(bh, theta);
    const double phi = 2.0 * kPi * (double)j / (double)n_azimuthal;
    out[3 * k + 0] = (float)(re * strictm::sl_sin(theta) * strictm::sl_cos(phi));
    out[3 * k + 1] = (float)(re * strictm::sl_cos(theta));
    out[3 * k + 2] = (float)(re * strictm::sl_sin(theta) * strictm::sl_sin(phi));
}

inline dim3 grid_for(uint32_t n) { return dim3((n + 255u) / 256u); }

} // namespace

double viz_kretschner(const VizHole &bh, double r, double theta) { return kretschner_at(bh, r, theta); }
double viz_light_cone_tilt(const VizHole &bh, double r, double theta) { return tilt_at(bh, r, theta); }
double viz_frame_drag_omega(const VizHole &bh, double r, double theta) { return omega_at(bh, r, theta); }
double viz_flamm_height(double r, double mass) { return flamm_at(r, mass); }
double viz_proper_distance(const VizHole &bh, double r1, double r2, size_t n_steps) {
    return r1 < r2 ? radial_midpoint_sum<false>(bh, r1, r2, n_steps)
                   : radial_midpoint_sum<false>(bh, r2, r1, n_steps);
}

hipError_t launch_viz_field(int field, const VizHole &bh, double r_min, double r_max,
                            uint32_t n_radial, uint32_t n_polar, float *d_out, hipStream_t s) {
    const uint32_t n = n_radial * n_polar;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(viz_field_kernel, grid_for(n), dim3(256), 0, s, field, bh, r_min, r_max,
                       n_radial, n_polar, d_out);
    return hipGetLastError();
}

hipError_t launch_embedding_mesh(const VizHole &bh, double r_min, double r_max, uint32_t n_radial,
                                 uint32_t n_angular, float *d_out, hipStream_t s) {
    const uint32_t n = n_radial * n_angular;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(embedding_mesh_kernel, grid_for(n), dim3(256), 0, s, bh, r_min, r_max,
                       n_radial, n_angular, d_out);
    return hipGetLastError();
}

hipError_t launch_ergosphere_mesh(const VizHole &bh, uint32_t n_polar, uint32_t n_azimuthal,
                                  float *d_out, hipStream_t s) {
    const uint32_t n = n_polar * n_azimuthal;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ergosphere_mesh_kernel, grid_for(n), dim3(256), 0, s, bh, n_polar,
                       n_azimuthal, d_out);
    return hipGetLastError();
}

} // namespace grvhip
