// control_plane.hip -- see control_plane.hpp.  Compiled with -ffp-contract=off so the
// scalar formulas keep the reference's operation order.
#include "control_plane.hpp"
#include "strict_libm.hpp"

#include <cmath>

namespace grvhip {

namespace {

// ---- circular equatorial orbits, host + device (disk.rs:24-64) ----
__host__ __device__ inline double orbit_energy(double r, double m, double a) {
    const double rm = r / m, sq = sqrt(m / r), am = a / m;
    const double num = 1.0 - 2.0 / rm + am * sq;
    const double den = 1.0 - 3.0 / rm + 2.0 * am * sq;
    return den <= 0.0 ? 1.0 : num / sqrt(den);
}
__host__ __device__ inline double orbit_lz(double r, double m, double a) {
    const double rm = r / m, sq = sqrt(m / r), am = a / m;
    const double ar = a / r;
    const double num = sqrt(m) * sqrt(r) * (1.0 - 2.0 * am * sq + ar * ar);
    const double den = 1.0 - 3.0 / rm + 2.0 * am * sq;
    return den <= 0.0 ? 0.0 : num / sqrt(den);
}
__host__ __device__ inline double orbit_omega(double r, double m, double a) {
    return sqrt(m) / (strictm::sl_pow(r, 1.5) + a * sqrt(m));
}
__host__ __device__ inline double torque_integrand(double rp, double m, double a) {
    const double drp = rp * 1e-5;
    const double dlz = (orbit_lz(rp + drp, m, a) - orbit_lz(rp - drp, m, a)) / (2.0 * drp);
    return (orbit_energy(rp, m, a) - orbit_omega(rp, m, a) * orbit_lz(rp, m, a)) * dlz;
}
// metric/kerr.rs:100-123, prograde
__host__ __device__ inline double isco_pro(double m, double a_star) {
    if (fabs(a_star) < 1e-6) return m * 6.0;
    const double a2 = a_star * a_star;
    const double z1 = 1.0 + strictm::sl_pow(1.0 - a2, 1.0 / 3.0) *
                               (strictm::sl_pow(1.0 + a_star, 1.0 / 3.0) + strictm::sl_pow(1.0 - a_star, 1.0 / 3.0));
    const double z2 = sqrt(3.0 * a2 + z1 * z1);
    const double disc = (3.0 - z1) * (3.0 + z1 + 2.0 * z2);
    return m * (3.0 + z2 - (disc < 0.0 ? 0.0 : sqrt(disc)));
}
// disk.rs:90-151: composite Simpson, 200 panels, numerical dOmega/dr
__host__ __device__ inline double pt_flux(double r, double m, double a_star, double m_dot) {
    const double a = a_star * m;
    const double r_isco = isco_pro(m, a_star);
    if (r <= r_isco) return 0.0;
    const double denom = orbit_energy(r, m, a) - orbit_omega(r, m, a) * orbit_lz(r, m, a);
    if (fabs(denom) < 1e-30) return 0.0;
    const double dr = r * 1e-5;
    const double omega_dr = (orbit_omega(r + dr, m, a) - orbit_omega(r - dr, m, a)) / (2.0 * dr);
    const int n = 200;
    const double h = (r - r_isco) / (double)n;
    if (h <= 0.0) return 0.0;
    double sum = torque_integrand(r_isco, m, a) + torque_integrand(r, m, a);
    for (int i = 1; i < n; ++i) {
        const double w = (i % 2 == 0) ? 2.0 : 4.0;
        sum += w * torque_integrand(r_isco + (double)i * h, m, a);
    }
    const double integral = sum * h / 3.0;
    return fabs(-(omega_dr / (denom * denom)) * integral) * m_dot;
}
// disk.rs:160-170
__host__ __device__ inline double pt_temperature(double r, double m, double a_star, double m_dot) {
    const double flux = pt_flux(r, m, a_star, m_dot);
    if (flux <= 0.0) return 0.0;
    return 1e7 * strictm::sl_pow(m_dot, 0.25) * strictm::sl_pow(flux, 0.25);
}

// one block: thread i evaluates entry i (strided), block max, normalise (disk.rs:175-201)
__global__ __launch_bounds__(256) void disk_lut_kernel(float *__restrict__ out,
                                                       double *__restrict__ temps, uint32_t width,
                                                       double m, double a_star) {
    __shared__ double s_max[256];
    const double rin = isco_pro(m, a_star), rout = 50.0 * m;
    const uint32_t den = width > 1 ? width - 1 : 1;
    double mx = 0.0;
    for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) {
        const double t = (double)i / (double)den;
        const double temp = pt_temperature(rin + t * (rout - rin), m, a_star, 1.0);
        temps[i] = temp;
        mx = temp > mx ? temp : mx;
    }
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (uint32_t off = blockDim.x / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + off]);
        __syncthreads();
    }
    const double norm = s_max[0] > 0.0 ? 1.0 / s_max[0] : 1.0;
    for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) out[i] = (float)(temps[i] * norm);
}

// shadow.rs:39-59
void critical_orbit(double r, double m, double a, double &xi, double &eta) {
    const double r2 = r * r, r3 = r2 * r, a2 = a * a;
    xi = 0.0;
    eta = 0.0;
    const double d1 = a * (r - m);
    if (std::fabs(d1) < 1e-30) return;
    xi = -(r3 - 3.0 * m * r2 + a2 * r + a2 * m) / d1;
    const double d2 = a2 * (r - m) * (r - m);
    if (std::fabs(d2) < 1e-30) return;
    const double q = r - 3.0 * m;
    eta = r3 * (4.0 * m * a2 - r * (q * q)) / d2;
}

// glam 0.24.2 DQuat::from_rotation_y(angle).mul_vec3(v)
void rotate_about_y(double angle, double v[3]) {
    const double s = strictm::sl_sin(angle * 0.5), w = strictm::sl_cos(angle * 0.5);
    const double b[3] = {0.0, s, 0.0};
    const double b2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    const double dot = v[0] * b[0] + v[1] * b[1] + v[2] * b[2];
    const double c[3] = {b[1] * v[2] - b[2] * v[1], b[2] * v[0] - b[0] * v[2], b[0] * v[1] - b[1] * v[0]};
    const double k1 = w * w - b2, k2 = dot * 2.0, k3 = w * 2.0;
    for (int i = 0; i < 3; ++i) v[i] = v[i] * k1 + b[i] * k2 + c[i] * k3;
}

} // namespace

double page_thorne_flux_host(double r, double mass, double spin_clamped, double m_dot) {
    return pt_flux(r, mass, spin_clamped, m_dot);
}

hipError_t launch_disk_temperature_lut(float *d_out, double *d_scratch, uint32_t width, double mass,
                                       double spin_clamped, hipStream_t s) {
    if (width == 0) return hipSuccess;
    hipLaunchKernelGGL(disk_lut_kernel, dim3(1), dim3(256), 0, s, d_out, d_scratch, width, mass,
                       spin_clamped);
    return hipGetLastError();
}

double schwarzschild_shadow_radius_host(double mass) { return 3.0 * std::sqrt(3.0) * mass; }

std::vector<double> bardeen_shadow_host(double m, double a_star, double theta_obs, size_t n) {
    constexpr double kPi = 3.14159265358979323846;
    const double a = a_star * m;
    const double so = strictm::sl_sin(theta_obs), co = strictm::sl_cos(theta_obs);
    std::vector<double> pts;
    if (std::fabs(a) < 1e-10) { // shadow.rs:90-98
        const double radius = schwarzschild_shadow_radius_host(m);
        for (size_t i = 0; i < n; ++i) {
            const double phi = 2.0 * kPi * (double)i / (double)n;
            pts.push_back(radius * strictm::sl_cos(phi));
            pts.push_back(radius * strictm::sl_sin(phi));
        }
        return pts;
    }
    if (std::fabs(so) < 1e-10) { // shadow.rs:100-112
        const double r_ph = 2.0 * m * (1.0 + strictm::sl_cos((2.0 / 3.0) * strictm::sl_acos(-a_star)));
        double xi, eta;
        critical_orbit(r_ph, m, a, xi, eta);
        const double radius = std::sqrt(std::fmax(eta + a * a, 0.0));
        for (size_t i = 0; i < 2 * n; ++i) {
            const double phi = 2.0 * kPi * (double)i / (2.0 * (double)n);
            pts.push_back(radius * strictm::sl_cos(phi));
            pts.push_back(radius * strictm::sl_sin(phi));
        }
        return pts;
    }
    auto beta_sq = [&](double r) {
        double xi, eta;
        critical_orbit(r, m, a, xi, eta);
        return eta + a * a * co * co - xi * xi * co * co / (so * so);
    };
    const double as = a / m;
    const double r_pro = 2.0 * m * (1.0 + strictm::sl_cos((2.0 / 3.0) * strictm::sl_acos(-std::fabs(as))));
    const double r_ret = 2.0 * m * (1.0 + strictm::sl_cos((2.0 / 3.0) * strictm::sl_acos(std::fabs(as))));
    double r_min = r_pro, r_max = r_ret;
    const int steps = 1000;
    for (int i = 0; i <= steps; ++i) {
        const double r = r_pro + ((double)i / (double)steps) * (r_ret - r_pro);
        if (beta_sq(r) >= 0.0) {
            r_min = r;
            break;
        }
    }
    for (int i = steps; i >= 0; --i) {
        const double r = r_pro + ((double)i / (double)steps) * (r_ret - r_pro);
        if (beta_sq(r) >= 0.0) {
            r_max = r;
            break;
        }
    }
    const size_t den = n > 1 ? n - 1 : 1;
    auto emit = [&](size_t i, double sign) {
        const double phase = kPi * (double)i / (double)den;
        const double t = 0.5 - 0.5 * strictm::sl_cos(phase);
        const double r = r_min + t * (r_max - r_min);
        double xi, eta;
        critical_orbit(r, m, a, xi, eta);
        pts.push_back(a * so - xi / so);
        const double b = std::sqrt(std::fmax(beta_sq(r), 0.0));
        pts.push_back(sign < 0.0 ? -b : b);
    };
    for (size_t i = 0; i < n; ++i) emit(i, -1.0);
    for (size_t i = n; i-- > 0;) emit(i, 1.0);
    return pts;
}

bool CameraFilter::finite() const {
    for (int i = 0; i < 3; ++i)
        if (!std::isfinite(position[i]) || !std::isfinite(velocity[i])) return false;
    for (int i = 0; i < 4; ++i)
        if (!std::isfinite(orientation[i])) return false;
    return true;
}

void CameraFilter::update(double mouse_dx, double mouse_dy, double zoom_delta, double dt) {
    (void)mouse_dy; // camera.rs reads only dx
    if (dt <= 0.0) return;
    const double friction = strictm::sl_exp(-5.0 * dt);
    for (int i = 0; i < 3; ++i) velocity[i] *= friction;
    for (int i = 0; i < 3; ++i) position[i] += velocity[i] * dt;
    rotate_about_y(-mouse_dx * 2.0 * dt, position);
    if (auto_spin) rotate_about_y(0.15 * dt, position);
    const double zoom = 1.0 + zoom_delta * dt;
    for (int i = 0; i < 3; ++i) position[i] *= zoom;
}

void tick_sab_host(float *sab, double mass, double spin, double spin_clamped, double horizon,
                   double isco, CameraFilter &cam, CameraFilter &last_good, double dt_override) {
    constexpr int kControl = 0, kCamera = 64, kPhysics = 128, kTelemetry = 256; // lib.rs:36-40
    const double mdx = sab[kControl + 1], mdy = sab[kControl + 2], zoom = sab[kControl + 3];
    const double dt = dt_override > 0.0 ? dt_override : (double)sab[kControl + 4];
    sab[kControl + 1] = sab[kControl + 2] = sab[kControl + 3] = 0.0f;

    cam.update(mdx, mdy, zoom, dt);
    if (!cam.finite())
        cam = last_good;
    else
        last_good = cam;

    for (int i = 0; i < 3; ++i) {
        sab[kCamera + i] = (float)cam.position[i];
        sab[kCamera + 4 + i] = (float)cam.velocity[i];
    }
    for (int i = 0; i < 4; ++i) sab[kCamera + 8 + i] = (float)cam.orientation[i];

    sab[kPhysics] = (float)horizon;
    sab[kPhysics + 1] = (float)isco;
    sab[kPhysics + 2] = (float)mass;
    sab[kPhysics + 3] = (float)spin;

    const double *p = cam.position;
    const double r_cam = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    if (r_cam > 0.0) {
        const double theta_obs = strictm::sl_acos(p[1] / r_cam);
        const std::vector<double> curve = bardeen_shadow_host(mass, spin_clamped, theta_obs, 32);
        const size_t n = curve.size() / 2;
        // the reference clears 128 floats from PHYSICS+16, i.e. through TELEMETRY+15
        // (SURVEY F10): reproduced, not "fixed"
        for (int i = 0; i < 128; ++i) sab[kPhysics + 16 + i] = 0.0f;
        const size_t actual = n < 64 ? n : 64;
        sab[kPhysics + 15] = (float)actual;
        for (size_t i = 0; i < actual; ++i) {
            sab[kPhysics + 16 + 2 * i] = (float)curve[2 * i];
            sab[kPhysics + 16 + 2 * i + 1] = (float)curve[2 * i + 1];
        }
        double lo = 0.0, hi = 0.0;
        if (n > 0) {
            lo = hi = curve[0];
            for (size_t i = 0; i < n; ++i) {
                lo = curve[2 * i] < lo ? curve[2 * i] : lo;
                hi = curve[2 * i] > hi ? curve[2 * i] : hi;
            }
        }
        sab[kPhysics + 4] = (float)lo;
        sab[kPhysics + 5] = (float)hi;
    }
    sab[kTelemetry] += 1.0f;
}

} // namespace grvhip
