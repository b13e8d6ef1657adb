// engine_control.hip -- LUTs, disk / shadow helpers, the SAB protocol and the spacetime
// read-outs behind the C ABI.  See engine_internal.hpp.
#include "engine_internal.hpp"
#include "strict_libm.hpp"

using namespace grvhost;

namespace {
VizHole viz_hole(const grv_engine *e) { return VizHole{e->mass, e->spin, e->spin_c * e->mass}; }

// run one of the grid kernels into the staging buffer and copy n_floats back
template <typename Launch>
int viz_grid(grv_engine *e, size_t n_a, size_t n_b, float *out, Launch &&launch) {
    if (!e) return GRV_ERR_INVALID;
    if (n_a == 0 || n_b == 0) return GRV_OK; // empty loops upstream
    if (!out) return fail(e, GRV_ERR_INVALID, "null output");
    if (n_a > 0xFFFFu * 16u || n_b > 0xFFFFu * 16u || n_a * n_b > (size_t)1 << 28)
        return fail(e, GRV_ERR_INVALID, "grid %zu x %zu too large", n_a, n_b);
    GRV_HIP(e, hipSetDevice(e->device));
    const size_t bytes = n_a * n_b * 3 * sizeof(float);
    int rc = ensure_stage(e, bytes);
    if (rc != GRV_OK) return rc;
    float *d_out = static_cast<float *>(e->stage_mem);
    hipStream_t cs;
    rc = control_stream(e, &cs);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, launch(d_out, cs));
    GRV_HIP(e, hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, cs));
    GRV_HIP(e, hipStreamSynchronize(cs));
    return GRV_OK;
}
} // namespace

extern "C" {

int grv_generate_spectrum_lut_device(grv_engine *e, size_t width, size_t height, double max_temp,
                                     float *d_out, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!d_out || width == 0 || height == 0 || width * height > (1ull << 26))
        return fail(e, GRV_ERR_INVALID, "bad LUT request");
    GRV_HIP(e, hipSetDevice(e->device));
    GRV_HIP(e, launch_spectrum_lut(d_out, (uint32_t)width, (uint32_t)height, max_temp,
                                   static_cast<hipStream_t>(stream)));
    return GRV_OK;
}

int grv_generate_spectrum_lut(grv_engine *e, size_t width, size_t height, double max_temp,
                              float *out_host) {
    if (!e) return GRV_ERR_INVALID;
    if (!out_host || width == 0 || height == 0 || width * height > (1ull << 26))
        return fail(e, GRV_ERR_INVALID, "bad LUT request");
    GRV_HIP(e, hipSetDevice(e->device));
    const size_t bytes = width * height * 4 * sizeof(float);
    int rc = ensure_stage(e, bytes);
    if (rc != GRV_OK) return rc;
    hipStream_t cs;
    rc = control_stream(e, &cs);
    if (rc != GRV_OK) return rc;
    rc = grv_generate_spectrum_lut_device(e, width, height, max_temp, static_cast<float *>(e->stage_mem), cs);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, hipMemcpyAsync(out_host, e->stage_mem, bytes, hipMemcpyDeviceToHost, cs));
    GRV_HIP(e, hipStreamSynchronize(cs));
    return GRV_OK;
}

int grv_strict_math(grv_engine *e, int op, size_t n, const double *x, const double *y, double *out) {
    if (!e) return GRV_ERR_INVALID;
    if ((op & ~GRV_MATH_F32) < GRV_MATH_SINCOS_SIN || (op & ~GRV_MATH_F32) > GRV_MATH_DIV_CONST ||
        ((op & GRV_MATH_F32) && (op & ~GRV_MATH_F32) > GRV_MATH_ATAN2))
        return fail(e, GRV_ERR_INVALID, "bad op %d", op);
    if (n == 0) return GRV_OK;
    if (!x || !out || (((op & ~GRV_MATH_F32) == GRV_MATH_POW || ((op & ~GRV_MATH_F32) >= GRV_MATH_ATAN2 && (op & ~GRV_MATH_F32) != GRV_MATH_RCP_R2)) && !y) || n > (1ull << 26))
        return fail(e, GRV_ERR_INVALID, "bad strict_math request");
    GRV_HIP(e, hipSetDevice(e->device));
    const size_t b = align_up(n * sizeof(double), 256);
    int rc = ensure_stage(e, 3 * b);
    if (rc != GRV_OK) return rc;
    char *base = static_cast<char *>(e->stage_mem);
    double *dx = reinterpret_cast<double *>(base), *dy = reinterpret_cast<double *>(base + b),
           *dout = reinterpret_cast<double *>(base + 2 * b);
    GRV_HIP(e, hipMemcpy(dx, x, n * sizeof(double), hipMemcpyHostToDevice));
    if (y) GRV_HIP(e, hipMemcpy(dy, y, n * sizeof(double), hipMemcpyHostToDevice));
    GRV_HIP(e, launch_strict_math(op, (uint32_t)n, dx, y ? dy : dx, dout, nullptr));
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(out, dout, n * sizeof(double), hipMemcpyDeviceToHost));
    return GRV_OK;
}

int grv_strict_rhs_probe(grv_engine *e, int form, size_t n, const double *states, double *out) {
    if (!e) return GRV_ERR_INVALID;
    if (form < GRV_RHS_FORM_IEEE || form > GRV_RHS_FORM_NOFIXUP) return fail(e, GRV_ERR_INVALID, "bad form %d", form);
    if (n == 0) return GRV_OK;
    if (!states || !out || n > (1ull << 24)) return fail(e, GRV_ERR_INVALID, "bad rhs_probe request");
    GRV_HIP(e, hipSetDevice(e->device));
    const size_t b_in = align_up(n * 8 * sizeof(double), 256), b_out = align_up(n * 7 * sizeof(double), 256);
    int rc = ensure_stage(e, b_in + b_out);
    if (rc != GRV_OK) return rc;
    char *base = static_cast<char *>(e->stage_mem);
    double *din = reinterpret_cast<double *>(base), *dout = reinterpret_cast<double *>(base + b_in);
    GRV_HIP(e, hipMemcpy(din, states, n * 8 * sizeof(double), hipMemcpyHostToDevice));
    GRV_HIP(e, launch_strict_rhs_probe(form, (uint32_t)n, e->mass, e->spin_c * e->mass, din, dout, nullptr));
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(out, dout, n * 7 * sizeof(double), hipMemcpyDeviceToHost));
    return GRV_OK;
}

int grv_strict_math_host(int op, size_t n, const double *x, const double *y, double *out) {
    const int base = op & ~GRV_MATH_F32;
    if (base < GRV_MATH_SINCOS_SIN || base > GRV_MATH_DIV_CONST) return GRV_ERR_INVALID;
    if ((op & GRV_MATH_F32) && base > GRV_MATH_ATAN2) return GRV_ERR_INVALID;
    if (n == 0) return GRV_OK;
    if (!x || !out || ((base == GRV_MATH_POW || (base >= GRV_MATH_ATAN2 && base != GRV_MATH_RCP_R2)) && !y)) return GRV_ERR_INVALID;
    const bool f32 = (op & GRV_MATH_F32) != 0;
    for (size_t i = 0; i < n; ++i) {
        const double a = f32 ? (double)(float)x[i] : x[i];
        const double b = y ? (f32 ? (double)(float)y[i] : y[i]) : 0.0;
        double r;
        switch (base) {
        case GRV_MATH_SINCOS_SIN:
        case GRV_MATH_SIN: r = strictm::sl_sin(a); break;
        case GRV_MATH_SINCOS_COS:
        case GRV_MATH_COS: r = strictm::sl_cos(a); break;
        case GRV_MATH_POW: r = strictm::sl_pow(a, b); break;
        case GRV_MATH_EXP: r = strictm::sl_exp(a); break;
        case GRV_MATH_ATAN: r = strictm::sl_atan(a); break;
        case GRV_MATH_LOG: r = strictm::sl_log(a); break;
        case GRV_MATH_ACOS: r = strictm::sl_acos(a); break;
        case GRV_MATH_DIV:
        case GRV_MATH_DIV_SHARED:
        case GRV_MATH_DIV_NOFIX:
        case GRV_MATH_DIV_CONST: r = a / b; break; // the host's IEEE quotient: what every device form must return
        case GRV_MATH_RCP_R2: r = 1.0 / a; break;  // correctly rounded; the device's refined seed equals it for most a
        default: r = strictm::sl_atan2(a, b); break;
        }
        out[i] = f32 ? (double)(float)r : r;
    }
    return GRV_OK;
}

const float *grv_get_sab_ptr(const grv_engine *e) { return e ? e->sab.data() : nullptr; }

int grv_attach_sab(grv_engine *e, float *ptr) {
    if (!e) return GRV_ERR_INVALID;
    e->sab_ext = ptr;
    return GRV_OK;
}

void grv_set_camera_state(grv_engine *e, double px, double py, double pz) {
    if (!e) return;
    e->camera.position[0] = px;
    e->camera.position[1] = py;
    e->camera.position[2] = pz;
}

void grv_set_auto_spin(grv_engine *e, int enabled) {
    if (e) e->camera.auto_spin = enabled != 0;
}

int grv_tick_sab(grv_engine *e, double dt_override) {
    if (!e) return GRV_ERR_INVALID;
    tick_sab_host(e->sab_block(), e->mass, e->spin, e->spin_c, event_horizon(e->mass, e->spin_c),
                  isco_prograde(e->mass, e->spin_c), e->camera, e->last_good_camera, dt_override);
    return GRV_OK;
}

double grv_compute_disk_flux(const grv_engine *e, double r) {
    return page_thorne_flux_host(r, e->mass, e->spin_c, 1.0);
}

double grv_compute_shadow_radius(const grv_engine *e) { return schwarzschild_shadow_radius_host(e->mass); }

size_t grv_compute_shadow_curve(const grv_engine *e, double theta_obs, size_t n_points, float *out,
                                size_t out_capacity) {
    if (!e) return 0;
    const std::vector<double> c = bardeen_shadow_host(e->mass, e->spin_c, theta_obs, n_points);
    if (out) {
        size_t n = c.size() < out_capacity ? c.size() : out_capacity;
        n &= ~(size_t)1; // whole (alpha, beta) pairs
        for (size_t i = 0; i < n; ++i) out[i] = (float)c[i];
    }
    return c.size() / 2;
}

int grv_compute_shadow_shift(const grv_engine *e, double theta_obs, float out2[2]) {
    if (!e || !out2) return GRV_ERR_INVALID;
    const std::vector<double> c = bardeen_shadow_host(e->mass, e->spin_c, theta_obs, 32);
    double lo = 0.0, hi = 0.0;
    if (!c.empty()) {
        lo = hi = c[0];
        for (size_t i = 0; i < c.size(); i += 2) {
            lo = c[i] < lo ? c[i] : lo;
            hi = c[i] > hi ? c[i] : hi;
        }
    }
    out2[0] = (float)lo;
    out2[1] = (float)hi;
    return GRV_OK;
}

int grv_generate_disk_lut(grv_engine *e, float *out512) {
    if (!e) return GRV_ERR_INVALID;
    if (!out512) return fail(e, GRV_ERR_INVALID, "null output");
    GRV_HIP(e, hipSetDevice(e->device));
    const uint32_t w = 512; // lut_width, lib.rs:65
    int rc = ensure_stage(e, 4096 + w * sizeof(double));
    if (rc != GRV_OK) return rc;
    float *d_out = static_cast<float *>(e->stage_mem);
    double *d_tmp = reinterpret_cast<double *>(static_cast<char *>(e->stage_mem) + 4096);
    hipStream_t cs;
    rc = control_stream(e, &cs);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, launch_disk_temperature_lut(d_out, d_tmp, w, e->mass, e->spin_c, cs));
    GRV_HIP(e, hipMemcpyAsync(out512, d_out, w * sizeof(float), hipMemcpyDeviceToHost, cs));
    GRV_HIP(e, hipStreamSynchronize(cs));
    std::memcpy(e->disk_lut.data(), out512, w * sizeof(float)); // self.lut_buffer = ... (lib.rs:108)
    return GRV_OK;
}

const float *grv_get_disk_lut_ptr(const grv_engine *e) { return e ? e->disk_lut.data() : nullptr; }

// ---- spacetime read-outs (spacetime_viz.hip) ----

double grv_compute_kretschner(const grv_engine *e, double r, double theta) {
    return e ? viz_kretschner(viz_hole(e), r, theta) : NAN;
}
double grv_compute_light_cone_tilt(const grv_engine *e, double r, double theta) {
    return e ? viz_light_cone_tilt(viz_hole(e), r, theta) : NAN;
}
double grv_compute_frame_drag_omega(const grv_engine *e, double r, double theta) {
    return e ? viz_frame_drag_omega(viz_hole(e), r, theta) : NAN;
}
double grv_compute_flamm_height(const grv_engine *e, double r) {
    return e ? viz_flamm_height(r, e->mass) : NAN;
}
double grv_compute_proper_distance(const grv_engine *e, double r1, double r2, size_t n_steps) {
    return e ? viz_proper_distance(viz_hole(e), r1, r2, n_steps) : NAN;
}

int grv_generate_field(grv_engine *e, int field, double r_min, double r_max, size_t n_radial,
                       size_t n_polar, float *out) {
    if (e && (field < GRV_FIELD_CURVATURE || field > GRV_FIELD_FRAME_DRAG))
        return fail(e, GRV_ERR_INVALID, "unknown field %d", field);
    return viz_grid(e, n_radial, n_polar, out, [&](float *d, hipStream_t cs) {
        return launch_viz_field(field, viz_hole(e), r_min, r_max, (uint32_t)n_radial,
                                (uint32_t)n_polar, d, cs);
    });
}

int grv_generate_embedding_mesh(grv_engine *e, double r_min, double r_max, size_t n_radial,
                                size_t n_angular, float *out) {
    return viz_grid(e, n_radial, n_angular, out, [&](float *d, hipStream_t cs) {
        return launch_embedding_mesh(viz_hole(e), r_min, r_max, (uint32_t)n_radial,
                                     (uint32_t)n_angular, d, cs);
    });
}

int grv_generate_ergosphere_mesh(grv_engine *e, size_t n_polar, size_t n_azimuthal, float *out) {
    return viz_grid(e, n_polar, n_azimuthal, out, [&](float *d, hipStream_t cs) {
        return launch_ergosphere_mesh(viz_hole(e), (uint32_t)n_polar, (uint32_t)n_azimuthal, d, cs);
    });
}

void grv_get_sab_layout(size_t out5[5]) {
    if (!out5) return;
    out5[0] = kOffControl;
    out5[1] = kOffCamera;
    out5[2] = kOffPhysics;
    out5[3] = kOffTelemetry;
    out5[4] = kOffLuts;
}

} // extern "C"
