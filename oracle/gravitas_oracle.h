/*
 * gravitas_oracle.h -- CPU restatement of the reference's geodesic hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * engine: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / CPU baseline.
 *
 * PARITY STATUS: the closed-form pieces (radii, metric, BL<->KS Hamiltonian
 * invariance, g-factor) are pinned against every known-answer test the
 * reference holds (see tests/test_oracle_pins.py).  The reference has NO test
 * that executes integrate()/integrate_ray_relativistic and its
 * Rust toolchain is absent here, so trajectory ENDPOINTS ARE "parity unpinned"
 * by the reference itself; they are cross-checked against an independent
 * scipy DOP853 integration of the same Hamilton equations instead.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference/physics-engine/).
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, IEEE divides) so the
 * operation order is the reference's.
 */
#ifndef GRAVITAS_ORACLE_H
#define GRAVITAS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* gravitas-core/src/geodesic/mod.rs:23-30  (#[repr(C)] GeodesicState) */
typedef struct {
    double x[4]; /* t, r, theta, phi */
    double p[4]; /* p_t, p_r, p_theta, p_phi (covariant) */
} orc_state;

/* metric selector: Kerr in BL / KS (metric/kerr.rs:17-22), Schwarzschild
 * (metric/schwarzschild.rs:20-23) */
enum { ORC_KERR_BL = 0, ORC_KERR_KS = 1, ORC_SCHWARZSCHILD = 2 };

typedef struct {
    int kind;
    double mass;
    double spin; /* dimensionless a*, clamped to [-1,1] by orc_metric_make */
} orc_metric;

/* geodesic/termination.rs:4-17 (#[repr(C)]) */
enum { ORC_TERM_NONE = 0, ORC_TERM_HORIZON = 1, ORC_TERM_ESCAPE = 2, ORC_TERM_MAXSTEPS = 3,
       ORC_TERM_DISK = 4 };

/* geodesic/integrator.rs:13-21 */
enum { ORC_METHOD_RKF45 = 0, ORC_METHOD_RK4 = 1, ORC_METHOD_SYMPLECTIC = 2 };

/* geodesic/integrator.rs:24-33 (+ step_size from IntegrationMethod) */
typedef struct {
    int method;
    double tolerance;
    double initial_step;
    uint64_t max_steps;
    double escape_radius;
    uint64_t renormalize_interval;
    double step_size; /* RK4 / Symplectic fixed step */
} orc_options;

/* geodesic/mod.rs:149-161 (path omitted) + try/reject counters (ours) */
typedef struct {
    orc_state final_state;
    int termination;
    uint64_t steps_taken;
    double max_hamiltonian_drift;
    uint64_t rkf_tries; /* total RKF45 evaluations incl. rejected + forced */
} orc_trajectory;

orc_metric orc_metric_make(int kind, double mass, double spin);
orc_options orc_options_default(void);

/* closed forms */
double orc_event_horizon(const orc_metric *m);
double orc_cauchy_horizon(const orc_metric *m);
double orc_photon_sphere(const orc_metric *m);
double orc_isco(const orc_metric *m, int retrograde);
double orc_ergosphere(const orc_metric *m, double theta);
double orc_keplerian_frequency(const orc_metric *m, double r);
double orc_time_dilation(const orc_metric *m, double r, double theta);
double orc_compute_dilation(const orc_metric *m, double r);

/* metric tensors, row-major g[mu*4+nu] */
void orc_covariant(const orc_metric *m, double r, double theta, double g[16]);
void orc_contravariant(const orc_metric *m, double r, double theta, double g[16]);
void orc_hamiltonian_derivatives(const orc_metric *m, double r, double theta, const double p[4],
                                 double *dh_dr, double *dh_dtheta);
double orc_contract(const double g[16], const double p[4]);

/* geodesic */
orc_state orc_state_derivative(const orc_state *s, const orc_metric *m);
double orc_hamiltonian(const orc_state *s, const orc_metric *m);
void orc_renormalize_null(orc_state *s, const orc_metric *m);
double orc_carter_constant(const orc_state *s, const orc_metric *m);
double orc_rkf45_step(const orc_state *s, const orc_metric *m, double h, orc_state *out);
double orc_adaptive_step(orc_state *s, const orc_metric *m, double h_try, double tolerance,
                         uint64_t *tries);
void orc_step_rk4(orc_state *s, const orc_metric *m, double h);
void orc_step_symplectic(orc_state *s, const orc_metric *m, double h);
void orc_integrate(const orc_state *initial, const orc_metric *m, const orc_options *opt,
                   orc_trajectory *out);
/* optional path recording: writes up to cap states (incl. initial), returns count */
size_t orc_integrate_path(const orc_state *initial, const orc_metric *m, const orc_options *opt,
                          orc_trajectory *out, orc_state *path, size_t cap);

/* gravitas-wasm/src/lib.rs:422-464; returns number of doubles written (n if n<8) */
size_t orc_integrate_ray_relativistic(double mass, double spin, const double *initial, size_t n,
                                      uint64_t steps, double tolerance, int use_kerr_schild,
                                      double *out);

/* batch over rays (OpenMP static chunks when nthreads>1); AoS in/out.
 * steps/term/drift/tries may be NULL. */
void orc_integrate_batch(const orc_metric *m, const orc_options *opt, size_t n,
                         const orc_state *in, orc_state *out, uint32_t *steps, uint8_t *term,
                         double *drift, uint32_t *tries, int nthreads);

/* physics/redshift.rs */
double orc_kerr_g_factor(double r, double mass, double spin, double lambda);
double orc_intensity_scaling(double g, int optically_thick);
double orc_doppler_factor(double beta, double cos_theta);
double orc_gravitational_factor(double r, double mass);

/* physics/spectrum.rs */
double orc_planck_law(double lambda, double temperature);
void orc_integrate_planck_xyz(double temperature, double xyz[3]);
void orc_cie_1931(double lambda, double xyz[3]);
void orc_xyz_to_linear_rgb(double x, double y, double z, float rgb[3]);
void orc_generate_blackbody_lut(size_t width, size_t height, double max_temp, float *out);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
