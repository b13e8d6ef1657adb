/*
 * viz_oracle.h -- CPU restatement of the spacetime read-outs the reference's FFI exposes next
 * to the path: gravitas-core/src/spacetime/{curvature,lightcone,frame_drag,
 * embedding}.rs behind gravitas-wasm/src/lib.rs:139-306.  TEST INFRASTRUCTURE ONLY.
 * Pinned by the reference's tests embedding.rs:113-129 (the only ones these files hold); the
 * curvature, light-cone, frame-drag and mesh functions are "parity unpinned" upstream and are
 * pinned here by closed forms (tests/test_spacetime_viz.py).
 * `spin` is the dimensionless a/M; functions that take the engine's Boyer-Lindquist Kerr
 * metric receive the spin already clamped to [-1, 1] (metric/kerr.rs:48-63).
 */
#ifndef VIZ_ORACLE_H
#define VIZ_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

double orc_kretschner_kerr(double r, double theta, double mass, double spin);
double orc_light_cone_tilt_bl(double r, double theta, double mass, double spin);
double orc_frame_dragging_omega(double r, double theta, double mass, double spin);
double orc_ergosphere_radius(double theta, double mass, double spin);
double orc_flamm_height(double r, double mass);
double orc_kerr_embedding_height(double r, double r_ref, size_t n_steps, double mass, double spin);
double orc_proper_distance(double r1, double r2, size_t n_steps, double mass, double spin);

/* (r, theta, value) triples as f32; out holds 3 * n_radial * n_polar floats.
 * kind: 0 Kretschner scalar, 1 light-cone tilt, 2 frame-drag omega */
void orc_scalar_field(int kind, double mass, double spin, double r_min, double r_max,
                      size_t n_radial, size_t n_polar, float *out);
/* xyz vertices as f32; spin_raw is the engine's unclamped spin (lib.rs:146-148) */
void orc_embedding_mesh(double mass, double spin_raw, double r_min, double r_max, size_t n_radial,
                        size_t n_angular, float *out);
void orc_ergosphere_mesh(double mass, double spin, size_t n_polar, size_t n_azimuthal, float *out);

#ifdef __cplusplus
}
#endif
#endif
