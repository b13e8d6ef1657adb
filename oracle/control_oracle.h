/*
 * control_oracle.h -- CPU restatement of the rows SURVEY.md 8(f) ranks next to the
 * path: Novikov-Thorne disk (physics/disk.rs), Bardeen shadow curve (physics/shadow.rs),
 * camera filter (gravitas-wasm/src/camera.rs) and the tick_sab SAB protocol
 * (gravitas-wasm/src/lib.rs:308-409).  TEST INFRASTRUCTURE ONLY.
 * Pinned by the reference's tests disk.rs:226-309 and shadow.rs:260-335.
 */
#ifndef CONTROL_ORACLE_H
#define CONTROL_ORACLE_H

#include "gravitas_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

double orc_page_thorne_flux(double r, double mass, double spin, double m_dot);
double orc_disk_temperature(double r, double mass, double spin, double m_dot);
void orc_generate_temperature_lut(double mass, double spin, size_t width, float *out);

double orc_schwarzschild_shadow_radius(double mass);
/* writes (alpha, beta) pairs; out must hold 2 * 2 * n_points doubles; returns point count */
size_t orc_bardeen_shadow(double mass, double spin, double theta_obs, size_t n_points,
                          double *out);

typedef struct {
    double position[3];
    double velocity[3];
    double orientation[4]; /* x y z w */
    int auto_spin;
} orc_camera_state;

typedef struct {
    double mass, spin;
    orc_camera_state camera, last_good;
    float sab[2048];
} orc_sab_engine;

void orc_sab_engine_init(orc_sab_engine *e, double mass, double spin);
void orc_camera_update(orc_camera_state *s, double mouse_dx, double mouse_dy, double zoom_delta,
                       double dt);
void orc_tick_sab(orc_sab_engine *e, double dt_override);

#ifdef __cplusplus
}
#endif
#endif
