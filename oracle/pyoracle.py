"""ctypes binding of the CPU oracle (oracle/libgravitas_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by the engine package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgravitas_oracle.so")

KERR_BL, KERR_KS, SCHWARZSCHILD = 0, 1, 2
TERM_NONE, TERM_HORIZON, TERM_ESCAPE, TERM_MAXSTEPS, TERM_DISK = 0, 1, 2, 3, 4
METHOD_RKF45, METHOD_RK4, METHOD_SYMPLECTIC = 0, 1, 2


class Metric(C.Structure):
    _fields_ = [("kind", C.c_int), ("mass", C.c_double), ("spin", C.c_double)]


class State(C.Structure):
    _fields_ = [("x", C.c_double * 4), ("p", C.c_double * 4)]


class Options(C.Structure):
    _fields_ = [("method", C.c_int), ("tolerance", C.c_double), ("initial_step", C.c_double),
                ("max_steps", C.c_uint64), ("escape_radius", C.c_double),
                ("renormalize_interval", C.c_uint64), ("step_size", C.c_double)]


class Trajectory(C.Structure):
    _fields_ = [("final_state", State), ("termination", C.c_int), ("steps_taken", C.c_uint64),
                ("max_hamiltonian_drift", C.c_double), ("rkf_tries", C.c_uint64)]


class Camera(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("inv_view", C.c_double * 16),
                ("inv_proj", C.c_double * 16), ("pixel_offset", C.c_double * 2)]


class FrameParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("metric_kind", C.c_int),
                ("mass", C.c_double), ("spin", C.c_double), ("opt", Options),
                ("shading", C.c_int), ("disk_inner", C.c_double), ("disk_outer", C.c_double),
                ("disk_temp", C.c_double), ("disk_opacity", C.c_double), ("exposure", C.c_double),
                ("lut_width", C.c_uint32), ("lut_height", C.c_uint32),
                ("lut_max_temp", C.c_double), ("disk_profile", C.c_int), ("disk_lut", C.c_void_p)]


class FrameStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("accepted_steps", C.c_uint64), ("rkf_tries", C.c_uint64),
                ("term_count", C.c_uint64 * 5), ("crossings", C.c_uint64),
                ("max_drift", C.c_double)]


class CameraState(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("velocity", C.c_double * 3),
                ("orientation", C.c_double * 4), ("auto_spin", C.c_int)]


class SabEngine(C.Structure):
    _fields_ = [("mass", C.c_double), ("spin", C.c_double), ("camera", CameraState),
                ("last_good", CameraState), ("sab", C.c_float * 2048)]


class WgslParams(C.Structure):
    _fields_ = [("inv_view", C.c_float * 16), ("inv_proj", C.c_float * 16),
                ("position", C.c_float * 3), ("mass", C.c_float), ("spin", C.c_float),
                ("width", C.c_uint32), ("height", C.c_uint32), ("jitter", C.c_float * 2),
                ("max_steps", C.c_int32), ("stars", C.c_int32)]


class GlslParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("mass", C.c_float),
                ("spin", C.c_float), ("zoom", C.c_float), ("mouse", C.c_float * 2),
                ("disk_size", C.c_float), ("disk_scale_height", C.c_float),
                ("disk_density", C.c_float), ("disk_temp", C.c_float),
                ("lensing_strength", C.c_float), ("time", C.c_float), ("turbulence", C.c_float),
                ("max_ray_steps", C.c_int32), ("tone_map", C.c_int32),
                ("features", C.c_uint32), ("quality", C.c_int32),
                ("show_redshift", C.c_float), ("show_kerr_shadow", C.c_float),
                ("debug", C.c_float), ("cam_pos", C.c_float * 3), ("cam_quat", C.c_float * 4),
                ("shadow_count", C.c_float), ("shadow_curve", (C.c_float * 2) * 64),
                ("noise_r", C.c_void_p), ("blue_r", C.c_void_p)]


def build(force=False):
    """Compile the oracle with its Makefile (gcc); make decides what is stale."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        d, i, u64, p = C.c_double, C.c_int, C.c_uint64, C.c_void_p
        MP, SP, OP = C.POINTER(Metric), C.POINTER(State), C.POINTER(Options)
        for name in ("orc_sin", "orc_cos", "orc_exp", "orc_atan", "orc_log", "orc_acos"):  # ref_libm.c
            getattr(L, name).restype = d
            getattr(L, name).argtypes = [d]
        for name in ("orc_pow", "orc_atan2"):
            getattr(L, name).restype = d
            getattr(L, name).argtypes = [d, d]
        for name in ("orc_sinf", "orc_cosf", "orc_expf", "orc_logf", "orc_acosf"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float]
        for name in ("orc_powf", "orc_atan2f"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float, C.c_float]
        L.orc_metric_make.restype = Metric
        L.orc_metric_make.argtypes = [i, d, d]
        L.orc_options_default.restype = Options
        for name in ("orc_event_horizon", "orc_cauchy_horizon", "orc_photon_sphere"):
            getattr(L, name).restype = d
            getattr(L, name).argtypes = [MP]
        L.orc_isco.restype = d
        L.orc_isco.argtypes = [MP, i]
        L.orc_ergosphere.restype = d
        L.orc_ergosphere.argtypes = [MP, d]
        L.orc_keplerian_frequency.restype = d
        L.orc_keplerian_frequency.argtypes = [MP, d]
        L.orc_time_dilation.restype = d
        L.orc_time_dilation.argtypes = [MP, d, d]
        L.orc_compute_dilation.restype = d
        L.orc_compute_dilation.argtypes = [MP, d]
        L.orc_covariant.argtypes = [MP, d, d, p]
        L.orc_contravariant.argtypes = [MP, d, d, p]
        L.orc_hamiltonian_derivatives.argtypes = [MP, d, d, p, p, p]
        L.orc_contract.restype = d
        L.orc_contract.argtypes = [p, p]
        L.orc_state_derivative.restype = State
        L.orc_state_derivative.argtypes = [SP, MP]
        L.orc_hamiltonian.restype = d
        L.orc_hamiltonian.argtypes = [SP, MP]
        L.orc_renormalize_null.argtypes = [SP, MP]
        L.orc_carter_constant.restype = d
        L.orc_carter_constant.argtypes = [SP, MP]
        L.orc_rkf45_step.restype = d
        L.orc_rkf45_step.argtypes = [SP, MP, d, SP]
        L.orc_adaptive_step.restype = d
        L.orc_adaptive_step.argtypes = [SP, MP, d, d, C.POINTER(u64)]
        L.orc_step_rk4.argtypes = [SP, MP, d]
        L.orc_step_symplectic.argtypes = [SP, MP, d]
        L.orc_integrate.argtypes = [SP, MP, OP, C.POINTER(Trajectory)]
        L.orc_integrate_path.restype = C.c_size_t
        L.orc_integrate_path.argtypes = [SP, MP, OP, C.POINTER(Trajectory), p, C.c_size_t]
        L.orc_integrate_ray_relativistic.restype = C.c_size_t
        L.orc_integrate_ray_relativistic.argtypes = [d, d, p, C.c_size_t, u64, d, i, p]
        L.orc_integrate_batch.argtypes = [MP, OP, C.c_size_t, p, p, p, p, p, p, i]
        L.orc_kerr_g_factor.restype = d
        L.orc_kerr_g_factor.argtypes = [d, d, d, d]
        L.orc_intensity_scaling.restype = d
        L.orc_intensity_scaling.argtypes = [d, i]
        L.orc_doppler_factor.restype = d
        L.orc_doppler_factor.argtypes = [d, d]
        L.orc_gravitational_factor.restype = d
        L.orc_gravitational_factor.argtypes = [d, d]
        L.orc_planck_law.restype = d
        L.orc_planck_law.argtypes = [d, d]
        L.orc_integrate_planck_xyz.argtypes = [d, p]
        L.orc_cie_1931.argtypes = [d, p]
        L.orc_xyz_to_linear_rgb.argtypes = [d, d, d, p]
        L.orc_generate_blackbody_lut.argtypes = [C.c_size_t, C.c_size_t, d, p]
        L.orc_max_threads.restype = i
        CP, FP = C.POINTER(Camera), C.POINTER(FrameParams)
        L.orc_camera_look_at.argtypes = [p, p, p, d, d, CP]
        L.orc_pixel_state.argtypes = [CP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, SP]
        L.orc_render_frame.argtypes = [CP, FP, p, C.c_uint32, C.c_uint32, p, p, p, p, p,
                                       C.POINTER(FrameStats), i]
        L.orc_lut_sample.argtypes = [p, C.c_uint32, C.c_uint32, d, d, d, p]
        L.orc_disk_temp_profile.restype = d
        L.orc_disk_temp_profile.argtypes = [d, d]
        L.orc_page_thorne_flux.restype = d
        L.orc_page_thorne_flux.argtypes = [d, d, d, d]
        L.orc_disk_temperature.restype = d
        L.orc_disk_temperature.argtypes = [d, d, d, d]
        L.orc_generate_temperature_lut.argtypes = [d, d, C.c_size_t, p]
        L.orc_schwarzschild_shadow_radius.restype = d
        L.orc_schwarzschild_shadow_radius.argtypes = [d]
        L.orc_bardeen_shadow.restype = C.c_size_t
        L.orc_bardeen_shadow.argtypes = [d, d, d, C.c_size_t, p]
        L.orc_wgsl_frame.argtypes = [C.POINTER(WgslParams), C.c_uint32, C.c_uint32, p, p, i]
        L.orc_glsl_frame.argtypes = [C.POINTER(GlslParams), C.c_uint32, C.c_uint32, p, p, i]
        L.orc_seeded_noise_rgba8.argtypes = [C.c_uint32, C.c_uint32, p]
        # test hooks of the f32 shader oracle (single building blocks, shader_oracle.c)
        f32 = C.c_float
        L.orc_hook_wgsl_derivs.argtypes = [p, p, f32, f32, p, p]
        L.orc_hook_wgsl_step.argtypes = [p, p, f32, f32, f32, p, p]
        L.orc_hook_glsl_accel.argtypes = [p, p, f32, f32, p, p]
        L.orc_hook_glsl_blackbody.argtypes = [f32, p]
        L.orc_hook_glsl_verlet_oscillator.argtypes = [p, p, f32, f32, i]
        for name, n in (("orc_hook_wgsl_horizon", 2), ("orc_hook_wgsl_isco", 2),
                        ("orc_hook_glsl_horizon", 2), ("orc_hook_glsl_isco", 2),
                        ("orc_hook_glsl_photon_sphere", 2), ("orc_hook_glsl_redshift_potential", 2),
                        ("orc_hook_glsl_ergosphere_radius", 3), ("orc_hook_glsl_beaming", 1),
                        ("orc_hook_glsl_disk_delta", 5)):
            getattr(L, name).restype = f32
            getattr(L, name).argtypes = [f32] * n
        L.orc_round_to_half.restype = C.c_float
        L.orc_round_to_half.argtypes = [C.c_float]
        L.orc_taa_resolve.argtypes = [C.c_uint32, C.c_uint32, p, p, C.c_float, i, i, p]
        L.orc_ataa_resolve.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(AtaaCamera), p, p, i, p]
        L.orc_bloom.argtypes = [C.c_uint32, C.c_uint32, p, C.c_float, C.c_float, i, i, p]
        L.orc_sab_engine_init.argtypes = [C.POINTER(SabEngine), d, d]
        L.orc_camera_update.argtypes = [C.POINTER(CameraState), d, d, d, d]
        L.orc_tick_sab.argtypes = [C.POINTER(SabEngine), d]
        for name in ("orc_kretschner_kerr", "orc_light_cone_tilt_bl", "orc_frame_dragging_omega"):
            getattr(L, name).restype = d
            getattr(L, name).argtypes = [d, d, d, d]
        L.orc_ergosphere_radius.restype = d
        L.orc_ergosphere_radius.argtypes = [d, d, d]
        L.orc_flamm_height.restype = d
        L.orc_flamm_height.argtypes = [d, d]
        L.orc_kerr_embedding_height.restype = d
        L.orc_kerr_embedding_height.argtypes = [d, d, C.c_size_t, d, d]
        L.orc_proper_distance.restype = d
        L.orc_proper_distance.argtypes = [d, d, C.c_size_t, d, d]
        L.orc_scalar_field.argtypes = [i, d, d, d, d, C.c_size_t, C.c_size_t, p]
        L.orc_embedding_mesh.argtypes = [d, d, d, d, C.c_size_t, C.c_size_t, p]
        L.orc_ergosphere_mesh.argtypes = [d, d, C.c_size_t, C.c_size_t, p]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def ref_sin(x):
    f = lib().orc_sin
    return np.array([f(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def ref_cos(x):
    f = lib().orc_cos
    return np.array([f(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def ref_fn(name, x, y=None):
    """orc_<name> of ref_libm.c over arrays (name: sin, cos, exp, log, atan, acos, pow, atan2, and
    the f32 forms sinf ... atan2f)."""
    f = getattr(lib(), "orc_" + name)
    dt = np.float32 if name.endswith("f") else np.float64
    x = np.asarray(x, dt)
    if y is None:
        return np.array([f(v.item()) for v in x.ravel()], dt).reshape(x.shape)
    x, y = np.broadcast_arrays(x, np.asarray(y, dt))
    return np.array([f(a.item(), b.item()) for a, b in zip(x.ravel(), y.ravel())], dt).reshape(x.shape)


def ref_exp(x):
    f = lib().orc_exp
    return np.array([f(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def ref_atan(x):
    f = lib().orc_atan
    return np.array([f(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def ref_pow(x, y):
    f = lib().orc_pow
    x, y = np.broadcast_arrays(np.asarray(x, np.float64), np.asarray(y, np.float64))
    return np.array([f(float(a), float(b)) for a, b in zip(x.ravel(), y.ravel())]).reshape(x.shape)


def metric(kind, mass, spin):
    return lib().orc_metric_make(kind, float(mass), float(spin))


def options(method=METHOD_RKF45, tolerance=1e-8, initial_step=0.01, max_steps=10000,
            escape_radius=1000.0, renormalize_interval=10, step_size=0.0):
    return Options(method, tolerance, initial_step, max_steps, escape_radius,
                   renormalize_interval, step_size)


def make_state(v8):
    s = State()
    for k in range(4):
        s.x[k] = float(v8[k])
        s.p[k] = float(v8[4 + k])
    return s


def state_to_list(s):
    return [s.x[0], s.x[1], s.x[2], s.x[3], s.p[0], s.p[1], s.p[2], s.p[3]]


def contravariant(m, r, theta):
    g = np.zeros(16)
    lib().orc_contravariant(C.byref(m), r, theta, _ptr(g))
    return g


def covariant(m, r, theta):
    g = np.zeros(16)
    lib().orc_covariant(C.byref(m), r, theta, _ptr(g))
    return g


def hamiltonian_derivatives(m, r, theta, p4):
    p = np.asarray(p4, dtype=np.float64)
    a, b = C.c_double(), C.c_double()
    lib().orc_hamiltonian_derivatives(C.byref(m), r, theta, _ptr(p), C.byref(a), C.byref(b))
    return a.value, b.value


def integrate(v8, m, opt):
    s = make_state(v8)
    t = Trajectory()
    lib().orc_integrate(C.byref(s), C.byref(m), C.byref(opt), C.byref(t))
    return t


def integrate_path(v8, m, opt, cap=20000):
    s = make_state(v8)
    t = Trajectory()
    path = np.zeros((cap, 8))
    n = lib().orc_integrate_path(C.byref(s), C.byref(m), C.byref(opt), C.byref(t), _ptr(path), cap)
    return t, path[:n].copy()


def integrate_ray_relativistic(mass, spin, initial, steps, tolerance, use_ks):
    a = np.ascontiguousarray(initial, dtype=np.float64)
    out = np.zeros(max(8, a.size))
    n = lib().orc_integrate_ray_relativistic(mass, spin, _ptr(a), a.size, steps, tolerance,
                                             1 if use_ks else 0, _ptr(out))
    return out[:n].copy()


def integrate_batch(m, opt, states, nthreads=1):
    """states: (n, 8) f64 AoS.  Returns dict of arrays."""
    a = np.ascontiguousarray(states, dtype=np.float64)
    n = a.shape[0]
    out = np.zeros_like(a)
    steps = np.zeros(n, np.uint32)
    term = np.zeros(n, np.uint8)
    drift = np.zeros(n, np.float64)
    tries = np.zeros(n, np.uint32)
    lib().orc_integrate_batch(C.byref(m), C.byref(opt), n, _ptr(a), _ptr(out), _ptr(steps),
                              _ptr(term), _ptr(drift), _ptr(tries), nthreads)
    return dict(states=out, steps=steps, term=term, drift=drift, tries=tries)


def blackbody_lut(width, height, max_temp):
    out = np.zeros(width * height * 4, np.float32)
    lib().orc_generate_blackbody_lut(width, height, max_temp, _ptr(out))
    return out


def camera_look_at(eye, target=(0, 0, 0), up=(0, 1, 0), fovy_deg=60.0, aspect=16 / 9):
    cam = Camera()
    e = np.asarray(eye, np.float64)
    t = np.asarray(target, np.float64)
    u = np.asarray(up, np.float64)
    lib().orc_camera_look_at(_ptr(e), _ptr(t), _ptr(u), np.deg2rad(fovy_deg), aspect,
                             C.byref(cam))
    return cam


def pixel_state(cam, width, height, i, j):
    s = State()
    lib().orc_pixel_state(C.byref(cam), width, height, i, j, C.byref(s))
    return np.array(state_to_list(s))


def frame_params(width, height, mass=1.0, spin=0.999, metric_kind=KERR_KS, opt=None, shading=1,
                 disk_inner=0.0, disk_outer=30.0, disk_temp=9500.0, disk_opacity=0.6,
                 exposure=1.0, lut_width=512, lut_height=64, lut_max_temp=1e5, disk_profile=0):
    if opt is None:
        opt = options(max_steps=2048)
    return FrameParams(width, height, metric_kind, mass, spin, opt, shading, disk_inner,
                       disk_outer, disk_temp, disk_opacity, exposure, lut_width, lut_height,
                       lut_max_temp, disk_profile, None)


def render_frame(cam, fp, lut=None, stride=(1, 1), nthreads=1, want_states=True):
    nx = (fp.width + stride[0] - 1) // stride[0]
    ny = (fp.height + stride[1] - 1) // stride[1]
    n = nx * ny
    rgba = np.zeros((ny, nx, 4), np.float32)
    states = np.zeros((n, 8), np.float64) if want_states else None
    steps = np.zeros(n, np.uint32)
    term = np.zeros(n, np.uint8)
    drift = np.zeros(n, np.float64)
    st = FrameStats()
    lib().orc_render_frame(C.byref(cam), C.byref(fp), _ptr(lut), stride[0], stride[1],
                           _ptr(rgba), _ptr(states), _ptr(steps), _ptr(term), _ptr(drift),
                           C.byref(st), nthreads)
    return dict(rgba=rgba, states=states, steps=steps, term=term, drift=drift, stats=st,
                shape=(ny, nx))


def temperature_lut(mass, spin, width=512):
    out = np.zeros(width, np.float32)
    lib().orc_generate_temperature_lut(mass, spin, width, _ptr(out))
    return out


def bardeen_shadow(mass, spin, theta_obs, n_points):
    out = np.zeros(4 * n_points + 4, np.float64)
    n = lib().orc_bardeen_shadow(mass, spin, theta_obs, n_points, _ptr(out))
    return out[:2 * n].reshape(n, 2).copy()


def scalar_field(kind, mass, spin, r_min, r_max, n_radial, n_polar):
    """kind: 0 Kretschner, 1 light-cone tilt, 2 frame-drag omega -> (r, theta, value) f32 triples"""
    out = np.zeros(3 * n_radial * n_polar, np.float32)
    lib().orc_scalar_field(kind, mass, spin, r_min, r_max, n_radial, n_polar, _ptr(out))
    return out


def embedding_mesh(mass, spin, r_min, r_max, n_radial, n_angular):
    out = np.zeros(3 * n_radial * n_angular, np.float32)
    lib().orc_embedding_mesh(mass, spin, r_min, r_max, n_radial, n_angular, _ptr(out))
    return out


def ergosphere_mesh(mass, spin, n_polar, n_azimuthal):
    out = np.zeros(3 * n_polar * n_azimuthal, np.float32)
    lib().orc_ergosphere_mesh(mass, spin, n_polar, n_azimuthal, _ptr(out))
    return out


def sab_engine(mass, spin):
    e = SabEngine()
    lib().orc_sab_engine_init(C.byref(e), mass, spin)
    return e


def tick_sab(e, dt_override):
    lib().orc_tick_sab(C.byref(e), dt_override)
    return np.frombuffer(e.sab, dtype=np.float32).copy()


def wgsl_params_from(gp):
    """oracle WgslParams from the engine's GrvWgslParams (same uniform values)."""
    o = WgslParams()
    for k in range(16):
        o.inv_view[k] = gp.inv_view[k]
        o.inv_proj[k] = gp.inv_proj[k]
    for k in range(3):
        o.position[k] = gp.position[k]
    o.mass, o.spin, o.width, o.height = gp.mass, gp.spin, gp.width, gp.height
    o.jitter[0], o.jitter[1] = gp.jitter[0], gp.jitter[1]
    o.max_steps = gp.max_steps
    o.stars = gp.stars
    return o


class AtaaCamera(C.Structure):
    _fields_ = [("inv_view", C.c_float * 16), ("inv_proj", C.c_float * 16),
                ("prev_view_proj", C.c_float * 16), ("position", C.c_float * 3)]


def taa_resolve(current, history, blend_factor=0.75, camera_moving=False, half_storage=True):
    h, w, _ = current.shape
    out = np.zeros_like(current)
    lib().orc_taa_resolve(w, h, _ptr(np.ascontiguousarray(current, np.float32)),
                          _ptr(np.ascontiguousarray(history, np.float32)), blend_factor,
                          1 if camera_moving else 0, 1 if half_storage else 0, _ptr(out))
    return out


def ataa_resolve(cam, current, history, half_storage=True):
    h, w, _ = current.shape
    out = np.zeros_like(current)
    lib().orc_ataa_resolve(w, h, C.byref(cam), _ptr(np.ascontiguousarray(current, np.float32)),
                           _ptr(np.ascontiguousarray(history, np.float32)), 1 if half_storage else 0,
                           _ptr(out))
    return out


def bloom(scene, threshold=0.8, intensity=0.5, blur_passes=2, half_storage=True):
    h, w, _ = scene.shape
    out = np.zeros_like(scene)
    lib().orc_bloom(w, h, _ptr(np.ascontiguousarray(scene, np.float32)), threshold, intensity,
                    blur_passes, 1 if half_storage else 0, _ptr(out))
    return out


def seeded_noise_rgba8(seed, size=256):
    out = np.zeros(size * size * 4, np.uint8)
    lib().orc_seeded_noise_rgba8(int(seed), int(size), _ptr(out))
    return out


def glsl_params_from(gp, noise_rgba8=None, blue_rgba8=None):
    """Oracle uniforms from the engine-side GrvGlslParams mirror.  Textures default to the
    seeded planes an engine starts with (seeds 1 and 2)."""
    o = GlslParams()
    for f, _ in GlslParams._fields_:
        if f in ("noise_r", "blue_r"):
            continue
        v = getattr(gp, f)
        if f in ("mouse", "cam_pos", "cam_quat"):
            dst = getattr(o, f)
            for j in range(len(dst)):
                dst[j] = v[j]
        elif f == "shadow_curve":
            for j in range(64):
                o.shadow_curve[j][0], o.shadow_curve[j][1] = v[j][0], v[j][1]
        else:
            setattr(o, f, v)
    noise = seeded_noise_rgba8(1) if noise_rgba8 is None else np.asarray(noise_rgba8, np.uint8)
    blue = seeded_noise_rgba8(2) if blue_rgba8 is None else np.asarray(blue_rgba8, np.uint8)
    o._planes = (np.ascontiguousarray(noise.reshape(-1, 4)[:, 0]),
                 np.ascontiguousarray(blue.reshape(-1, 4)[:, 0]))  # keep alive
    o.noise_r = o._planes[0].ctypes.data
    o.blue_r = o._planes[1].ctypes.data
    return o


def _shader_frame(fn, params, stride, nthreads):
    nx = (params.width + stride[0] - 1) // stride[0]
    ny = (params.height + stride[1] - 1) // stride[1]
    rgba = np.zeros((ny, nx, 4), np.float32)
    steps = np.zeros((ny, nx), np.uint32)
    fn(C.byref(params), stride[0], stride[1], _ptr(rgba), _ptr(steps), nthreads)
    return rgba, steps


def wgsl_frame(params, stride=(1, 1), nthreads=1):
    return _shader_frame(lib().orc_wgsl_frame, params, stride, nthreads)


def wgsl_pixels_f64(params, xy, nthreads=1):
    """The compute march in double (wgsl_f64_twin.c) at the pixels xy[n, 2] -> dict of rgb[n, 3],
    steps[n], cls[n] (0 horizon, 1 escape, 2 budget, 3 opaque), min_r[n], axis_margin[n] (min over the
    visited states of min(theta, pi - theta): closest approach to the polar axis, negative = crossed)."""
    xy = np.ascontiguousarray(xy, np.uint32).reshape(-1, 2)
    n = xy.shape[0]
    rgb, steps = np.zeros((n, 3)), np.zeros(n, np.uint32)
    cls, min_r, min_sin = np.zeros(n, np.int32), np.zeros(n), np.zeros(n)
    L = lib()
    L.orc_wgsl_pixels_f64.argtypes = [C.POINTER(WgslParams), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_wgsl_pixels_f64.restype = None
    L.orc_wgsl_pixels_f64(C.byref(params), n, _ptr(xy), _ptr(rgb), _ptr(steps), _ptr(cls), _ptr(min_r), _ptr(min_sin),
                          nthreads)
    return {"rgb": rgb, "steps": steps, "cls": cls, "min_r": min_r, "axis_margin": min_sin}


def glsl_frame(params, stride=(1, 1), nthreads=1):
    return _shader_frame(lib().orc_glsl_frame, params, stride, nthreads)
