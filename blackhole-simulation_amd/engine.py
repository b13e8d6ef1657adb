"""ctypes host binding of libgravitas_hip.so.

`PhysicsEngine` keeps the method names, argument meaning and error behaviour of
the reference's wasm-bindgen class (physics-engine/gravitas-wasm/src/lib.rs:56-465;
TS consumers src/engine/physics-bridge.ts, src/workers/physics.worker.ts), plus the
batch / frame extensions of include/gravitas_abi.h.  PyTorch is optional plumbing:
device-pointer entry points accept any object exposing ``data_ptr()``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgravitas_hip.so")

KERR_BL, KERR_KS, SCHWARZSCHILD = 0, 1, 2
METHOD_RKF45, METHOD_RK4, METHOD_SYMPLECTIC = 0, 1, 2
ARITH_STRICT, ARITH_FAST, ARITH_FAST_PACKED = 0, 1, 2
SCHEDULE_DEFAULT, SCHEDULE_SLOT_ORDER = 0, 1  # GrvRenderParams.schedule
FAULT_NONE, FAULT_RENDER, FAULT_SEND, FAULT_PEER_COPY = 0, 1, 2, 3  # grv_multi_test_inject_fault
DISK_PROFILE_SHORTCUT, DISK_PROFILE_PAGE_THORNE = 0, 1
MATH_SINCOS_SIN, MATH_SINCOS_COS, MATH_SIN, MATH_COS, MATH_POW, MATH_EXP, MATH_ATAN = 0, 1, 2, 3, 4, 5, 6
MATH_LOG, MATH_ACOS, MATH_ATAN2, MATH_F32 = 7, 8, 9, 16
MATH_DIV, MATH_DIV_SHARED, MATH_DIV_NOFIX, MATH_RCP_R2, MATH_DIV_CONST = 10, 11, 12, 13, 14
RHS_FORM_IEEE, RHS_FORM_SHARED, RHS_FORM_NOFIXUP = 0, 1, 2
TERM_NONE, TERM_HORIZON, TERM_ESCAPE, TERM_MAXSTEPS, TERM_DISK_CROSSING = 0, 1, 2, 3, 4
_STATUS = {0: "GRV_OK", 1: "GRV_ERR_INVALID", 2: "GRV_ERR_NO_DEVICE", 3: "GRV_ERR_HIP",
           4: "GRV_ERR_OOM"}


class GravitasError(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [("method", C.c_int32), ("metric_kind", C.c_int32), ("tolerance", C.c_double),
                ("initial_step", C.c_double), ("max_steps", C.c_uint64),
                ("escape_radius", C.c_double), ("renormalize_interval", C.c_uint64),
                ("step_size", C.c_double), ("arith", C.c_int32), ("segment_tries", C.c_int32),
                ("record_path", C.c_int32), ("reserved", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("inv_view", C.c_double * 16),
                ("inv_proj", C.c_double * 16), ("pixel_offset", C.c_double * 2)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("opt", Options),
                ("shading", C.c_int32), ("reserved0", C.c_int32), ("disk_inner", C.c_double),
                ("disk_outer", C.c_double), ("disk_temp", C.c_double),
                ("disk_opacity", C.c_double), ("exposure", C.c_double),
                ("lut_width", C.c_uint32), ("lut_height", C.c_uint32),
                ("lut_max_temp", C.c_double), ("tile_world", C.c_uint32),
                ("tile_rank", C.c_uint32), ("segment_tries", C.c_uint32),
                ("profile", C.c_uint32), ("disk_profile", C.c_uint32), ("schedule", C.c_uint32)]


class FrameStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("accepted_steps", C.c_uint64), ("rkf_tries", C.c_uint64),
                ("term_count", C.c_uint64 * 5), ("crossings", C.c_uint64),
                ("max_drift", C.c_double), ("launches", C.c_uint32), ("init_ms", C.c_float),
                ("integrate_ms", C.c_float), ("compact_ms", C.c_float), ("shade_ms", C.c_float),
                ("total_ms", C.c_float)]


class WgslParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("inv_view", C.c_float * 16),
                ("inv_proj", C.c_float * 16), ("position", C.c_float * 3), ("mass", C.c_float),
                ("spin", C.c_float), ("jitter", C.c_float * 2), ("max_steps", C.c_int32),
                ("tile_world", C.c_uint32), ("tile_rank", C.c_uint32), ("arith", C.c_int32),
                ("stars", C.c_int32)]


class GlslParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("mass", C.c_float),
                ("spin", C.c_float), ("zoom", C.c_float), ("mouse", C.c_float * 2),
                ("disk_size", C.c_float), ("disk_scale_height", C.c_float),
                ("disk_density", C.c_float), ("disk_temp", C.c_float),
                ("lensing_strength", C.c_float), ("time", C.c_float), ("turbulence", C.c_float),
                ("max_ray_steps", C.c_int32), ("tone_map", C.c_int32),
                ("tile_world", C.c_uint32), ("tile_rank", C.c_uint32),
                ("features", C.c_uint32), ("quality", C.c_int32),
                ("show_redshift", C.c_float), ("show_kerr_shadow", C.c_float),
                ("debug", C.c_float), ("cam_pos", C.c_float * 3), ("cam_quat", C.c_float * 4),
                ("shadow_count", C.c_float), ("shadow_curve", (C.c_float * 2) * 64),
                ("arith", C.c_int32)]


# ShaderManager #defines as GlslParams.features bits (include/gravitas_abi.h)
GLSL_LENSING, GLSL_DISK, GLSL_DOPPLER, GLSL_STARS = 1, 2, 4, 8
GLSL_PHOTON_GLOW, GLSL_JETS, GLSL_REDSHIFT, GLSL_DITHER = 16, 32, 64, 128
GLSL_FEATURES_DEFAULT = (GLSL_LENSING | GLSL_DISK | GLSL_DOPPLER | GLSL_STARS | GLSL_PHOTON_GLOW
                         | GLSL_JETS | GLSL_DITHER)


class TaaParams(C.Structure):  # reprojection.glsl.ts uniforms
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("blend_factor", C.c_float),
                ("camera_moving", C.c_int32), ("half_storage", C.c_int32), ("arith", C.c_int32)]


class AtaaParams(C.Structure):  # ataa.wgsl.ts CameraUniforms subset
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("inv_view", C.c_float * 16),
                ("inv_proj", C.c_float * 16), ("prev_view_proj", C.c_float * 16),
                ("position", C.c_float * 3), ("half_storage", C.c_int32), ("arith", C.c_int32)]


class BloomParams(C.Structure):  # bloom.ts BloomConfig
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("intensity", C.c_float),
                ("threshold", C.c_float), ("blur_passes", C.c_int32), ("half_storage", C.c_int32),
                ("arith", C.c_int32)]


class FrameBuffers(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("final_state", C.c_void_p), ("steps", C.c_void_p),
                ("termination", C.c_void_p), ("drift", C.c_void_p)]


def library_path():
    return _LIB


def build_library(force=False):
    """Compile libgravitas_hip.so for gfx950 with the in-tree Makefile (hipcc)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "-s", "clean"])
    subprocess.check_call(["make", "-C", csrc, "-s", "-j4"])
    return _LIB


_lib = None


def load_library():
    """dlopen the engine.  Raises if the HIP library has not been built: this
    package has no other compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise GravitasError(
            "libgravitas_hip.so is missing (%s): run __graft_entry__.build() / make -C "
            "blackhole-simulation_amd/csrc; there is no CPU fallback" % _LIB)
    try:
        # PyTorch-ROCm ships its own libamdhip64; two HIP runtimes in one process cannot
        # both open the device.  Import torch first so this library binds to the copy
        # torch already loaded.  (Standalone C/N-API hosts use the system runtime.)
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(_LIB)
    d, i, p, sz = C.c_double, C.c_int, C.c_void_p, C.c_size_t
    L.grv_abi_version.restype = i
    L.grv_engine_create.restype = i
    L.grv_engine_create.argtypes = [d, d, i, C.POINTER(p)]
    L.grv_engine_destroy.argtypes = [p]
    L.grv_last_error.restype = C.c_char_p
    L.grv_last_error.argtypes = [p]
    L.grv_update_params.restype = i
    L.grv_update_params.argtypes = [p, d, d]
    for name in ("grv_compute_horizon", "grv_compute_isco", "grv_compute_photon_sphere"):
        getattr(L, name).restype = d
        getattr(L, name).argtypes = [p]
    L.grv_compute_dilation.restype = d
    L.grv_compute_dilation.argtypes = [p, d]
    L.grv_compute_g_factor.restype = d
    L.grv_compute_g_factor.argtypes = [p, d, d]
    L.grv_integrate_ray_relativistic.restype = sz
    L.grv_integrate_ray_relativistic.argtypes = [p, p, sz, sz, d, i, p]
    L.grv_integrate_batch.restype = i
    L.grv_integrate_batch.argtypes = [p, sz, p, C.POINTER(Options), p, p, p, p]
    L.grv_integrate_batch_device.restype = i
    L.grv_integrate_batch_device.argtypes = [p, sz, p, C.POINTER(Options), p, p, p, p, p]
    L.grv_integrate_paths.restype = i
    L.grv_integrate_paths.argtypes = [p, sz, p, C.POINTER(Options), sz, p, p, p, p, p, p]
    L.grv_integrate_paths_device.restype = i
    L.grv_integrate_paths_device.argtypes = [p, sz, p, C.POINTER(Options), sz, p, p, p, p, p, p, p]
    L.grv_tile_pitch.restype = C.c_uint32
    L.grv_tile_pitch.argtypes = [C.c_uint32, C.c_uint32]
    for name, args in (("grv_tiles_total", 3), ("grv_max_tiles_per_rank", 3)):
        getattr(L, name).restype = C.c_uint32
        getattr(L, name).argtypes = [C.c_uint32] * args
    L.grv_tiles_of_rank.restype = C.c_uint32
    L.grv_tiles_of_rank.argtypes = [C.c_uint32] * 4 + [p, C.c_uint32]
    L.grv_tile_origin.restype = None
    L.grv_tile_origin.argtypes = [C.c_uint32] * 3 + [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.grv_engine_host_bytes.restype = sz
    L.grv_engine_host_bytes.argtypes = [p]
    L.grv_engine_profile_shader_frames.restype = i
    L.grv_engine_profile_shader_frames.argtypes = [p, i]
    L.grv_test_hooks_unlock.restype = i
    L.grv_test_hooks_unlock.argtypes = [C.c_uint32]
    L.grv_test_hooks_unlocked.restype = i
    L.grv_test_hooks_unlocked.argtypes = []
    L.grv_test_try_bound.restype = C.c_uint32
    L.grv_test_try_bound.argtypes = [p]
    L.grv_multi_test_inject_fault.restype = i
    L.grv_multi_test_inject_fault.argtypes = [p, i, i]
    L.grv_multi_set_exchange_format.restype = i
    L.grv_multi_set_exchange_format.argtypes = [p, i]
    L.grv_multi_exchange_format.restype = i
    L.grv_multi_exchange_format.argtypes = [p]
    L.grv_multi_exchange_bytes_per_frame.restype = sz
    L.grv_multi_exchange_bytes_per_frame.argtypes = [p, C.c_uint32, C.c_uint32]
    L.grv_frame_ray_count.restype = sz
    L.grv_frame_ray_count.argtypes = [C.POINTER(RenderParams)]
    L.grv_render_frame.restype = i
    L.grv_render_frame.argtypes = [p, C.POINTER(Camera), C.POINTER(RenderParams), p,
                                   C.POINTER(FrameStats)]
    L.grv_render_frame_device.restype = i
    L.grv_render_frame_device.argtypes = [p, C.POINTER(Camera), C.POINTER(RenderParams),
                                          C.POINTER(FrameBuffers), p]
    L.grv_frame_stats.restype = i
    L.grv_frame_stats.argtypes = [p, p, C.POINTER(FrameStats)]
    L.grv_stats_accumulate.restype = i
    L.grv_stats_accumulate.argtypes = [p, i]
    L.grv_frame_stats_reset.restype = i
    L.grv_frame_stats_reset.argtypes = [p, p]
    L.grv_engine_device_bytes.restype = sz
    L.grv_engine_device_bytes.argtypes = [p]
    L.grv_integrate_ray_relativistic_ex.restype = sz
    L.grv_integrate_ray_relativistic_ex.argtypes = [p, p, sz, sz, d, i, p, p, p, p]
    L.grv_unpack_tiles.restype = i
    L.grv_unpack_tiles.argtypes = [C.POINTER(RenderParams), C.c_uint32, p, p, sz]
    L.grv_unpack_tiles_device.restype = i
    L.grv_unpack_tiles_device.argtypes = [p, C.POINTER(RenderParams), C.c_uint32, p, p, sz, p]
    L.grv_camera_look_at.argtypes = [p, p, p, d, d, C.POINTER(Camera)]
    L.grv_camera_from_uniforms.argtypes = [p, C.POINTER(Camera)]
    L.grv_render_params_default.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(RenderParams)]
    L.grv_options_default.argtypes = [C.POINTER(Options)]
    L.grv_generate_spectrum_lut.restype = i
    L.grv_generate_spectrum_lut.argtypes = [p, sz, sz, d, p]
    L.grv_strict_math_host.restype = i
    L.grv_strict_math_host.argtypes = [i, sz, p, p, p]
    L.grv_strict_math.restype = i
    L.grv_strict_math.argtypes = [p, i, sz, p, p, p]
    L.grv_strict_rhs_probe.restype = i
    L.grv_strict_rhs_probe.argtypes = [p, i, sz, p, p]
    L.grv_generate_spectrum_lut_device.restype = i
    L.grv_generate_spectrum_lut_device.argtypes = [p, sz, sz, d, p, p]
    L.grv_wgsl_params_default.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(Camera), d, d,
                                          C.POINTER(WgslParams)]
    L.grv_glsl_params_default.argtypes = [C.c_uint32, C.c_uint32, d, d, C.POINTER(GlslParams)]
    L.grv_render_frame_wgsl.restype = i
    L.grv_render_frame_wgsl.argtypes = [p, C.POINTER(WgslParams), p, p, C.POINTER(C.c_uint64), p]
    L.grv_render_frame_glsl.restype = i
    L.grv_render_frame_glsl.argtypes = [p, C.POINTER(GlslParams), p, p, C.POINTER(C.c_uint64), p]
    L.grv_post_taa_resolve.restype = i
    L.grv_post_taa_resolve.argtypes = [p, C.POINTER(TaaParams), p, p, p, p]
    L.grv_taa_effective_blend.restype = C.c_float
    L.grv_taa_effective_blend.argtypes = [C.c_float, C.c_float]
    L.grv_post_ataa_resolve.restype = i
    L.grv_post_ataa_resolve.argtypes = [p, C.POINTER(AtaaParams), p, p, p, p]
    L.grv_bloom_params_default.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(BloomParams)]
    L.grv_post_bloom.restype = i
    L.grv_post_bloom.argtypes = [p, C.POINTER(BloomParams), p, p, p]
    L.grv_webgpu_render.restype = i
    L.grv_webgpu_render.argtypes = [p, p, p, C.c_int32, C.c_int32, p, p]
    L.grv_webgl_render.restype = i
    L.grv_webgl_render.argtypes = [p, C.POINTER(GlslParams), C.c_int32, C.c_int32, p, p]
    L.grv_renderer_reset.argtypes = [p]
    L.grv_renderer_frame_count.restype = C.c_uint32
    L.grv_renderer_frame_count.argtypes = [p]
    L.grv_seeded_noise_rgba8.argtypes = [C.c_uint32, C.c_uint32, p]
    L.grv_set_glsl_noise.restype = i
    L.grv_set_glsl_noise.argtypes = [p, p, p]
    L.grv_generate_disk_lut.restype = i
    L.grv_generate_disk_lut.argtypes = [p, p]
    L.grv_compute_disk_flux.restype = d
    L.grv_compute_disk_flux.argtypes = [p, d]
    L.grv_compute_shadow_curve.restype = sz
    L.grv_compute_shadow_curve.argtypes = [p, d, sz, p, sz]
    L.grv_compute_shadow_radius.restype = d
    L.grv_compute_shadow_radius.argtypes = [p]
    L.grv_compute_shadow_shift.restype = i
    L.grv_compute_shadow_shift.argtypes = [p, d, p]
    L.grv_get_disk_lut_ptr.restype = p
    L.grv_get_disk_lut_ptr.argtypes = [p]
    for name in ("grv_compute_kretschner", "grv_compute_light_cone_tilt",
                 "grv_compute_frame_drag_omega"):
        getattr(L, name).restype = d
        getattr(L, name).argtypes = [p, d, d]
    L.grv_compute_flamm_height.restype = d
    L.grv_compute_flamm_height.argtypes = [p, d]
    L.grv_compute_proper_distance.restype = d
    L.grv_compute_proper_distance.argtypes = [p, d, d, sz]
    L.grv_generate_field.restype = i
    L.grv_generate_field.argtypes = [p, i, d, d, sz, sz, p]
    L.grv_generate_embedding_mesh.restype = i
    L.grv_generate_embedding_mesh.argtypes = [p, d, d, sz, sz, p]
    L.grv_generate_ergosphere_mesh.restype = i
    L.grv_generate_ergosphere_mesh.argtypes = [p, sz, sz, p]
    # device images (ABI 8)
    L.grv_image_create.restype = i
    L.grv_image_create.argtypes = [p, C.c_uint32, C.c_uint32, C.POINTER(p)]
    L.grv_image_create_shared.restype = i
    L.grv_image_create_shared.argtypes = [p, C.c_uint32, C.c_uint32, p, C.POINTER(p)]
    L.grv_image_destroy.argtypes = [p]
    L.grv_image_width.restype = C.c_uint32
    L.grv_image_width.argtypes = [p]
    L.grv_image_height.restype = C.c_uint32
    L.grv_image_height.argtypes = [p]
    L.grv_image_bytes.restype = sz
    L.grv_image_bytes.argtypes = [p]
    L.grv_image_data.restype = p
    L.grv_image_data.argtypes = [p]
    L.grv_image_stream.restype = p
    L.grv_image_stream.argtypes = [p]
    L.grv_image_last_error.restype = C.c_char_p
    L.grv_image_last_error.argtypes = [p]
    L.grv_render_frame_image.restype = i
    L.grv_render_frame_image.argtypes = [p, C.POINTER(Camera), C.POINTER(RenderParams), p]
    L.grv_render_frame_glsl_image.restype = i
    L.grv_render_frame_glsl_image.argtypes = [p, C.POINTER(GlslParams), p]
    L.grv_render_frame_wgsl_image.restype = i
    L.grv_render_frame_wgsl_image.argtypes = [p, C.POINTER(WgslParams), p]
    L.grv_webgl_render_image.restype = i
    L.grv_webgl_render_image.argtypes = [p, C.POINTER(GlslParams), C.c_int32, C.c_int32, p]
    L.grv_webgpu_render_image.restype = i
    L.grv_webgpu_render_image.argtypes = [p, p, p, C.c_int32, C.c_int32, p]
    L.grv_post_bloom_image.restype = i
    L.grv_post_bloom_image.argtypes = [p, C.POINTER(BloomParams), p, p]
    L.grv_post_taa_resolve_image.restype = i
    L.grv_post_taa_resolve_image.argtypes = [p, C.POINTER(TaaParams), p, p, p]
    L.grv_image_read_async.restype = i
    L.grv_image_read_async.argtypes = [p, p, sz]
    L.grv_image_read.restype = i
    L.grv_image_read.argtypes = [p, p, sz]
    L.grv_image_wait.restype = i
    L.grv_image_wait.argtypes = [p]
    L.grv_image_query.restype = i
    L.grv_image_query.argtypes = [p]
    L.grv_image_frame_stats.restype = i
    L.grv_image_frame_stats.argtypes = [p, C.POINTER(FrameStats)]
    L.grv_engine_synchronize.restype = i
    L.grv_engine_synchronize.argtypes = [p]
    L.grv_engine_create_multi.restype = i
    L.grv_engine_create_multi.argtypes = [d, d, C.c_uint64, i, C.POINTER(p)]
    L.grv_engine_create_multi_virtual.restype = i
    L.grv_engine_create_multi_virtual.argtypes = [d, d, i, i, C.POINTER(p)]
    L.grv_multi_destroy.argtypes = [p]
    L.grv_multi_last_error.restype = C.c_char_p
    L.grv_multi_last_error.argtypes = [p]
    for name in ("grv_multi_rank_count", "grv_multi_transport", "grv_multi_synchronize",
                 "grv_multi_frame_stats_reset"):
        getattr(L, name).restype = i
        getattr(L, name).argtypes = [p]
    L.grv_multi_rank_device.restype = i
    L.grv_multi_rank_device.argtypes = [p, i]
    L.grv_multi_engine.restype = p
    L.grv_multi_engine.argtypes = [p, i]
    L.grv_multi_update_params.restype = i
    L.grv_multi_update_params.argtypes = [p, d, d]
    L.grv_render_frame_multi_device.restype = i
    L.grv_render_frame_multi_device.argtypes = [p, C.POINTER(Camera), C.POINTER(RenderParams), p, p]
    L.grv_render_frame_multi.restype = i
    L.grv_render_frame_multi.argtypes = [p, C.POINTER(Camera), C.POINTER(RenderParams), p,
                                         C.POINTER(FrameStats)]
    L.grv_render_frame_wgsl_multi_device.restype = i
    L.grv_render_frame_wgsl_multi_device.argtypes = [p, C.POINTER(WgslParams), p, p]
    L.grv_multi_stats_accumulate.restype = i
    L.grv_multi_stats_accumulate.argtypes = [p, i]
    L.grv_multi_frame_stats.restype = i
    L.grv_multi_frame_stats.argtypes = [p, C.POINTER(FrameStats)]
    L.grv_multi_create_error.restype = C.c_char_p
    L.grv_multi_create_error.argtypes = []
    L.grv_rccl_probe.restype = i
    L.grv_rccl_probe.argtypes = [C.POINTER(C.c_int), C.c_char_p, sz]
    L.grv_multi_rank_frame_stats.restype = i
    L.grv_multi_rank_frame_stats.argtypes = [p, i, C.POINTER(FrameStats)]
    L.grv_last_ray_clocks.restype = i
    L.grv_last_ray_clocks.argtypes = [p, C.POINTER(C.c_uint64 * 3)]
    L.grv_engine_set_ray_arith.restype = i
    L.grv_engine_set_ray_arith.argtypes = [p, C.c_int32]
    L.grv_test_set_try_bound.restype = i
    L.grv_test_set_try_bound.argtypes = [p, C.c_uint32]
    L.grv_multi_test_self_exchange.restype = i
    L.grv_multi_test_self_exchange.argtypes = [p, i]
    L.grv_attach_sab.restype = i
    L.grv_attach_sab.argtypes = [p, p]
    L.grv_set_camera_state.argtypes = [p, d, d, d]
    L.grv_set_auto_spin.argtypes = [p, i]
    L.grv_tick_sab.restype = i
    L.grv_tick_sab.argtypes = [p, d]
    L.grv_get_sab_ptr.restype = p
    L.grv_get_sab_ptr.argtypes = [p]
    L.grv_get_sab_layout.argtypes = [p]
    _lib = L
    return L


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dev_ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


def rccl_probe():
    """(version code, "") when librccl binds, (0, reason) when it does not.  No device needed."""
    v = C.c_int(0)
    msg = C.create_string_buffer(512)
    rc = load_library().grv_rccl_probe(C.byref(v), msg, len(msg))
    return (v.value, "") if rc == 0 else (0, msg.value.decode())


def strict_math_host(op, x, y=None):
    """The STRICT contract's transcendental routines, host build (no device needed)."""
    x = np.ascontiguousarray(x, np.float64)
    y = None if y is None else np.ascontiguousarray(y, np.float64)
    out = np.zeros_like(x)
    rc = load_library().grv_strict_math_host(int(op), x.size, _np_ptr(x), _np_ptr(y), _np_ptr(out))
    if rc != 0:
        raise GravitasError("strict_math_host failed (%d)" % rc)
    return out


def default_options(**kw):
    o = Options()
    load_library().grv_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def camera_look_at(eye, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovy_deg=60.0,
                   aspect=16.0 / 9.0, pixel_offset=(0.5, 0.5)):
    cam = Camera()
    e = np.asarray(eye, np.float64)
    t = np.asarray(target, np.float64)
    u = np.asarray(up, np.float64)
    load_library().grv_camera_look_at(_np_ptr(e), _np_ptr(t), _np_ptr(u),
                                      float(np.deg2rad(fovy_deg)), float(aspect), C.byref(cam))
    cam.pixel_offset[0], cam.pixel_offset[1] = pixel_offset
    return cam


def render_params(width, height, **kw):
    p = RenderParams()
    load_library().grv_render_params_default(width, height, C.byref(p))
    for k, v in kw.items():
        if hasattr(p.opt, k) and not hasattr(p, k):
            setattr(p.opt, k, v)
        else:
            setattr(p, k, v)
    return p


def wgsl_params(width, height, camera, mass=1.0, spin=0.999, **kw):
    p = WgslParams()
    load_library().grv_wgsl_params_default(width, height, C.byref(camera), mass, spin, C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def seeded_noise_rgba8(seed, size=256):
    """Deterministic stand-in for createNoiseTexture's Math.random() bytes (RGBA8)."""
    out = np.zeros(size * size * 4, np.uint8)
    load_library().grv_seeded_noise_rgba8(int(seed), int(size), _np_ptr(out))
    return out


def glsl_params(width, height, mass=1.0, spin=0.999, **kw):
    p = GlslParams()
    load_library().grv_glsl_params_default(width, height, mass, spin, C.byref(p))
    for k, v in kw.items():
        if k == "shadow_curve":  # (n, 2) array of (alpha, beta); also sets shadow_count
            v = np.asarray(v, np.float32).reshape(-1, 2)[:64]
            for j in range(len(v)):
                p.shadow_curve[j][0], p.shadow_curve[j][1] = float(v[j, 0]), float(v[j, 1])
            p.shadow_count = float(len(v))
        elif k in ("cam_pos", "cam_quat", "mouse"):
            arr = getattr(p, k)
            for j, x in enumerate(v):
                arr[j] = float(x)
        else:
            setattr(p, k, v)
    return p


def unpack_tiles(params, rank, packed, channels, dtype):
    """Host scatter of one rank's packed tile-order pixels into a row-major image."""
    packed = np.ascontiguousarray(packed)
    img = np.zeros((params.height, params.width, channels), dtype)
    rc = load_library().grv_unpack_tiles(C.byref(params), rank, _np_ptr(packed), _np_ptr(img),
                                         channels * np.dtype(dtype).itemsize)
    if rc != 0:
        raise GravitasError("grv_unpack_tiles: %s" % _STATUS.get(rc, rc))
    return img


class DeviceImage:
    """A W x H RGBA f32 image in HBM with a stream of its own (include/gravitas_abi.h "device images"):
    the texture the reference's renderers hold between passes (webgpu/renderer.ts:280-411).  Frames and
    post passes into it are queued; pixels cross PCIe in read() only."""

    def __init__(self, engine, width, height, stream_of=None):
        self._lib = engine._lib
        h = C.c_void_p()
        if stream_of is not None:  # written on that image's compute stream, in queue order
            engine._check(self._lib.grv_image_create_shared(engine._h, int(width), int(height), stream_of._h,
                                                            C.byref(h)), "image_create_shared")
        else:
            engine._check(self._lib.grv_image_create(engine._h, int(width), int(height), C.byref(h)), "image_create")
        self._h = h
        self.width, self.height = int(width), int(height)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.grv_image_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.grv_image_last_error(self._h)
            raise GravitasError("%s: %s: %s" % (what, _STATUS.get(rc, rc), msg.decode() if msg else ""))

    @property
    def data_ptr(self):
        return self._lib.grv_image_data(self._h)

    @property
    def stream(self):
        return self._lib.grv_image_stream(self._h)

    def read(self, out=None):
        """-> float32 [h, w, 4] (one D2H behind everything queued on the image)"""
        if out is None:
            out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._lib.grv_image_read(self._h, _np_ptr(out), out.size), "image_read")
        return out

    def read_async(self, out):
        """queue the D2H into `out` (a pinned torch tensor or numpy array of h*w*4 float32) on the image's copy
        stream; wait() / ready() tell when it has landed"""
        ptr = out.data_ptr() if hasattr(out, "data_ptr") else out.ctypes.data
        n = out.numel() if hasattr(out, "numel") else out.size
        self._check(self._lib.grv_image_read_async(self._h, C.c_void_p(ptr), int(n)), "image_read_async")

    def wait(self):
        self._check(self._lib.grv_image_wait(self._h), "image_wait")

    def ready(self):
        q = self._lib.grv_image_query(self._h)
        if q < 0:
            self._check(-q, "image_query")
        return q == 1

    def stats(self):
        st = FrameStats()
        self._check(self._lib.grv_image_frame_stats(self._h, C.byref(st)), "image_frame_stats")
        return st


class PhysicsEngine:
    """`new PhysicsEngine(mass, spin)` (gravitas-wasm/src/lib.rs:59)."""

    def __init__(self, mass, spin, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.grv_engine_create(float(mass), float(spin), int(device), C.byref(h))
        if rc != 0:
            raise GravitasError("grv_engine_create failed: %s (no usable HIP device? this engine "
                                "has no CPU path)" % _STATUS.get(rc, rc))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.grv_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.grv_last_error(self._h)
            raise GravitasError("%s: %s: %s" % (what, _STATUS.get(rc, rc),
                                                msg.decode() if msg else ""))

    # ---- lib.rs:78-105, 202-205 ----
    def update_params(self, mass, spin):
        self._check(self._lib.grv_update_params(self._h, float(mass), float(spin)), "update_params")

    def compute_horizon(self):
        return self._lib.grv_compute_horizon(self._h)

    def compute_isco(self):
        return self._lib.grv_compute_isco(self._h)

    def compute_photon_sphere(self):
        return self._lib.grv_compute_photon_sphere(self._h)

    def compute_dilation(self, r):
        return self._lib.grv_compute_dilation(self._h, float(r))

    def compute_g_factor(self, r, lam):
        return self._lib.grv_compute_g_factor(self._h, float(r), float(lam))

    # ---- lib.rs:422-464 ----
    def integrate_ray_relativistic(self, initial_state, steps, tolerance, use_kerr_schild):
        a = np.ascontiguousarray(initial_state, dtype=np.float64)
        out = np.zeros(max(8, a.size), np.float64)
        n = self._lib.grv_integrate_ray_relativistic(self._h, _np_ptr(a), a.size, int(steps),
                                                     float(tolerance), 1 if use_kerr_schild else 0,
                                                     _np_ptr(out))
        return out[:n].copy()

    integratePhotonGeodesic = integrate_ray_relativistic  # BASELINE.json's name for the same export

    def set_ray_arith(self, arith):
        """Contract of the one-ray entry on this handle: ARITH_STRICT (default, the oracle's bits) or
        ARITH_FAST (rounding differences only, a third of the latency)."""
        self._check(self._lib.grv_engine_set_ray_arith(self._h, int(arith)), "set_ray_arith")

    # ---- batch extension ----
    def integrate_batch(self, states, options):
        a = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 8)
        n = a.shape[0]
        out = np.zeros_like(a)
        steps = np.zeros(n, np.uint32)
        term = np.zeros(n, np.uint8)
        drift = np.zeros(n, np.float64)
        self._check(self._lib.grv_integrate_batch(self._h, n, _np_ptr(a), C.byref(options),
                                                  _np_ptr(out), _np_ptr(steps), _np_ptr(term),
                                                  _np_ptr(drift)), "integrate_batch")
        return dict(states=out, steps=steps, term=term, drift=drift)

    def integrate_batch_device(self, n, d_states, options, d_out, d_steps=None, d_term=None,
                               d_drift=None, stream=None):
        self._check(self._lib.grv_integrate_batch_device(
            self._h, int(n), _dev_ptr(d_states), C.byref(options), _dev_ptr(d_out),
            _dev_ptr(d_steps), _dev_ptr(d_term), _dev_ptr(d_drift),
            C.c_void_p(stream) if stream else None), "integrate_batch_device")

    # ---- Trajectory.path (geodesic/mod.rs:150-161, IntegrationOptions.record_path) ----
    def integrate_paths(self, states, options, max_points=None):
        """integrate() with record_path: returns the batch dict plus `paths` (a list of [k, 8] arrays,
        the reference's Vec<GeodesicState> per ray: the initial state as handed in, then the state
        after every completed step) and `counts` (1 + steps_taken; a ray's array holds
        min(count, max_points) of them).  With options.record_path == 0 `paths` is None (path: None).
        max_points defaults to max_steps + 1, the longest path a ray can have."""
        a = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 8)
        n = a.shape[0]
        rec = bool(options.record_path)
        if max_points is None:
            max_points = int(options.max_steps) + 1
        out = np.zeros_like(a)
        steps = np.zeros(n, np.uint32)
        term = np.zeros(n, np.uint8)
        drift = np.zeros(n, np.float64)
        counts = np.zeros(n, np.uint32)
        rows = np.zeros((n, max_points, 8), np.float64) if rec else None
        self._check(self._lib.grv_integrate_paths(self._h, n, _np_ptr(a), C.byref(options), int(max_points),
                                                  _np_ptr(rows) if rec else None, _np_ptr(counts), _np_ptr(out),
                                                  _np_ptr(steps), _np_ptr(term), _np_ptr(drift)), "integrate_paths")
        paths = [rows[i, :min(int(counts[i]), max_points)].copy() for i in range(n)] if rec else None
        return dict(states=out, steps=steps, term=term, drift=drift, counts=counts, paths=paths)

    def integrate_paths_device(self, n, d_states, options, max_points, d_paths, d_counts, d_out=None,
                               d_steps=None, d_term=None, d_drift=None, stream=None):
        self._check(self._lib.grv_integrate_paths_device(
            self._h, int(n), _dev_ptr(d_states), C.byref(options), int(max_points), _dev_ptr(d_paths),
            _dev_ptr(d_counts), _dev_ptr(d_out), _dev_ptr(d_steps), _dev_ptr(d_term), _dev_ptr(d_drift),
            C.c_void_p(stream) if stream else None), "integrate_paths_device")

    # ---- frame ----
    def frame_ray_count(self, params):
        return self._lib.grv_frame_ray_count(C.byref(params))

    def render_frame(self, camera, params):
        """Host-buffer frame: returns (rgba[n,4] float32 in this rank's pixel order, FrameStats)."""
        n = self.frame_ray_count(params)
        rgba = np.zeros((n, 4), np.float32)
        st = FrameStats()
        self._check(self._lib.grv_render_frame(self._h, C.byref(camera), C.byref(params),
                                               _np_ptr(rgba), C.byref(st)), "render_frame")
        return rgba, st

    renderFrame = render_frame

    def render_frame_device(self, camera, params, rgba=None, final_state=None, steps=None,
                            termination=None, drift=None, stream=None):
        fb = FrameBuffers(_dev_ptr(rgba), _dev_ptr(final_state), _dev_ptr(steps),
                          _dev_ptr(termination), _dev_ptr(drift))
        self._check(self._lib.grv_render_frame_device(
            self._h, C.byref(camera), C.byref(params), C.byref(fb),
            C.c_void_p(stream) if stream else None), "render_frame_device")

    def render_frame_wgsl(self, params, rgba, steps=None, stream=None, want_total=True):
        """f32 WGSL-semantics frame (compute.wgsl.ts); returns total symplectic steps.
        want_total=False queues the frame without waiting (the count stays in the device-side
        counters, see stats_accumulate / frame_stats) and returns None."""
        tot = C.c_uint64(0)
        self._check(self._lib.grv_render_frame_wgsl(
            self._h, C.byref(params), _dev_ptr(rgba), _dev_ptr(steps),
            C.byref(tot) if want_total else None,
            C.c_void_p(stream) if stream else None), "render_frame_wgsl")
        return tot.value if want_total else None

    def render_frame_glsl(self, params, rgba, steps=None, stream=None, want_total=True):
        """f32 GLSL-semantics frame (fragment.glsl.ts march); returns total Verlet steps."""
        tot = C.c_uint64(0)
        self._check(self._lib.grv_render_frame_glsl(
            self._h, C.byref(params), _dev_ptr(rgba), _dev_ptr(steps),
            C.byref(tot) if want_total else None,
            C.c_void_p(stream) if stream else None), "render_frame_glsl")
        return tot.value if want_total else None

    def set_glsl_noise(self, noise_rgba8=None, blue_rgba8=None):
        """Override the shader's 256x256 RGBA8 noise / blue-noise textures (None keeps one)."""
        for a in (noise_rgba8, blue_rgba8):
            if a is not None and (a.dtype != np.uint8 or a.size != 256 * 256 * 4):
                raise ValueError("textures are 256x256 RGBA8")
        self._check(self._lib.grv_set_glsl_noise(self._h, _np_ptr(noise_rgba8), _np_ptr(blue_rgba8)),
                    "set_glsl_noise")

    # ---- renderer layer: WebGPURenderer.render / WebGLRenderer.render ----
    def webgpu_render(self, camera_uniforms, physics_params, screen, max_steps=150, arith=ARITH_STRICT,
                      stream=None):
        """camera_uniforms: float32[88] (352-byte CameraUniforms), physics_params: float32[8]
        (32-byte PhysicsParams, frame_index as u32 bits); screen: device RGBA f32 [h, w, 4]."""
        cu = np.ascontiguousarray(camera_uniforms, np.float32)
        pp = np.ascontiguousarray(physics_params, np.float32)
        assert cu.size == 88 and pp.size == 8
        self._check(self._lib.grv_webgpu_render(self._h, _np_ptr(cu), _np_ptr(pp), int(max_steps),
                                                int(arith), _dev_ptr(screen), stream), "webgpu_render")

    def webgl_render(self, params, screen, bloom=True, camera_moving=False, stream=None):
        self._check(self._lib.grv_webgl_render(self._h, C.byref(params), 1 if bloom else 0,
                                               1 if camera_moving else 0, _dev_ptr(screen), stream),
                    "webgl_render")

    def renderer_reset(self):
        self._lib.grv_renderer_reset(self._h)

    def renderer_frame_count(self):
        return int(self._lib.grv_renderer_frame_count(self._h))

    # ---- post chain (reprojection.ts / ataa.wgsl.ts / bloom.ts): device RGBA f32 images ----
    def post_taa_resolve(self, width, height, current, history, out, blend_factor=0.75,
                         camera_moving=False, half_storage=True, stream=None, arith=ARITH_STRICT):
        p = TaaParams(width, height, blend_factor, 1 if camera_moving else 0, 1 if half_storage else 0,
                      int(arith))
        self._check(self._lib.grv_post_taa_resolve(self._h, C.byref(p), _dev_ptr(current),
                                                   _dev_ptr(history), _dev_ptr(out), stream),
                    "post_taa_resolve")

    def post_ataa_resolve(self, params, current, history, out, stream=None):
        self._check(self._lib.grv_post_ataa_resolve(self._h, C.byref(params), _dev_ptr(current),
                                                    _dev_ptr(history), _dev_ptr(out), stream),
                    "post_ataa_resolve")

    def post_bloom(self, width, height, scene, out, stream=None, **kw):
        p = BloomParams()
        self._lib.grv_bloom_params_default(width, height, C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        self._check(self._lib.grv_post_bloom(self._h, C.byref(p), _dev_ptr(scene), _dev_ptr(out),
                                             stream), "post_bloom")

    # ---- device images (ABI 8): frames and post passes that stay in HBM ----
    def create_image(self, width, height, stream_of=None):
        return DeviceImage(self, width, height, stream_of)

    def render_frame_image(self, cam, params, image):
        self._check(self._lib.grv_render_frame_image(self._h, C.byref(cam), C.byref(params), image._h),
                    "render_frame_image")

    def render_frame_glsl_image(self, params, image):
        self._check(self._lib.grv_render_frame_glsl_image(self._h, C.byref(params), image._h), "render_frame_glsl_image")

    def render_frame_wgsl_image(self, params, image):
        self._check(self._lib.grv_render_frame_wgsl_image(self._h, C.byref(params), image._h), "render_frame_wgsl_image")

    def webgl_render_image(self, params, image, bloom=True, camera_moving=False):
        self._check(self._lib.grv_webgl_render_image(self._h, C.byref(params), int(bool(bloom)),
                                                     int(bool(camera_moving)), image._h), "webgl_render_image")

    def webgpu_render_image(self, camera_uniforms, physics_params, image, max_steps=150, arith=ARITH_STRICT):
        cu = np.ascontiguousarray(camera_uniforms, np.float32)
        pp = np.ascontiguousarray(physics_params, np.float32)
        self._check(self._lib.grv_webgpu_render_image(self._h, _np_ptr(cu), _np_ptr(pp), int(max_steps), int(arith),
                                                      image._h), "webgpu_render_image")

    def post_bloom_image(self, scene, out, **kw):
        p = BloomParams()
        self._lib.grv_bloom_params_default(scene.width, scene.height, C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        self._check(self._lib.grv_post_bloom_image(self._h, C.byref(p), scene._h, out._h), "post_bloom_image")

    def post_taa_resolve_image(self, current, history, out, blend_factor=0.75, camera_moving=False,
                               half_storage=True, arith=ARITH_STRICT):
        p = TaaParams(current.width, current.height, float(blend_factor), int(bool(camera_moving)),
                      int(bool(half_storage)), int(arith))
        self._check(self._lib.grv_post_taa_resolve_image(self._h, C.byref(p), current._h, history._h, out._h),
                    "post_taa_resolve_image")

    def synchronize(self):
        self._check(self._lib.grv_engine_synchronize(self._h), "engine_synchronize")

    def frame_stats(self, stream=None):
        st = FrameStats()
        self._check(self._lib.grv_frame_stats(self._h, C.c_void_p(stream) if stream else None,
                                              C.byref(st)), "frame_stats")
        return st

    def device_bytes(self):
        return int(self._lib.grv_engine_device_bytes(self._h))

    def host_bytes(self):
        return self._lib.grv_engine_host_bytes(self._h)

    def profile_shader_frames(self, enable=True):
        self._check(self._lib.grv_engine_profile_shader_frames(self._h, 1 if enable else 0),
                    "engine_profile_shader_frames")

    def stats_accumulate(self, enable=True):
        """Frames stop clearing the device-side counters: one frame_stats() after a loop of
        frames reads their sums (no host wait inside the loop)."""
        self._check(self._lib.grv_stats_accumulate(self._h, 1 if enable else 0), "stats_accumulate")

    def frame_stats_reset(self, stream=None):
        self._check(self._lib.grv_frame_stats_reset(self._h, C.c_void_p(stream) if stream else None),
                    "frame_stats_reset")

    def unpack_tiles_device(self, params, rank, d_packed, d_image, bytes_per_pixel, stream=None):
        self._check(self._lib.grv_unpack_tiles_device(
            self._h, C.byref(params), int(rank), _dev_ptr(d_packed), _dev_ptr(d_image),
            int(bytes_per_pixel), C.c_void_p(stream) if stream else None), "unpack_tiles_device")

    # ---- lib.rs:128-136 ----
    def generate_spectrum_lut(self, width, height, max_temp):
        out = np.zeros(width * height * 4, np.float32)
        self._check(self._lib.grv_generate_spectrum_lut(self._h, width, height, float(max_temp),
                                                        _np_ptr(out)), "generate_spectrum_lut")
        return out

    def strict_math(self, op, x, y=None):
        """The STRICT contract's sin / cos / pow evaluated on the device (op: MATH_*)."""
        x = np.ascontiguousarray(x, np.float64)
        y = None if y is None else np.ascontiguousarray(y, np.float64)
        out = np.zeros_like(x)
        self._check(self._lib.grv_strict_math(self._h, int(op), x.size, _np_ptr(x), _np_ptr(y),
                                              _np_ptr(out)), "strict_math")
        return out

    def strict_rhs_probe(self, form, states):
        """The STRICT Kerr-Schild right-hand side of [n, 8] states through division form RHS_FORM_*:
        [n, 7] = (dt, dr, dtheta, dphi, dp_r, dp_theta, form that ran)."""
        st = np.ascontiguousarray(states, np.float64).reshape(-1, 8)
        out = np.zeros((st.shape[0], 7))
        self._check(self._lib.grv_strict_rhs_probe(self._h, int(form), st.shape[0], _np_ptr(st), _np_ptr(out)),
                    "strict_rhs_probe")
        return out

    # ---- lib.rs:107-110, 161-205 ----
    def generate_disk_lut(self):
        out = np.zeros(512, np.float32)
        self._check(self._lib.grv_generate_disk_lut(self._h, _np_ptr(out)), "generate_disk_lut")
        return out

    def compute_disk_flux(self, r):
        return self._lib.grv_compute_disk_flux(self._h, float(r))

    def compute_shadow_curve(self, theta_obs, n_points):
        n = self._lib.grv_compute_shadow_curve(self._h, float(theta_obs), int(n_points), None, 0)
        out = np.zeros(2 * n, np.float32)  # size query first: 2n points on axis (shadow.rs:96-113)
        self._lib.grv_compute_shadow_curve(self._h, float(theta_obs), int(n_points), _np_ptr(out),
                                           out.size)
        return out

    def compute_shadow_radius(self):
        return self._lib.grv_compute_shadow_radius(self._h)

    def compute_shadow_shift(self, theta_obs):
        out = np.zeros(2, np.float32)
        self._check(self._lib.grv_compute_shadow_shift(self._h, float(theta_obs), _np_ptr(out)),
                    "compute_shadow_shift")
        return out

    def get_disk_lut_view(self):
        """Float32 view of the engine-owned disk LUT (get_disk_lut_ptr, lib.rs:112)."""
        ptr = self._lib.grv_get_disk_lut_ptr(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(512,))

    # ---- spacetime read-outs: lib.rs:139-159, 214-306 ----
    def compute_kretschner(self, r, theta):
        return self._lib.grv_compute_kretschner(self._h, float(r), float(theta))

    def compute_light_cone_tilt(self, r, theta):
        return self._lib.grv_compute_light_cone_tilt(self._h, float(r), float(theta))

    def compute_frame_drag_omega(self, r, theta):
        return self._lib.grv_compute_frame_drag_omega(self._h, float(r), float(theta))

    def compute_flamm_height(self, r):
        return self._lib.grv_compute_flamm_height(self._h, float(r))

    def compute_proper_distance(self, r1, r2, n_steps):
        return self._lib.grv_compute_proper_distance(self._h, float(r1), float(r2), int(n_steps))

    def _field(self, kind, r_min, r_max, n_radial, n_polar, what):
        out = np.zeros(3 * n_radial * n_polar, np.float32)
        self._check(self._lib.grv_generate_field(self._h, kind, float(r_min), float(r_max),
                                                 int(n_radial), int(n_polar), _np_ptr(out)), what)
        return out

    def generate_curvature_field(self, r_min, r_max, n_radial, n_polar):
        return self._field(0, r_min, r_max, n_radial, n_polar, "generate_curvature_field")

    def generate_tilt_field(self, r_min, r_max, n_radial, n_polar):
        return self._field(1, r_min, r_max, n_radial, n_polar, "generate_tilt_field")

    def generate_frame_drag_field(self, r_min, r_max, n_radial, n_polar):
        return self._field(2, r_min, r_max, n_radial, n_polar, "generate_frame_drag_field")

    def generate_embedding_mesh(self, r_min, r_max, n_radial, n_angular):
        out = np.zeros(3 * n_radial * n_angular, np.float32)
        self._check(self._lib.grv_generate_embedding_mesh(
            self._h, float(r_min), float(r_max), int(n_radial), int(n_angular), _np_ptr(out)),
            "generate_embedding_mesh")
        return out

    def generate_ergosphere_mesh(self, n_polar, n_azimuthal):
        out = np.zeros(3 * n_polar * n_azimuthal, np.float32)
        self._check(self._lib.grv_generate_ergosphere_mesh(
            self._h, int(n_polar), int(n_azimuthal), _np_ptr(out)), "generate_ergosphere_mesh")
        return out

    # ---- SAB protocol: lib.rs:74, 116-126, 308-409 ----
    def attach_sab(self, array):
        """`array`: C-contiguous float32 numpy array of >= 2048 elements kept alive by the caller."""
        self._sab_keepalive = array
        self._check(self._lib.grv_attach_sab(self._h, _np_ptr(array)), "attach_sab")

    def set_camera_state(self, px, py, pz, lx=0.0, ly=0.0, lz=0.0):
        self._lib.grv_set_camera_state(self._h, float(px), float(py), float(pz))

    def set_auto_spin(self, enabled):
        self._lib.grv_set_auto_spin(self._h, 1 if enabled else 0)

    def tick_sab(self, dt_override):
        self._check(self._lib.grv_tick_sab(self._h, float(dt_override)), "tick_sab")

    def sab_view(self):
        """Float32 view of the engine-owned 2048-float block (get_sab_ptr, lib.rs:116)."""
        ptr = self._lib.grv_get_sab_ptr(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(2048,))

    # ---- lib.rs:411-419 ----
    def get_sab_layout(self):
        a = (C.c_size_t * 5)()
        self._lib.grv_get_sab_layout(a)
        return list(a)


TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_PEER_COPY = 0, 1, 2
EXCHANGE_RGBA32F, EXCHANGE_RGBA16F = 0, 1
TEST_HOOKS_KEY = 0x47525654


def unlock_test_hooks():
    """Verification hooks (grv_test_set_try_bound, grv_multi_test_self_exchange) answer only after this
    call: tests make it, product hosts never do."""
    rc = load_library().grv_test_hooks_unlock(TEST_HOOKS_KEY)
    if rc != 0:
        raise GravitasError("grv_test_hooks_unlock refused the key")


class MultiEngine:
    """The image plane across the GPUs of one node through the C ABI (grv_engine_create_multi):
    one process, one host thread and two streams per device, one gather of finished tiles per
    frame (RCCL over xGMI, or peer copies).  `virtual_ranks=G` puts G ranks on ONE device."""

    def __init__(self, mass, spin, devices=None, transport=TRANSPORT_AUTO, virtual_ranks=0, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        if virtual_ranks:
            rc = self._lib.grv_engine_create_multi_virtual(float(mass), float(spin), int(device),
                                                           int(virtual_ranks), C.byref(h))
        else:
            mask = 0
            for dv in (devices if devices is not None else [0]):
                mask |= 1 << int(dv)
            rc = self._lib.grv_engine_create_multi(float(mass), float(spin), mask, int(transport), C.byref(h))
        if rc != 0:
            why = self._lib.grv_multi_create_error()
            raise GravitasError("grv_engine_create_multi: %s: %s (no CPU fallback, and no silent change of "
                                "transport, exists)" % (_STATUS.get(rc, rc), why.decode() if why else ""))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.grv_multi_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.grv_multi_last_error(self._h)
            raise GravitasError("%s: %s: %s" % (what, _STATUS.get(rc, rc), msg.decode() if msg else ""))

    @property
    def ranks(self):
        return self._lib.grv_multi_rank_count(self._h)

    @property
    def transport(self):
        return self._lib.grv_multi_transport(self._h)

    def rank_devices(self):
        return [self._lib.grv_multi_rank_device(self._h, r) for r in range(self.ranks)]

    def update_params(self, mass, spin):
        self._check(self._lib.grv_multi_update_params(self._h, float(mass), float(spin)), "update_params")

    def render_frame_device(self, camera, params, rgba, stream=None):
        self._check(self._lib.grv_render_frame_multi_device(
            self._h, C.byref(camera), C.byref(params), _dev_ptr(rgba),
            C.c_void_p(stream) if stream else None), "render_frame_multi_device")

    def render_frame_wgsl_device(self, params, rgba, stream=None):
        self._check(self._lib.grv_render_frame_wgsl_multi_device(
            self._h, C.byref(params), _dev_ptr(rgba), C.c_void_p(stream) if stream else None),
            "render_frame_wgsl_multi_device")

    def render_frame(self, camera, params):
        rgba = np.zeros((params.height, params.width, 4), np.float32)
        st = FrameStats()
        self._check(self._lib.grv_render_frame_multi(self._h, C.byref(camera), C.byref(params),
                                                     _np_ptr(rgba), C.byref(st)), "render_frame_multi")
        return rgba, st

    def synchronize(self):
        self._check(self._lib.grv_multi_synchronize(self._h), "multi_synchronize")

    def stats_accumulate(self, enable=True):
        self._check(self._lib.grv_multi_stats_accumulate(self._h, 1 if enable else 0), "multi_stats_accumulate")

    def frame_stats_reset(self):
        self._check(self._lib.grv_multi_frame_stats_reset(self._h), "multi_frame_stats_reset")

    def frame_stats(self):
        st = FrameStats()
        self._check(self._lib.grv_multi_frame_stats(self._h, C.byref(st)), "multi_frame_stats")
        return st

    def rank_frame_stats(self, rank):
        st = FrameStats()
        self._check(self._lib.grv_multi_rank_frame_stats(self._h, int(rank), C.byref(st)), "multi_rank_frame_stats")
        return st

    def profile_shader_frames(self, enable=True):
        """f32 march frames of every rank take HIP events on the rank's launch stream
        (grv_engine_profile_shader_frames on each rank's engine)."""
        for r in range(self.ranks):
            eng = self._lib.grv_multi_engine(self._h, r)
            self._check(self._lib.grv_engine_profile_shader_frames(C.c_void_p(eng), 1 if enable else 0),
                        "engine_profile_shader_frames")

    def set_exchange_format(self, fmt):
        """EXCHANGE_RGBA32F (default) or EXCHANGE_RGBA16F: the compute pass's own rgba16float output
        format on the wire, half the bytes of the one exchange."""
        self._check(self._lib.grv_multi_set_exchange_format(self._h, int(fmt)), "multi_set_exchange_format")

    @property
    def exchange_format(self):
        return self._lib.grv_multi_exchange_format(self._h)

    def exchange_bytes_per_frame(self, width, height):
        return self._lib.grv_multi_exchange_bytes_per_frame(self._h, int(width), int(height))

    def test_self_exchange(self, enable=True):
        """Verification hook: rank 0's own share goes through the transport too.  Locked until the process has
        called unlock_test_hooks() (tests do, explicitly; this wrapper never unlocks on a caller's behalf)."""
        self._check(self._lib.grv_multi_test_self_exchange(self._h, 1 if enable else 0), "multi_test_self_exchange")

    def test_inject_fault(self, kind, rank):
        """Verification hook (locked like test_self_exchange): the next frame fails on `rank` at FAULT_RENDER /
        FAULT_SEND (RCCL) / FAULT_PEER_COPY."""
        self._check(self._lib.grv_multi_test_inject_fault(self._h, int(kind), int(rank)), "multi_test_inject_fault")
