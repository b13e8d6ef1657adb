/*
 * gravitas_napi.c -- thin N-API addon over the C ABI of include/gravitas_abi.h.
 *
 * Presents the surface wasm-pack generates for the reference's
 * `#[wasm_bindgen] impl PhysicsEngine`
 * (physics-engine/gravitas-wasm/src/lib.rs:56-465; consumers
 * src/engine/physics-bridge.ts, src/workers/physics.worker.ts):
 *   default export  init()  -> Promise<{memory: {buffer: ArrayBuffer}}>
 *   class PhysicsEngine(mass, spin) with the wasm-bindgen method names
 *   init_hooks()
 * `memory.buffer` + `get_sab_ptr()` are emulated with one module-wide arena:
 * each engine's 2048-float SAB block (lib.rs:67) lives inside it and
 * get_sab_ptr() returns its byte offset, so
 *   new Float32Array(wasm.memory.buffer, engine.get_sab_ptr(), 2048)
 * (physics.worker.ts:61-68) works unchanged.
 *
 * Plain C, N-API v3 (works on Node >= 10; Bun implements the same API).
 * Build: make -C napi   (gcc against /usr/include/node/node_api.h)
 */
#include <node_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gravitas_abi.h"

#define ARENA_BYTES (1u << 20)
#define SAB_BYTES (2048u * 4u)
#define LUT_BYTES (512u * 4u)

#define SLOT_BYTES (SAB_BYTES + LUT_BYTES)
#define N_SLOTS (ARENA_BYTES / SLOT_BYTES)

static uint8_t *g_arena = NULL;
static uint8_t g_slot_used[N_SLOTS]; /* one SAB + LUT region per live engine, reused after free() / GC */
static napi_ref g_arena_ref = NULL;

typedef struct {
    grv_engine *h;
    int slot;       /* arena slot, -1 once released */
    size_t sab_off;
    size_t lut_off; /* 512-float disk LUT copy (get_disk_lut_ptr, lib.rs:112) */
    double mass, spin; /* as last given (constructor / update_params): what the lazily created handles below start from */
    /* renderFrame({devices: n}) / ({virtualRanks: n}): the multi-GPU handle of the C ABI, created on
     * first use and re-created when n changes */
    grv_multi *multi;
    int multi_ranks, multi_virtual;
    /* the *Async methods run on the libuv pool.  A handle is not thread-safe, so they get handles of
     * their own (never touched by the synchronous methods) and one mutex serialises them. */
    grv_engine *async_h;
    grv_multi *async_multi;
    int async_multi_ranks, async_multi_virtual;
    pthread_mutex_t async_mu;
    int async_pending; /* works queued or running (main thread only) */
    int freed;         /* free() was called while works were pending: the last one to complete cleans up */
    /* renderFrameAsync({out}): pinned staging images, kept across calls (an async frame loop holds one or
     * two frames in flight; allocating / freeing page-locked memory per frame costs milliseconds and
     * serialises on the driver).  Taken and returned on the main thread only. */
    struct { float *p; size_t bytes; int busy; } stage[2];
    /* renderFrameAsync on one device: the frame is rendered into one of two device images of the async
     * handle (one compute stream) and copied out on that image's copy stream, so frame i's D2H runs under
     * frame i+1's kernels
     * (guarded by async_mu; users = works between queueing on the image and the end of their wait) */
    grv_image *async_img[2];
    int async_img_users[2];
    int async_turn;
} engine_box;

/* ---- page-locked host blocks handed to JS by allocPinned() ----
 * Kept in a registry so that (1) an `out` array that lives inside one is recognised: the DMA of an async
 * frame / image read then lands in it directly, no staging copy on the JS thread; (2) a block is freed
 * only when its ArrayBuffer has been collected AND no queued work still writes into it -- detaching or
 * transferring the buffer while a frame is in flight can therefore never free memory under the DMA. */
typedef struct pinned_block {
    uint8_t *p;
    size_t bytes;
    int refs; /* queued works writing into it */
    int dead; /* its ArrayBuffer was collected */
    struct pinned_block *next;
} pinned_block;
static pinned_block *g_pinned = NULL;
static pthread_mutex_t g_pinned_mu = PTHREAD_MUTEX_INITIALIZER;

static void pinned_drop_locked(pinned_block *blk) {
    for (pinned_block **q = &g_pinned; *q; q = &(*q)->next)
        if (*q == blk) {
            *q = blk->next;
            break;
        }
    grv_host_free(blk->p);
    free(blk);
}
/* the registry block that holds [ptr, ptr + bytes), with a reference taken; NULL if there is none */
static pinned_block *pinned_acquire(const void *ptr, size_t bytes) {
    pinned_block *hit = NULL;
    pthread_mutex_lock(&g_pinned_mu);
    for (pinned_block *q = g_pinned; q; q = q->next)
        if (!q->dead && (const uint8_t *)ptr >= q->p && (const uint8_t *)ptr + bytes <= q->p + q->bytes) {
            q->refs++;
            hit = q;
            break;
        }
    pthread_mutex_unlock(&g_pinned_mu);
    return hit;
}
static void pinned_release(pinned_block *blk) {
    if (!blk) return;
    pthread_mutex_lock(&g_pinned_mu);
    if (--blk->refs == 0 && blk->dead) pinned_drop_locked(blk);
    pthread_mutex_unlock(&g_pinned_mu);
}

/* a pinned staging image of at least `bytes`: one of the box's two cached buffers (grown on demand), or --
 * with both in flight -- a one-off allocation (*idx = -1) */
static float *stage_acquire(engine_box *b, size_t bytes, int *idx) {
    for (int k = 0; k < 2; k++)
        if (!b->stage[k].busy && b->stage[k].p && b->stage[k].bytes >= bytes) {
            b->stage[k].busy = 1;
            *idx = k;
            return b->stage[k].p;
        }
    for (int k = 0; k < 2; k++)
        if (!b->stage[k].busy) {
            if (b->stage[k].p) grv_host_free(b->stage[k].p);
            b->stage[k].p = (float *)grv_host_alloc(bytes);
            b->stage[k].bytes = b->stage[k].p ? bytes : 0;
            if (!b->stage[k].p) return NULL;
            b->stage[k].busy = 1;
            *idx = k;
            return b->stage[k].p;
        }
    *idx = -1;
    return (float *)grv_host_alloc(bytes);
}
static void stage_release(engine_box *b, float *p, int idx) {
    if (!p) return;
    if (idx >= 0 && b) b->stage[idx].busy = 0;
    else grv_host_free(p);
}

static int slot_acquire(void) {
    for (unsigned k = 0; k < N_SLOTS; ++k)
        if (!g_slot_used[k]) {
            g_slot_used[k] = 1;
            return (int)k;
        }
    return -1;
}
static void slot_release(engine_box *box) {
    if (box->slot >= 0) {
        g_slot_used[box->slot] = 0;
        /* the slot's own region (sab_off may have been re-pointed by attach_sab): the next owner starts from zeros */
        if (g_arena) memset(g_arena + (size_t)box->slot * SLOT_BYTES, 0, SLOT_BYTES);
        box->slot = -1;
    }
}

#define NAPI_OK(call)                                                        \
    do {                                                                     \
        if ((call) != napi_ok) {                                             \
            napi_throw_error(env, NULL, "N-API call failed: " #call);        \
            return NULL;                                                     \
        }                                                                    \
    } while (0)

static napi_value get_arena(napi_env env) {
    napi_value buf;
    if (g_arena_ref) {
        if (napi_get_reference_value(env, g_arena_ref, &buf) == napi_ok && buf) return buf;
    }
    if (!g_arena) g_arena = (uint8_t *)calloc(1, ARENA_BYTES);
    if (napi_create_external_arraybuffer(env, g_arena, ARENA_BYTES, NULL, NULL, &buf) != napi_ok)
        return NULL;
    napi_create_reference(env, buf, 1, &g_arena_ref);
    return buf;
}

static napi_ref g_engine_ctor = NULL;
/* the engine_box behind `self`, or NULL: only for objects made by the PhysicsEngine constructor (a method borrowed
 * onto a DeviceImage would otherwise unwrap that object's box as an engine's) */
static engine_box *engine_of(napi_env env, napi_value self) {
    engine_box *box = NULL;
    napi_value ctor;
    bool is = false;
    if (!g_engine_ctor || napi_get_reference_value(env, g_engine_ctor, &ctor) != napi_ok ||
        napi_instanceof(env, self, ctor, &is) != napi_ok || !is) return NULL;
    if (napi_unwrap(env, self, (void **)&box) != napi_ok) return NULL;
    return box;
}

static engine_box *unwrap(napi_env env, napi_callback_info info, size_t *argc, napi_value *argv) {
    napi_value self;
    engine_box *box = NULL;
    if (napi_get_cb_info(env, info, argc, argv, &self, NULL) != napi_ok) return NULL;
    box = engine_of(env, self);
    if (!box || !box->h) {
        napi_throw_error(env, NULL, "PhysicsEngine: invalid receiver");
        return NULL;
    }
    return box;
}

static double arg_f64(napi_env env, napi_value v) {
    double d = 0.0;
    napi_get_value_double(env, v, &d);
    return d;
}

static napi_value mk_f64(napi_env env, double d) {
    napi_value v;
    napi_create_double(env, d, &v);
    return v;
}

static void box_destroy_handles(engine_box *box) {
    for (int k = 0; k < 2; k++) { /* (no work is pending here: nobody waits on them) */
        if (box->async_img[k]) grv_image_destroy(box->async_img[k]);
        box->async_img[k] = NULL;
        box->async_img_users[k] = 0;
    }
    if (box->multi) grv_multi_destroy(box->multi);
    if (box->async_multi) grv_multi_destroy(box->async_multi);
    if (box->async_h) grv_engine_destroy(box->async_h);
    if (box->h) grv_engine_destroy(box->h);
    box->multi = box->async_multi = NULL;
    box->async_h = box->h = NULL;
    for (int k = 0; k < 2; k++) { /* (no work is pending here: every work released its staging before this runs) */
        if (box->stage[k].p) grv_host_free(box->stage[k].p);
        box->stage[k].p = NULL;
        box->stage[k].bytes = 0;
        box->stage[k].busy = 0;
    }
}

static void engine_finalize(napi_env env, void *data, void *hint) {
    (void)env;
    (void)hint;
    engine_box *box = (engine_box *)data;
    if (box) {
        /* a pending async work holds a reference to the JS object, so none is pending here */
        box_destroy_handles(box);
        slot_release(box);
        pthread_mutex_destroy(&box->async_mu);
        free(box);
    }
}

/* new PhysicsEngine(mass, spin)  -- lib.rs:59 */
static napi_value engine_new(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2], self;
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, NULL));
    double mass = argc > 0 ? arg_f64(env, argv[0]) : 1.0;
    double spin = argc > 1 ? arg_f64(env, argv[1]) : 0.0;
    engine_box *box = (engine_box *)calloc(1, sizeof *box);
    if (!box) {
        napi_throw_error(env, NULL, "PhysicsEngine: out of memory");
        return NULL;
    }
    box->mass = mass;
    box->spin = spin;
    pthread_mutex_init(&box->async_mu, NULL);
    int rc = grv_engine_create(mass, spin, 0, &box->h);
    if (rc != GRV_OK) {
        pthread_mutex_destroy(&box->async_mu);
        free(box);
        napi_throw_error(env, NULL, rc == GRV_ERR_NO_DEVICE
                                        ? "PhysicsEngine: no HIP device (this engine has no CPU path)"
                                        : "PhysicsEngine: grv_engine_create failed");
        return NULL;
    }
    (void)get_arena(env);
    box->slot = g_arena ? slot_acquire() : -1;
    if (box->slot < 0) { /* no silent fallback onto another engine's region */
        grv_engine_destroy(box->h);
        pthread_mutex_destroy(&box->async_mu);
        free(box);
        napi_throw_error(env, NULL, "PhysicsEngine: memory arena exhausted (free() engines no longer in use)");
        return NULL;
    }
    box->sab_off = (size_t)box->slot * SLOT_BYTES;
    box->lut_off = box->sab_off + SAB_BYTES;
    grv_attach_sab(box->h, (float *)(g_arena + box->sab_off)); /* attach_sab lib.rs:74 */
    NAPI_OK(napi_wrap(env, self, box, engine_finalize, NULL, NULL));
    return self;
}

static napi_value m_update_params(napi_env env, napi_callback_info info) { /* lib.rs:78 */
    size_t argc = 2;
    napi_value argv[2];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    b->mass = arg_f64(env, argv[0]);
    b->spin = arg_f64(env, argv[1]);
    grv_update_params(b->h, b->mass, b->spin);
    if (b->multi) grv_multi_update_params(b->multi, b->mass, b->spin);
    /* the async handles take (mass, spin) with every work item */
    return NULL;
}

#define SCALAR0(name, fn)                                                      \
    static napi_value name(napi_env env, napi_callback_info info) {            \
        size_t argc = 0;                                                       \
        engine_box *b = unwrap(env, info, &argc, NULL);                        \
        return b ? mk_f64(env, fn(b->h)) : NULL;                               \
    }
SCALAR0(m_compute_horizon, grv_compute_horizon)             /* lib.rs:85 */
SCALAR0(m_compute_isco, grv_compute_isco)                   /* lib.rs:89 */
SCALAR0(m_compute_photon_sphere, grv_compute_photon_sphere) /* lib.rs:93 */

static napi_value m_compute_dilation(napi_env env, napi_callback_info info) { /* lib.rs:97 */
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    return b ? mk_f64(env, grv_compute_dilation(b->h, arg_f64(env, argv[0]))) : NULL;
}

static napi_value m_compute_g_factor(napi_env env, napi_callback_info info) { /* lib.rs:203 */
    size_t argc = 2;
    napi_value argv[2];
    engine_box *b = unwrap(env, info, &argc, argv);
    return b ? mk_f64(env, grv_compute_g_factor(b->h, arg_f64(env, argv[0]), arg_f64(env, argv[1])))
             : NULL;
}

/* integrate_ray_relativistic(Float64Array|number[], steps, tolerance, useKS) -> Float64Array
 * lib.rs:422-464 */
static napi_value m_integrate_ray(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    double in[64];
    size_t n = 0;
    bool is_ta = false;
    napi_is_typedarray(env, argv[0], &is_ta);
    if (is_ta) {
        napi_typedarray_type ty;
        size_t len;
        void *data;
        NAPI_OK(napi_get_typedarray_info(env, argv[0], &ty, &len, &data, NULL, NULL));
        if (ty != napi_float64_array) {
            napi_throw_type_error(env, NULL, "initial_state must be a Float64Array");
            return NULL;
        }
        n = len > 64 ? 64 : len; /* only [0, 8) is read once len >= 8 (lib.rs:429-442); shorter inputs are echoed */
        memcpy(in, data, n * sizeof(double));
    } else {
        uint32_t len = 0;
        NAPI_OK(napi_get_array_length(env, argv[0], &len));
        n = len > 64 ? 64 : len;
        for (uint32_t i = 0; i < n; i++) {
            napi_value e;
            napi_get_element(env, argv[0], i, &e);
            in[i] = arg_f64(env, e);
        }
    }
    uint32_t steps = 0;
    napi_get_value_uint32(env, argv[1], &steps);
    double tol = arg_f64(env, argv[2]);
    bool ks = false;
    napi_get_value_bool(env, argv[3], &ks);
    double out[64];
    size_t m = grv_integrate_ray_relativistic(b->h, in, n, steps, tol, ks ? 1 : 0, out);
    napi_value ab, ta;
    void *dst;
    NAPI_OK(napi_create_arraybuffer(env, m * sizeof(double), &dst, &ab));
    memcpy(dst, out, m * sizeof(double));
    NAPI_OK(napi_create_typedarray(env, napi_float64_array, m, ab, 0, &ta));
    return ta;
}

/* generate_spectrum_lut(w, h, maxTemp) -> Float32Array[w*h*4]   lib.rs:128-136 */
static napi_value m_generate_spectrum_lut(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    uint32_t w = 0, h = 0;
    napi_get_value_uint32(env, argv[0], &w);
    napi_get_value_uint32(env, argv[1], &h);
    double tmax = arg_f64(env, argv[2]);
    size_t n = (size_t)w * h * 4;
    napi_value ab, ta;
    void *dst;
    NAPI_OK(napi_create_arraybuffer(env, n * sizeof(float), &dst, &ab));
    if (n && grv_generate_spectrum_lut(b->h, w, h, tmax, (float *)dst) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, n, ab, 0, &ta));
    return ta;
}

/* generate_disk_lut() -> Float32Array[512]   lib.rs:107-110 */
static napi_value m_generate_disk_lut(napi_env env, napi_callback_info info) {
    size_t argc = 0;
    engine_box *b = unwrap(env, info, &argc, NULL);
    if (!b) return NULL;
    napi_value ab, ta;
    void *dst;
    NAPI_OK(napi_create_arraybuffer(env, 512 * sizeof(float), &dst, &ab));
    grv_generate_disk_lut(b->h, (float *)dst);
    if (g_arena && b->slot >= 0) memcpy(g_arena + b->lut_off, dst, LUT_BYTES); /* self.lut_buffer */
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, 512, ab, 0, &ta));
    return ta;
}

/* compute_shadow_curve(theta_obs, n) -> Float32Array[2n]   lib.rs:161-169 */
static napi_value m_compute_shadow_curve(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    uint32_t n = 0;
    napi_get_value_uint32(env, argv[1], &n);
    if (n > 4096) n = 4096;
    const double theta = arg_f64(env, argv[0]);
    /* size query first: n points off axis, 2n for an on-axis observer (shadow.rs:96-113) */
    const size_t m = grv_compute_shadow_curve(b->h, theta, n, NULL, 0);
    napi_value ab, ta;
    void *dst;
    NAPI_OK(napi_create_arraybuffer(env, 2 * m * sizeof(float), &dst, &ab));
    grv_compute_shadow_curve(b->h, theta, n, (float *)dst, 2 * m);
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, 2 * m, ab, 0, &ta));
    return ta;
}

static napi_value m_compute_shadow_radius(napi_env env, napi_callback_info info) { /* lib.rs:172 */
    size_t argc = 0;
    engine_box *b = unwrap(env, info, &argc, NULL);
    return b ? mk_f64(env, grv_compute_shadow_radius(b->h)) : NULL;
}

static napi_value m_compute_disk_flux(napi_env env, napi_callback_info info) { /* lib.rs:198 */
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    return b ? mk_f64(env, grv_compute_disk_flux(b->h, arg_f64(env, argv[0]))) : NULL;
}

static napi_value m_set_camera_state(napi_env env, napi_callback_info info) { /* lib.rs:120 */
    size_t argc = 6;
    napi_value argv[6];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    grv_set_camera_state(b->h, arg_f64(env, argv[0]), arg_f64(env, argv[1]), arg_f64(env, argv[2]));
    return NULL;
}

static napi_value m_set_auto_spin(napi_env env, napi_callback_info info) { /* lib.rs:124 */
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    bool on = false;
    napi_get_value_bool(env, argv[0], &on);
    grv_set_auto_spin(b->h, on ? 1 : 0);
    return NULL;
}

/* set_ray_arith("strict" | "fast") -- an extension (gravitas-wasm has one arithmetic): the contract of
 * integrate_ray_relativistic / integratePhotonGeodesic on this engine.  "strict", the default, returns the
 * reference-order bits; "fast" the same geodesic to rounding (<= 1e-5 relative, median <= 1e-9) in a
 * third of the time (grv_engine_set_ray_arith). */
static napi_value m_set_ray_arith(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    char name[16] = {0};
    size_t len = 0;
    if (argc < 1 || napi_get_value_string_utf8(env, argv[0], name, sizeof name, &len) != napi_ok ||
        (strcmp(name, "strict") != 0 && strcmp(name, "fast") != 0)) {
        napi_throw_type_error(env, NULL, "set_ray_arith: expected \"strict\" or \"fast\"");
        return NULL;
    }
    if (grv_engine_set_ray_arith(b->h, strcmp(name, "fast") == 0 ? GRV_ARITH_FAST : GRV_ARITH_STRICT) != GRV_OK)
        napi_throw_error(env, NULL, "set_ray_arith failed");
    return NULL;
}

static napi_value m_tick_sab(napi_env env, napi_callback_info info) { /* lib.rs:308 */
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    grv_tick_sab(b->h, argc > 0 ? arg_f64(env, argv[0]) : 0.0);
    return NULL;
}

static napi_value m_get_sab_ptr(napi_env env, napi_callback_info info) { /* lib.rs:116 */
    size_t argc = 0;
    engine_box *b = unwrap(env, info, &argc, NULL);
    if (!b) return NULL;
    napi_value v;
    napi_create_uint32(env, (uint32_t)b->sab_off, &v);
    return v;
}

/* attach_sab(ptr) lib.rs:74: the reference stores a raw `*mut f32` into the module's linear memory and
 * tick_sab writes through it from then on.  Here `memory.buffer` is the arena, so a pointer is a byte
 * offset into it: an offset that is 4-byte aligned and leaves room for the 2048-float block becomes the
 * block tick_sab reads its controls from and publishes into (lib.rs:309-313; get_sab_ptr still names the
 * engine's own block, lib.rs:116-118); anything else -- a raw pointer has no other meaning in JS --
 * throws a RangeError instead of being dereferenced. */
static napi_value m_attach_sab(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    const double ptr = argc > 0 ? arg_f64(env, argv[0]) : -1.0;
    if (!g_arena || !(ptr >= 0.0) || ptr != (double)(uint32_t)ptr || ((uint32_t)ptr & 3u) ||
        (uint32_t)ptr > ARENA_BYTES - SAB_BYTES) {
        napi_throw_range_error(env, NULL, "attach_sab: not a 4-byte aligned offset into memory.buffer with room for 2048 floats");
        return NULL;
    }
    if (grv_attach_sab(b->h, (float *)(g_arena + (uint32_t)ptr)) != GRV_OK) {
        napi_throw_error(env, NULL, "attach_sab failed");
        return NULL;
    }
    return NULL; /* get_sab_ptr keeps naming the engine's own block, as lib.rs:116-118 does */
}

static napi_value m_get_sab_layout(napi_env env, napi_callback_info info) { /* lib.rs:411 */
    (void)info;
    size_t off[5];
    grv_get_sab_layout(off);
    napi_value arr;
    NAPI_OK(napi_create_array_with_length(env, 5, &arr));
    for (uint32_t i = 0; i < 5; i++) {
        napi_value v;
        napi_create_uint32(env, (uint32_t)off[i], &v);
        napi_set_element(env, arr, i, v);
    }
    return arr;
}


/* ---- remaining wasm-bindgen methods: lib.rs:112-114, 139-159, 178-195, 214-306 ---- */
static napi_value f32_array(napi_env env, const float *src, size_t n) {
    napi_value ab, ta;
    void *dst;
    if (napi_create_arraybuffer(env, n * sizeof(float), &dst, &ab) != napi_ok) return NULL;
    if (n && src) memcpy(dst, src, n * sizeof(float));
    if (napi_create_typedarray(env, napi_float32_array, n, ab, 0, &ta) != napi_ok) return NULL;
    return ta;
}

static uint32_t arg_u32(napi_env env, napi_value v) {
    uint32_t u = 0;
    napi_get_value_uint32(env, v, &u);
    return u;
}

static napi_value m_compute_shadow_shift(napi_env env, napi_callback_info info) { /* lib.rs:178 */
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    float mm[2] = {0.0f, 0.0f};
    grv_compute_shadow_shift(b->h, arg_f64(env, argv[0]), mm);
    return f32_array(env, mm, 2);
}

#define SCALAR2(name, fn)                                                                  \
    static napi_value name(napi_env env, napi_callback_info info) {                        \
        size_t argc = 2;                                                                   \
        napi_value argv[2];                                                                \
        engine_box *b = unwrap(env, info, &argc, argv);                                    \
        return b ? mk_f64(env, fn(b->h, arg_f64(env, argv[0]), arg_f64(env, argv[1]))) : NULL; \
    }
SCALAR2(m_compute_kretschner, grv_compute_kretschner)             /* lib.rs:214 */
SCALAR2(m_compute_light_cone_tilt, grv_compute_light_cone_tilt)   /* lib.rs:239 */
SCALAR2(m_compute_frame_drag_omega, grv_compute_frame_drag_omega) /* lib.rs:268 */

static napi_value m_compute_flamm_height(napi_env env, napi_callback_info info) { /* lib.rs:297 */
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    return b ? mk_f64(env, grv_compute_flamm_height(b->h, arg_f64(env, argv[0]))) : NULL;
}

static napi_value m_compute_proper_distance(napi_env env, napi_callback_info info) { /* lib.rs:303 */
    size_t argc = 3;
    napi_value argv[3];
    engine_box *b = unwrap(env, info, &argc, argv);
    return b ? mk_f64(env, grv_compute_proper_distance(b->h, arg_f64(env, argv[0]),
                                                       arg_f64(env, argv[1]), arg_u32(env, argv[2])))
             : NULL;
}

/* generate_{curvature,tilt,frame_drag}_field(rMin, rMax, nRadial, nPolar) -> Float32Array[3*nr*np] */
static napi_value field_common(napi_env env, napi_callback_info info, int field) {
    size_t argc = 4;
    napi_value argv[4];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    const uint32_t nr = arg_u32(env, argv[2]), np = arg_u32(env, argv[3]);
    const size_t n = (size_t)3 * nr * np;
    float *tmp = (float *)malloc((n ? n : 1) * sizeof(float));
    if (!tmp) {
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    if (grv_generate_field(b->h, field, arg_f64(env, argv[0]), arg_f64(env, argv[1]), nr, np, tmp) !=
        GRV_OK) {
        free(tmp);
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    napi_value ta = f32_array(env, tmp, n);
    free(tmp);
    return ta;
}
static napi_value m_generate_curvature_field(napi_env env, napi_callback_info info) { /* lib.rs:220 */
    return field_common(env, info, GRV_FIELD_CURVATURE);
}
static napi_value m_generate_tilt_field(napi_env env, napi_callback_info info) { /* lib.rs:245 */
    return field_common(env, info, GRV_FIELD_TILT);
}
static napi_value m_generate_frame_drag_field(napi_env env, napi_callback_info info) { /* lib.rs:274 */
    return field_common(env, info, GRV_FIELD_FRAME_DRAG);
}

/* generate_embedding_mesh(rMin, rMax, nRadial, nAngular)  lib.rs:139 ; physics-bridge.ts:315 */
static napi_value m_generate_embedding_mesh(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    const uint32_t nr = arg_u32(env, argv[2]), na = arg_u32(env, argv[3]);
    const size_t n = (size_t)3 * nr * na;
    float *tmp = (float *)malloc((n ? n : 1) * sizeof(float));
    if (!tmp) {
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    if (grv_generate_embedding_mesh(b->h, arg_f64(env, argv[0]), arg_f64(env, argv[1]), nr, na, tmp) !=
        GRV_OK) {
        free(tmp);
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    napi_value ta = f32_array(env, tmp, n);
    free(tmp);
    return ta;
}

/* generate_ergosphere_mesh(nPolar, nAzimuthal)  lib.rs:153 ; physics-bridge.ts:333 */
static napi_value m_generate_ergosphere_mesh(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    const uint32_t np = arg_u32(env, argv[0]), na = arg_u32(env, argv[1]);
    const size_t n = (size_t)3 * np * na;
    float *tmp = (float *)malloc((n ? n : 1) * sizeof(float));
    if (!tmp) {
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    if (grv_generate_ergosphere_mesh(b->h, np, na, tmp) != GRV_OK) {
        free(tmp);
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    napi_value ta = f32_array(env, tmp, n);
    free(tmp);
    return ta;
}

/* get_disk_lut_ptr() lib.rs:112: byte offset of the engine's 512-float LUT copy in the arena
 * (refreshed by generate_disk_lut) */
static napi_value m_get_disk_lut_ptr(napi_env env, napi_callback_info info) {
    size_t argc = 0;
    engine_box *b = unwrap(env, info, &argc, NULL);
    if (!b) return NULL;
    napi_value v;
    napi_create_uint32(env, (uint32_t)b->lut_off, &v);
    return v;
}

static double obj_f64(napi_env env, napi_value obj, const char *key, double dflt) {
    bool has = false;
    napi_value v;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return dflt;
    if (napi_get_named_property(env, obj, key, &v) != napi_ok) return dflt;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_number) return dflt;
    return arg_f64(env, v);
}

/* a boolean option (true / false, or a number compared with 0) */
static int obj_flag(napi_env env, napi_value obj, const char *key) {
    bool has = false, b = false;
    napi_value v;
    napi_valuetype t;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return 0;
    if (napi_get_named_property(env, obj, key, &v) != napi_ok || napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_boolean) return napi_get_value_bool(env, v, &b) == napi_ok && b;
    if (t == napi_number) return arg_f64(env, v) != 0.0;
    return 0;
}

static void obj_vec3(napi_env env, napi_value obj, const char *key, double out[3]) {
    bool has = false;
    napi_value arr, e;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return;
    if (napi_get_named_property(env, obj, key, &arr) != napi_ok) return;
    for (uint32_t i = 0; i < 3; i++)
        if (napi_get_element(env, arr, i, &e) == napi_ok) out[i] = arg_f64(env, e);
}

/* ---------------------------------------------------------------------------------------------
 * The path's two bulk entries -- frames and batches of geodesics -- in a synchronous and an
 * asynchronous form.  A work item is filled on the JS thread (arguments parsed, output ArrayBuffers
 * created or taken from the caller), executed either in place or on the libuv pool, and turned into
 * the result object on the JS thread again.
 * ------------------------------------------------------------------------------------------- */
static int obj_str(napi_env env, napi_value obj, const char *key, char *buf, size_t cap) {
    bool has = false;
    napi_value v;
    size_t len = 0;
    buf[0] = 0;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return 0;
    if (napi_get_named_property(env, obj, key, &v) != napi_ok) return 0;
    return napi_get_value_string_utf8(env, v, buf, cap, &len) == napi_ok;
}

typedef struct {
    engine_box *box;
    int kind;       /* 0 frame, 1 batch */
    int is_async;
    double mass, spin;
    /* frame */
    GrvCamera cam;
    GrvRenderParams p;
    int devices, virtual_ranks;
    int exchange;   /* GRV_EXCHANGE_*: what the one exchange of a multi-rank frame carries */
    int transport;  /* GRV_TRANSPORT_* the frame's tiles travelled through; 0 = one device, no exchange */
    float *rgba;    /* W*H*4: memory of the result's ArrayBuffer, or own_rgba */
    /* The *Async forms run on a pool thread while JS keeps running: memory JS can reach in the
     * meantime (the caller's input array, a caller-supplied `out`) may be written, transferred or
     * detached under the work.  So the async forms never hold pointers into it: the input is copied
     * into the work item, and a caller's `out` is filled on the JS thread when the work completes,
     * after checking that its buffer is still attached. */
    float *own_rgba;   /* pinned staging of an async frame with a caller-supplied `out` (engine_box.stage / one-off) */
    int own_stage;     /* index into engine_box.stage, -1 = one-off allocation */
    size_t rgba_elems;
    void *own_in;      /* copy of an async batch's input states */
    pinned_block *pin; /* `out` lies in allocPinned() memory: the async frame's DMA lands in it directly (reference held) */
    GrvFrameStats st;
    /* batch */
    size_t n;
    GrvOptions opt;
    double *in, *out_states, *drift;
    uint32_t *steps;
    uint8_t *term;
    size_t max_points;  /* recordPath: rows of `paths` per ray */
    double *paths;      /* [n][max_points][8] */
    uint32_t *counts;   /* [n] */
    /* result */
    int rc;
    char err[256];
    napi_ref keep[7]; /* typed arrays of the result (and the caller's input) kept alive while queued */
    int n_keep;
    napi_ref self_ref;
    napi_deferred deferred;
    napi_async_work work;
} bulk_work;

/* the multi-GPU handle for `ranks` ranks (virtual: all on device 0), cached in *slot */
static int multi_for(grv_multi **slot, int *cur_ranks, int *cur_virtual, int ranks, int virt, double mass,
                     double spin, char *err, size_t cap) {
    if (*slot && (*cur_ranks != ranks || *cur_virtual != virt)) {
        grv_multi_destroy(*slot);
        *slot = NULL;
    }
    if (!*slot) {
        int rc = virt ? grv_engine_create_multi_virtual(mass, spin, 0, ranks, slot)
                      : grv_engine_create_multi(mass, spin, ranks >= 64 ? ~0ull : ((1ull << ranks) - 1ull),
                                                GRV_TRANSPORT_AUTO, slot);
        if (rc != GRV_OK) {
            snprintf(err, cap, "renderFrame: cannot open %d %s (status %d): %s", ranks,
                     virt ? "virtual ranks" : "HIP devices", rc, grv_multi_create_error());
            return rc;
        }
        *cur_ranks = ranks;
        *cur_virtual = virt;
    }
    return grv_multi_update_params(*slot, mass, spin);
}

/* runs on the JS thread (sync form) or on a pool thread (async form) */
static void bulk_execute(bulk_work *w) {
    engine_box *b = w->box;
    grv_engine *h = b->h;
    grv_multi **mslot = &b->multi;
    int *mr = &b->multi_ranks, *mv = &b->multi_virtual;
    if (w->is_async) {
        pthread_mutex_lock(&b->async_mu);
        if (!b->async_h && grv_engine_create(w->mass, w->spin, 0, &b->async_h) != GRV_OK) {
            w->rc = GRV_ERR_NO_DEVICE;
            snprintf(w->err, sizeof w->err, "async: grv_engine_create failed");
            pthread_mutex_unlock(&b->async_mu);
            return;
        }
        h = b->async_h;
        grv_update_params(h, w->mass, w->spin);
        mslot = &b->async_multi;
        mr = &b->async_multi_ranks;
        mv = &b->async_multi_virtual;
    }
    if (w->kind == 0) {
        const int ranks = w->virtual_ranks > 0 ? w->virtual_ranks : w->devices;
        if (ranks > 1 || w->virtual_ranks > 0) {
            w->rc = multi_for(mslot, mr, mv, ranks, w->virtual_ranks > 0, w->mass, w->spin, w->err, sizeof w->err);
            if (w->rc == GRV_OK) w->rc = grv_multi_set_exchange_format(*mslot, w->exchange);
            if (w->rc == GRV_OK) {
                w->transport = grv_multi_transport(*mslot);
                w->rc = grv_render_frame_multi(*mslot, &w->cam, &w->p, w->rgba, &w->st);
                if (w->rc != GRV_OK) snprintf(w->err, sizeof w->err, "%s", grv_multi_last_error(*mslot));
            }
        } else if (w->is_async) {
            /* queue the frame and its D2H on one of the handle's two images, let go of the handle, THEN
             * wait: the next work queues its kernels (other image, other stream) under this frame's copy */
            grv_image *img = NULL;
            int slot = (b->async_turn ^= 1), priv = 0;
            if (b->async_img_users[slot] > 0 && b->async_img_users[slot ^ 1] == 0) slot ^= 1;
            if (b->async_img_users[slot] > 0) {
                priv = 1; /* both images still have a work between its queueing and the end of its read (more than
                             two frames in flight): this one gets an image of its own */
            } else if (b->async_img[slot] && (grv_image_width(b->async_img[slot]) != w->p.width ||
                                              grv_image_height(b->async_img[slot]) != w->p.height)) {
                grv_image_destroy(b->async_img[slot]);
                b->async_img[slot] = NULL;
            }
            if (priv) w->rc = grv_image_create(h, w->p.width, w->p.height, &img);
            else {
                if (!b->async_img[slot]) /* one compute stream for both: frames in queue order, each D2H under the next frame's kernels */
                    w->rc = b->async_img[slot ^ 1] ? grv_image_create_shared(h, w->p.width, w->p.height, b->async_img[slot ^ 1], &b->async_img[slot])
                                                   : grv_image_create(h, w->p.width, w->p.height, &b->async_img[slot]);
                img = b->async_img[slot];
            }
            if (w->rc == GRV_OK) w->rc = grv_render_frame_image(h, &w->cam, &w->p, img);
            if (w->rc != GRV_OK) snprintf(w->err, sizeof w->err, "%s", grv_last_error(h));
            if (!priv && img) b->async_img_users[slot]++;
            pthread_mutex_unlock(&b->async_mu);
            if (img) { /* image-only calls: the next work queues its kernels meanwhile */
                /* waits for the frame on the host, then copies (even after a failed render: kernels may be queued) */
                int rc2 = w->rc == GRV_OK ? grv_image_read(img, w->rgba, (size_t)w->p.width * w->p.height * 4) : grv_image_wait(img);
                if (rc2 == GRV_OK && w->rc == GRV_OK) rc2 = grv_image_frame_stats(img, &w->st);
                if (w->rc == GRV_OK && rc2 != GRV_OK) {
                    w->rc = rc2;
                    snprintf(w->err, sizeof w->err, "%s", grv_image_last_error(img));
                }
                w->st.launches = 1;
            }
            pthread_mutex_lock(&b->async_mu);
            if (priv) grv_image_destroy(img);
            else if (img) b->async_img_users[slot]--;
        } else {
            w->rc = grv_render_frame(h, &w->cam, &w->p, w->rgba, &w->st);
            if (w->rc != GRV_OK) snprintf(w->err, sizeof w->err, "%s", grv_last_error(h));
        }
    } else {
        if (w->opt.record_path)
            w->rc = grv_integrate_paths(h, w->n, w->in, &w->opt, w->max_points, w->paths, w->counts, w->out_states,
                                        w->steps, w->term, w->drift);
        else
            w->rc = grv_integrate_batch(h, w->n, w->in, &w->opt, w->out_states, w->steps, w->term, w->drift);
        if (w->rc != GRV_OK) snprintf(w->err, sizeof w->err, "%s", grv_last_error(h));
    }
    if (w->is_async) pthread_mutex_unlock(&b->async_mu);
}

static napi_value bulk_result(napi_env env, bulk_work *w) {
    napi_value out, v[7];
    for (int k = 0; k < w->n_keep; k++)
        if (napi_get_reference_value(env, w->keep[k], &v[k]) != napi_ok) return NULL;
    if (napi_create_object(env, &out) != napi_ok) return NULL;
    if (w->kind == 0) {
        napi_set_named_property(env, out, "rgba", v[0]);
        napi_set_named_property(env, out, "width", mk_f64(env, w->p.width));
        napi_set_named_property(env, out, "height", mk_f64(env, w->p.height));
        napi_set_named_property(env, out, "rays", mk_f64(env, (double)w->st.rays));
        napi_set_named_property(env, out, "acceptedSteps", mk_f64(env, (double)w->st.accepted_steps));
        napi_set_named_property(env, out, "launches", mk_f64(env, (double)w->st.launches));
        napi_set_named_property(env, out, "devices",
                                mk_f64(env, w->virtual_ranks > 0 ? w->virtual_ranks : (w->devices > 1 ? w->devices : 1)));
        napi_value tr;
        const char *tn = w->transport == GRV_TRANSPORT_RCCL ? "rccl" : w->transport == GRV_TRANSPORT_PEER_COPY ? "peer_copy" : "none";
        if (napi_create_string_utf8(env, tn, NAPI_AUTO_LENGTH, &tr) == napi_ok) napi_set_named_property(env, out, "transport", tr);
    } else {
        napi_set_named_property(env, out, "states", v[0]);
        napi_set_named_property(env, out, "steps", v[1]);
        napi_set_named_property(env, out, "term", v[2]);
        napi_set_named_property(env, out, "drift", v[3]);
        if (w->opt.record_path) { /* Trajectory.path (mod.rs:160): row i holds min(counts[i], maxPoints) states */
            napi_set_named_property(env, out, "paths", v[4]);
            napi_set_named_property(env, out, "counts", v[5]);
            napi_set_named_property(env, out, "maxPoints", mk_f64(env, (double)w->max_points));
        }
    }
    return out;
}

static void bulk_release(napi_env env, bulk_work *w) {
    stage_release(w->box, w->own_rgba, w->own_stage);
    pinned_release(w->pin);
    free(w->own_in);
    for (int k = 0; k < w->n_keep; k++) napi_delete_reference(env, w->keep[k]);
    if (w->self_ref) napi_delete_reference(env, w->self_ref);
    free(w);
}

static void bulk_async_execute(napi_env env, void *data) {
    (void)env;
    bulk_execute((bulk_work *)data);
}

static void bulk_async_complete(napi_env env, napi_status status, void *data) {
    bulk_work *w = (bulk_work *)data;
    engine_box *b = w->box;
    napi_value res = NULL;
    if (status == napi_ok && w->rc == GRV_OK && w->pin) {
        /* the DMA went straight into allocPinned() memory (kept alive by the registry reference); an `out`
         * that was detached or transferred meanwhile is reported the same way as in the staged form */
        napi_value ta;
        napi_typedarray_type ty;
        size_t len = 0;
        void *data = NULL;
        if (!(napi_get_reference_value(env, w->keep[0], &ta) == napi_ok && ta &&
              napi_get_typedarray_info(env, ta, &ty, &len, &data, NULL, NULL) == napi_ok && data && len >= w->rgba_elems)) {
            w->rc = GRV_ERR_INVALID;
            snprintf(w->err, sizeof w->err, "renderFrameAsync: `out` was detached or transferred while the frame was queued");
        }
    }
    if (status == napi_ok && w->rc == GRV_OK && w->own_rgba) {
        /* caller-supplied `out`: look at it again NOW -- a buffer transferred or detached while the work
         * was queued reports no data / a shorter length, and the promise rejects instead of writing freed memory */
        napi_value ta;
        napi_typedarray_type ty;
        size_t len = 0;
        void *data = NULL;
        if (napi_get_reference_value(env, w->keep[0], &ta) == napi_ok && ta &&
            napi_get_typedarray_info(env, ta, &ty, &len, &data, NULL, NULL) == napi_ok && data && len >= w->rgba_elems) {
            memcpy(data, w->own_rgba, w->rgba_elems * sizeof(float));
        } else {
            w->rc = GRV_ERR_INVALID;
            snprintf(w->err, sizeof w->err, "renderFrameAsync: `out` was detached or transferred while the frame was queued");
        }
    }
    if (status == napi_ok && w->rc == GRV_OK) res = bulk_result(env, w);
    if (res) {
        napi_resolve_deferred(env, w->deferred, res);
    } else {
        napi_value msg, err;
        napi_create_string_utf8(env, w->err[0] ? w->err : "async work failed", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, w->deferred, err);
    }
    napi_delete_async_work(env, w->work);
    b->async_pending--;
    const int last = b->freed && b->async_pending == 0;
    bulk_release(env, w); /* hands the work's staging image back first: box_destroy_handles frees only idle ones */
    if (last) box_destroy_handles(b); /* free() came while works were queued */
}

/* queue (async) or run (sync) a filled work item; returns the promise / the result object */
static napi_value bulk_dispatch(napi_env env, napi_value self, bulk_work *w) {
    if (!w->is_async) {
        bulk_execute(w);
        napi_value res = w->rc == GRV_OK ? bulk_result(env, w) : NULL;
        if (!res) napi_throw_error(env, NULL, w->err[0] ? w->err : "call failed");
        bulk_release(env, w);
        return res;
    }
    napi_value promise, name;
    if (napi_create_promise(env, &w->deferred, &promise) != napi_ok ||
        napi_create_reference(env, self, 1, &w->self_ref) != napi_ok || /* the engine outlives its queued works */
        napi_create_string_utf8(env, "gravitas.bulk", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, NULL, name, bulk_async_execute, bulk_async_complete, w, &w->work) != napi_ok ||
        napi_queue_async_work(env, w->work) != napi_ok) {
        napi_throw_error(env, NULL, "cannot queue async work");
        bulk_release(env, w);
        return NULL;
    }
    w->box->async_pending++;
    return promise;
}

static int keep_ref(napi_env env, bulk_work *w, napi_value v) {
    if (napi_create_reference(env, v, 1, &w->keep[w->n_keep]) != napi_ok) return 0;
    w->n_keep++;
    return 1;
}

/* renderFrame({width, height, eye:[x,y,z], target?, up?, fovY?, maxSteps?, tolerance?, shading?,
 *              arith?: "fast"|"strict", diskProfile?, devices?: n, virtualRanks?: n,
 *              out?: Float32Array(w*h*4)})
 *   -> {rgba: Float32Array[w*h*4], width, height, rays, acceptedSteps, launches, devices}
 * The frame the reference only produces in its WebGPU compute pass (compute.wgsl.ts:147-258),
 * integrated by the f64 RKF45 kernel; pixel->ray of compute.wgsl.ts:159-187.
 *   devices: n      the image plane is tiled over the first n HIP devices (one gather of finished
 *                   tiles to device 0, RCCL over xGMI; include/gravitas_abi.h grv_engine_create_multi);
 *   virtualRanks: n the same assembly path with n ranks on device 0 (one-GPU hosts, tests);
 *   exchange: "rgba16f"  the gather carries the compute pass's own rgba16float format (renderer.ts:163-176): half
 *                   the bytes, the image is the one-device frame rounded through binary16 (default "rgba32f");
 *   out             render into a caller-owned array -- with allocPinned() memory the frame
 *                   leaves the device in one DMA and no copy is made on the JS side.
 * renderFrameAsync(opts) -> Promise of the same object: the frame runs on the libuv pool with an
 * engine handle of its own, so a worker's tick (physics.worker.ts:111-176) is never held. */
static int wants_image(napi_env env, napi_value opts);
static napi_value render_frame_to_image(napi_env env, engine_box *b, napi_value opts, const GrvCamera *cam,
                                        const GrvRenderParams *p);
static napi_value render_frame_common(napi_env env, napi_callback_info info, int is_async) {
    size_t argc = 1;
    napi_value argv[1], self;
    engine_box *b = NULL;
    if (napi_get_cb_info(env, info, &argc, argv, &self, NULL) != napi_ok) return NULL;
    b = engine_of(env, self);
    if (!b || !b->h) {
        napi_throw_error(env, NULL, "PhysicsEngine: invalid receiver");
        return NULL;
    }
    napi_valuetype t;
    if (argc < 1 || napi_typeof(env, argv[0], &t) != napi_ok || t != napi_object) {
        napi_throw_type_error(env, NULL, "renderFrame expects an options object");
        return NULL;
    }
    const uint32_t w = (uint32_t)obj_f64(env, argv[0], "width", 256);
    const uint32_t h = (uint32_t)obj_f64(env, argv[0], "height", 256);
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 27)) {
        napi_throw_range_error(env, NULL, "renderFrame: width/height out of range");
        return NULL;
    }
    const double devices = obj_f64(env, argv[0], "devices", 1.0), vranks = obj_f64(env, argv[0], "virtualRanks", 0.0);
    if (!(devices >= 1.0 && devices <= 64.0) || !(vranks >= 0.0 && vranks <= 64.0)) {
        napi_throw_range_error(env, NULL, "renderFrame: devices / virtualRanks out of range (1..64)");
        return NULL;
    }
    bulk_work *wk = (bulk_work *)calloc(1, sizeof *wk);
    if (!wk) {
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    wk->box = b;
    wk->kind = 0;
    wk->is_async = is_async;
    wk->mass = b->mass;
    wk->spin = b->spin;
    wk->devices = (int)devices;
    wk->virtual_ranks = (int)vranks;
    {
        char ex[16];
        wk->exchange = (obj_str(env, argv[0], "exchange", ex, sizeof ex) && strcmp(ex, "rgba16f") == 0) ? GRV_EXCHANGE_RGBA16F
                                                                                                      : GRV_EXCHANGE_RGBA32F;
    }
    double eye[3] = {0.0, 0.0, 60.0}, target[3] = {0.0, 0.0, 0.0}, up[3] = {0.0, 1.0, 0.0};
    obj_vec3(env, argv[0], "eye", eye);
    obj_vec3(env, argv[0], "target", target);
    obj_vec3(env, argv[0], "up", up);
    /* fovY in degrees (WebGPUCanvas.tsx:143-151 uses 60); the C ABI takes radians */
    grv_camera_look_at(eye, target, up, obj_f64(env, argv[0], "fovY", 60.0) * (3.14159265358979323846 / 180.0),
                       (double)w / (double)h, &wk->cam);
    GrvRenderParams *p = &wk->p;
    grv_render_params_default(w, h, p);
    p->opt.max_steps = (uint64_t)obj_f64(env, argv[0], "maxSteps", (double)p->opt.max_steps);
    p->opt.tolerance = obj_f64(env, argv[0], "tolerance", p->opt.tolerance);
    p->shading = (int32_t)obj_f64(env, argv[0], "shading", (double)p->shading);
    char buf[16];
    /* "pageThorne": the generate_disk_lut table as the radial temperature profile (disk.rs:175-201) */
    if (obj_str(env, argv[0], "diskProfile", buf, sizeof buf))
        p->disk_profile = strcmp(buf, "pageThorne") == 0 ? GRV_DISK_PROFILE_PAGE_THORNE : GRV_DISK_PROFILE_SHORTCUT;
    if (obj_str(env, argv[0], "arith", buf, sizeof buf))
        p->opt.arith = strcmp(buf, "strict") == 0 ? GRV_ARITH_STRICT : GRV_ARITH_FAST;
    if (wants_image(env, argv[0])) { /* keepOnDevice / image: queued into a device image, nothing crosses PCIe */
        napi_value res = NULL;
        if (is_async || wk->devices > 1 || wk->virtual_ranks > 0)
            napi_throw_type_error(env, NULL, "renderFrame: keepOnDevice / image is the synchronous one-device form "
                                             "(the call only queues; there is nothing to await)");
        else
            res = render_frame_to_image(env, b, argv[0], &wk->cam, &wk->p);
        free(wk);
        return res;
    }
    const size_t n = (size_t)w * h * 4;
    napi_value rgba;
    bool has = false;
    if (napi_has_named_property(env, argv[0], "out", &has) == napi_ok && has) {
        napi_typedarray_type ty;
        size_t len = 0;
        void *data = NULL;
        bool is_ta = false;
        if (napi_get_named_property(env, argv[0], "out", &rgba) != napi_ok ||
            napi_is_typedarray(env, rgba, &is_ta) != napi_ok || !is_ta ||
            napi_get_typedarray_info(env, rgba, &ty, &len, &data, NULL, NULL) != napi_ok ||
            ty != napi_float32_array || len < n) {
            free(wk);
            napi_throw_type_error(env, NULL, "renderFrame: out must be a Float32Array of width*height*4 elements");
            return NULL;
        }
        wk->rgba = (float *)data;
        wk->rgba_elems = n;
        if (is_async) wk->pin = pinned_acquire(data, n * sizeof(float));
        if (is_async && !wk->pin) { /* see bulk_work.own_rgba: the pool thread renders into staging, not into JS-reachable memory */
            wk->own_rgba = stage_acquire(b, n * sizeof(float), &wk->own_stage);
            if (!wk->own_rgba) {
                free(wk);
                napi_throw_error(env, NULL, "renderFrameAsync: cannot allocate the staging image");
                return NULL;
            }
            wk->rgba = wk->own_rgba;
            wk->rgba_elems = n;
        }
    } else {
        napi_value ab;
        void *dst;
        if (napi_create_arraybuffer(env, n * sizeof(float), &dst, &ab) != napi_ok ||
            napi_create_typedarray(env, napi_float32_array, n, ab, 0, &rgba) != napi_ok) {
            free(wk);
            napi_throw_error(env, NULL, "renderFrame: cannot allocate the image");
            return NULL;
        }
        wk->rgba = (float *)dst;
    }
    if (!keep_ref(env, wk, rgba)) {
        stage_release(b, wk->own_rgba, wk->own_stage);
        pinned_release(wk->pin);
        free(wk);
        napi_throw_error(env, NULL, "renderFrame: reference failed");
        return NULL;
    }
    return bulk_dispatch(env, self, wk);
}
static napi_value m_render_frame(napi_env env, napi_callback_info info) { return render_frame_common(env, info, 0); }
static napi_value m_render_frame_async(napi_env env, napi_callback_info info) { return render_frame_common(env, info, 1); }

/* integrate_batch(states: Float64Array(8 n), {method?: "rkf45"|"rk4"|"symplectic", metric?: "ks"|"bl"|
 *                 "schwarzschild", tolerance?, initialStep?, maxSteps?, escapeRadius?, renormalizeInterval?,
 *                 stepSize?, arith?: "strict"|"fast"})
 *   -> {states: Float64Array(8 n), steps: Uint32Array(n), term: Uint8Array(n), drift: Float64Array(n)}
 * n independent integrate() calls (geodesic/mod.rs:180-253) in one launch: what a consumer of
 * integrate_ray_relativistic (lib.rs:422-464, one ray per call) should use for more than a few rays.
 * Defaults are IntegrationOptions::default (integrator.rs:35-47); the Trajectory scalars
 * (mod.rs:150-161) come back per ray.  integrateBatchAsync: the same on the libuv pool. */
static napi_value integrate_batch_common(napi_env env, napi_callback_info info, int is_async) {
    size_t argc = 2;
    napi_value argv[2], self;
    engine_box *b = NULL;
    if (napi_get_cb_info(env, info, &argc, argv, &self, NULL) != napi_ok) return NULL;
    b = engine_of(env, self);
    if (!b || !b->h) {
        napi_throw_error(env, NULL, "PhysicsEngine: invalid receiver");
        return NULL;
    }
    napi_typedarray_type ty;
    size_t len = 0;
    void *data = NULL;
    bool is_ta = false;
    if (argc < 1 || napi_is_typedarray(env, argv[0], &is_ta) != napi_ok || !is_ta ||
        napi_get_typedarray_info(env, argv[0], &ty, &len, &data, NULL, NULL) != napi_ok ||
        ty != napi_float64_array || len % 8 != 0) {
        napi_throw_type_error(env, NULL, "integrate_batch: states must be a Float64Array of 8 n elements");
        return NULL;
    }
    bulk_work *wk = (bulk_work *)calloc(1, sizeof *wk);
    if (!wk) {
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    wk->box = b;
    wk->kind = 1;
    wk->is_async = is_async;
    wk->mass = b->mass;
    wk->spin = b->spin;
    wk->n = len / 8;
    wk->in = (double *)data;
    if (is_async && len) { /* the caller may overwrite or transfer `states` while the work is queued: take a copy */
        wk->own_in = malloc(len * sizeof(double));
        if (!wk->own_in) {
            free(wk);
            napi_throw_error(env, NULL, "out of memory");
            return NULL;
        }
        memcpy(wk->own_in, data, len * sizeof(double));
        wk->in = (double *)wk->own_in;
    }
    GrvOptions *o = &wk->opt;
    grv_options_default(o);
    napi_valuetype t;
    if (argc > 1 && napi_typeof(env, argv[1], &t) == napi_ok && t == napi_object) {
        char buf[24];
        if (obj_str(env, argv[1], "method", buf, sizeof buf))
            o->method = strcmp(buf, "rk4") == 0 ? GRV_METHOD_RK4
                        : strcmp(buf, "symplectic") == 0 ? GRV_METHOD_SYMPLECTIC : GRV_METHOD_RKF45;
        if (obj_str(env, argv[1], "metric", buf, sizeof buf))
            o->metric_kind = strcmp(buf, "bl") == 0 ? GRV_METRIC_KERR_BL
                             : strcmp(buf, "schwarzschild") == 0 ? GRV_METRIC_SCHWARZSCHILD : GRV_METRIC_KERR_KS;
        if (obj_str(env, argv[1], "arith", buf, sizeof buf))
            o->arith = strcmp(buf, "fast") == 0 ? GRV_ARITH_FAST : GRV_ARITH_STRICT;
        o->tolerance = obj_f64(env, argv[1], "tolerance", o->tolerance);
        o->initial_step = obj_f64(env, argv[1], "initialStep", o->initial_step);
        o->max_steps = (uint64_t)obj_f64(env, argv[1], "maxSteps", (double)o->max_steps);
        o->escape_radius = obj_f64(env, argv[1], "escapeRadius", o->escape_radius);
        o->renormalize_interval = (uint64_t)obj_f64(env, argv[1], "renormalizeInterval", (double)o->renormalize_interval);
        o->step_size = obj_f64(env, argv[1], "stepSize", o->step_size);
        /* IntegrationOptions.record_path (integrator.rs:32): the result gains `paths` and `counts` */
        o->record_path = obj_flag(env, argv[1], "recordPath");
        const double mp = obj_f64(env, argv[1], "maxPoints", (double)o->max_steps + 1.0);
        wk->max_points = o->record_path ? (size_t)(mp < 0.0 ? 0.0 : mp > 4294967295.0 ? 4294967295.0 : mp) : 0;
    }
    const size_t n = wk->n;
    if (o->record_path && wk->max_points && n > ((size_t)1 << 33) / 64 / wk->max_points) { /* 8 GiB of rows */
        free(wk->own_in);
        free(wk);
        napi_throw_range_error(env, NULL, "integrate_batch: n x maxPoints is too large; pass a smaller maxPoints");
        return NULL;
    }
    napi_value ab[6], ta[6];
    void *mem[6] = {NULL, NULL, NULL, NULL, NULL, NULL};
    const size_t bytes[6] = {n * 64, n * 4, n, n * 8, n * wk->max_points * 64, n * 4};
    const napi_typedarray_type tys[6] = {napi_float64_array, napi_uint32_array, napi_uint8_array, napi_float64_array,
                                         napi_float64_array, napi_uint32_array};
    const size_t counts[6] = {n * 8, n, n, n, n * wk->max_points * 8, n};
    const int n_out = o->record_path ? 6 : 4;
    for (int k = 0; k < n_out; k++) {
        if (napi_create_arraybuffer(env, bytes[k], &mem[k], &ab[k]) != napi_ok ||
            napi_create_typedarray(env, tys[k], counts[k], ab[k], 0, &ta[k]) != napi_ok || !keep_ref(env, wk, ta[k])) {
            bulk_release(env, wk);
            napi_throw_error(env, NULL, "integrate_batch: cannot allocate the outputs");
            return NULL;
        }
    }
    if (!keep_ref(env, wk, argv[0])) { /* the input stays alive while queued (the async form works on its own copy) */
        bulk_release(env, wk);
        napi_throw_error(env, NULL, "integrate_batch: reference failed");
        return NULL;
    }
    wk->out_states = (double *)mem[0];
    wk->steps = (uint32_t *)mem[1];
    wk->term = (uint8_t *)mem[2];
    wk->drift = (double *)mem[3];
    wk->paths = (double *)mem[4];
    wk->counts = (uint32_t *)mem[5];
    if (n == 0) wk->is_async = is_async; /* an empty batch still resolves (with empty arrays) */
    return bulk_dispatch(env, self, wk);
}
static napi_value m_integrate_batch(napi_env env, napi_callback_info info) { return integrate_batch_common(env, info, 0); }
static napi_value m_integrate_batch_async(napi_env env, napi_callback_info info) { return integrate_batch_common(env, info, 1); }

/* ---------------------------------------------------------------------------------------------
 * Device-resident frames (include/gravitas_abi.h "device images").
 *
 *   const img = engine.createImage(w, h)                    -> DeviceImage
 *   engine.renderFrame({..., keepOnDevice: true})           -> {image: DeviceImage (new), width, height, queued: true}
 *   engine.renderFrame({..., image: img})                   -> the same into an image the caller keeps
 *   engine.renderShaderFrame({kernel: "glsl"|"wgsl", width, height, maxSteps, arith, image})
 *   engine.renderWebGLFrame({..., image}) / renderWebGPUFrame(cu, pp, {image})
 *   engine.postBloom(scene, out, {intensity?, threshold?, blurPasses?, fast?}) / engine.postTaa(cur, hist, out, {...})
 *   img.read(out?) / engine.readImage(img, out?)            -> Float32Array (the one D2H; allocPinned memory: one DMA)
 *   img.readAsync(out)                                      -> Promise (out must live in allocPinned() memory)
 *   img.stats()                                             -> {rays, acceptedSteps, ...} of the frame that wrote it
 *   img.ready() / img.wait() / img.free()
 *   engine.statsAccumulate(on) / frameStats() / frameStatsReset() / synchronize()
 *
 * None of the render / post calls waits for the GPU: they queue on the image's stream and return, so
 * a JS frame loop that alternates two images keeps two frames in flight exactly as the reference's
 * renderer keeps submitting command buffers (src/rendering/webgpu/renderer.ts:280-411).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    grv_image *img;
    uint32_t w, h;
    int pending; /* readAsync works queued (main thread only) */
    int freed;   /* free() arrived while reads were pending: the last one destroys the image */
    int64_t accounted; /* bytes reported to the JS heap's external-memory counter (the collector cannot see HBM:
                          an unreachable 133 MB image should weigh on it like one) */
} image_box;

static void image_release(napi_env env, image_box *ib) {
    if (ib->img) grv_image_destroy(ib->img);
    ib->img = NULL;
    if (ib->accounted) {
        int64_t adj;
        napi_adjust_external_memory(env, -ib->accounted, &adj);
        ib->accounted = 0;
    }
}

static napi_ref g_image_ctor = NULL;

static void image_finalize(napi_env env, void *data, void *hint) {
    (void)hint;
    image_box *ib = (image_box *)data;
    if (ib) {
        image_release(env, ib); /* (a pending read holds a reference to the object) */
        free(ib);
    }
}
static napi_value image_ctor(napi_env env, napi_callback_info info) {
    napi_value self;
    NAPI_OK(napi_get_cb_info(env, info, NULL, NULL, &self, NULL));
    return self; /* wrapped by wrap_image(); a bare `new DeviceImage()` has no device memory and every method says so */
}
static image_box *image_of(napi_env env, napi_value v) {
    image_box *ib = NULL;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_object) return NULL;
    /* (an engine object unwraps too: tell them apart by the constructor BEFORE looking at the wrapped pointer) */
    napi_value ctor;
    bool is = false;
    if (!g_image_ctor || napi_get_reference_value(env, g_image_ctor, &ctor) != napi_ok ||
        napi_instanceof(env, v, ctor, &is) != napi_ok || !is) return NULL;
    if (napi_unwrap(env, v, (void **)&ib) != napi_ok || !ib || !ib->img) return NULL;
    return ib;
}
static napi_value wrap_image(napi_env env, grv_image *img) {
    napi_value ctor, obj;
    image_box *ib = (image_box *)calloc(1, sizeof *ib);
    if (!ib || !g_image_ctor || napi_get_reference_value(env, g_image_ctor, &ctor) != napi_ok ||
        napi_new_instance(env, ctor, 0, NULL, &obj) != napi_ok) {
        free(ib);
        grv_image_destroy(img);
        napi_throw_error(env, NULL, "DeviceImage: cannot create the object");
        return NULL;
    }
    ib->img = img;
    ib->w = grv_image_width(img);
    ib->h = grv_image_height(img);
    if (napi_wrap(env, obj, ib, image_finalize, NULL, NULL) != napi_ok) {
        free(ib);
        grv_image_destroy(img);
        napi_throw_error(env, NULL, "DeviceImage: cannot wrap the object");
        return NULL;
    }
    {
        int64_t adj;
        ib->accounted = (int64_t)grv_image_bytes(img);
        napi_adjust_external_memory(env, ib->accounted, &adj);
    }
    napi_set_named_property(env, obj, "width", mk_f64(env, ib->w));
    napi_set_named_property(env, obj, "height", mk_f64(env, ib->h));
    napi_set_named_property(env, obj, "bytes", mk_f64(env, (double)grv_image_bytes(img)));
    return obj;
}

/* the image a render call writes: opts.image, or a new one under opts.keepOnDevice; *created tells which */
static image_box *target_image(napi_env env, engine_box *b, napi_value opts, uint32_t w, uint32_t h, napi_value *obj,
                               const char *who) {
    bool has = false;
    char msg[160];
    if (napi_has_named_property(env, opts, "image", &has) == napi_ok && has) {
        if (napi_get_named_property(env, opts, "image", obj) != napi_ok) return NULL;
        image_box *ib = image_of(env, *obj);
        if (!ib) {
            snprintf(msg, sizeof msg, "%s: image must be a live DeviceImage (engine.createImage)", who);
            napi_throw_type_error(env, NULL, msg);
            return NULL;
        }
        if (ib->pending > 0) {
            snprintf(msg, sizeof msg, "%s: a readAsync of this image is still pending (await it first)", who);
            napi_throw_error(env, NULL, msg);
            return NULL;
        }
        if (ib->w != w || ib->h != h) {
            snprintf(msg, sizeof msg, "%s: a %u x %u frame into an image of %u x %u", who, w, h, ib->w, ib->h);
            napi_throw_range_error(env, NULL, msg);
            return NULL;
        }
        return ib;
    }
    grv_image *img = NULL;
    if (grv_image_create(b->h, w, h, &img) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    *obj = wrap_image(env, img);
    if (!*obj) return NULL;
    image_box *ib = NULL;
    napi_unwrap(env, *obj, (void **)&ib);
    return ib;
}
static int wants_image(napi_env env, napi_value opts) {
    bool has = false;
    if (napi_has_named_property(env, opts, "image", &has) == napi_ok && has) return 1;
    return obj_flag(env, opts, "keepOnDevice");
}
static napi_value queued_result(napi_env env, napi_value image_obj, uint32_t w, uint32_t h) {
    napi_value out, t;
    if (napi_create_object(env, &out) != napi_ok) return NULL;
    napi_set_named_property(env, out, "image", image_obj);
    napi_set_named_property(env, out, "width", mk_f64(env, w));
    napi_set_named_property(env, out, "height", mk_f64(env, h));
    napi_set_named_property(env, out, "rays", mk_f64(env, (double)w * h));
    napi_get_boolean(env, true, &t);
    napi_set_named_property(env, out, "queued", t);
    return out;
}

/* createImage(width, height, {streamOf?: DeviceImage}): with streamOf the new image is written on that image's
 * compute stream (frames in queue order; each image still reads back on a copy stream of its own) */
static napi_value m_create_image(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    const double w = argc > 0 ? arg_f64(env, argv[0]) : 0.0, h = argc > 1 ? arg_f64(env, argv[1]) : 0.0;
    if (!(w >= 1.0 && h >= 1.0 && w * h <= 134217728.0)) {
        napi_throw_range_error(env, NULL, "createImage: width/height out of range");
        return NULL;
    }
    grv_image *img = NULL, *share = NULL;
    napi_valuetype t;
    bool has = false;
    if (argc > 2 && napi_typeof(env, argv[2], &t) == napi_ok && t == napi_object &&
        napi_has_named_property(env, argv[2], "streamOf", &has) == napi_ok && has) {
        napi_value so;
        image_box *sb = napi_get_named_property(env, argv[2], "streamOf", &so) == napi_ok ? image_of(env, so) : NULL;
        if (!sb) {
            napi_throw_type_error(env, NULL, "createImage: streamOf must be a live DeviceImage");
            return NULL;
        }
        share = sb->img;
    }
    if ((share ? grv_image_create_shared(b->h, (uint32_t)w, (uint32_t)h, share, &img)
               : grv_image_create(b->h, (uint32_t)w, (uint32_t)h, &img)) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    return wrap_image(env, img);
}

static image_box *this_image(napi_env env, napi_callback_info info, size_t *argc, napi_value *argv, napi_value *self) {
    image_box *ib = NULL;
    if (napi_get_cb_info(env, info, argc, argv, self, NULL) != napi_ok) return NULL;
    ib = image_of(env, *self); /* (checks the constructor: a method borrowed onto another object is refused) */
    if (!ib) {
        napi_throw_error(env, NULL, "DeviceImage: no device memory behind this object (freed, or not made by createImage)");
        return NULL;
    }
    return ib;
}

/* the Float32Array a read lands in: `v` if given (checked), else a new one; *data its memory */
static napi_value read_target(napi_env env, napi_value v, int given, size_t n, float **data, const char *who) {
    napi_value ta;
    if (given) {
        napi_typedarray_type ty;
        size_t len = 0;
        void *d = NULL;
        bool is_ta = false;
        if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta ||
            napi_get_typedarray_info(env, v, &ty, &len, &d, NULL, NULL) != napi_ok || ty != napi_float32_array || len < n) {
            char msg[128];
            snprintf(msg, sizeof msg, "%s: out must be a Float32Array of width*height*4 elements", who);
            napi_throw_type_error(env, NULL, msg);
            return NULL;
        }
        *data = (float *)d;
        return v;
    }
    napi_value ab;
    void *d;
    if (napi_create_arraybuffer(env, n * sizeof(float), &d, &ab) != napi_ok ||
        napi_create_typedarray(env, napi_float32_array, n, ab, 0, &ta) != napi_ok) {
        napi_throw_error(env, NULL, "read: cannot allocate the image");
        return NULL;
    }
    *data = (float *)d;
    return ta;
}
static napi_value image_read_into(napi_env env, image_box *ib, napi_value out, int given) {
    const size_t n = (size_t)ib->w * ib->h * 4;
    if (ib->pending > 0) {
        napi_throw_error(env, NULL, "DeviceImage.read: a readAsync of this image is still pending (await it first)");
        return NULL;
    }
    float *dst = NULL;
    napi_value ta = read_target(env, out, given, n, &dst, "DeviceImage.read");
    if (!ta) return NULL;
    if (grv_image_read(ib->img, dst, n) != GRV_OK) {
        napi_throw_error(env, NULL, grv_image_last_error(ib->img));
        return NULL;
    }
    return ta;
}
static napi_value im_read(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1], self;
    image_box *ib = this_image(env, info, &argc, argv, &self);
    if (!ib) return NULL;
    napi_valuetype t = napi_undefined;
    if (argc > 0) napi_typeof(env, argv[0], &t);
    return image_read_into(env, ib, argc > 0 ? argv[0] : NULL, argc > 0 && t != napi_undefined && t != napi_null);
}
/* engine.readImage(img, out?) */
static napi_value m_read_image(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    image_box *ib = argc > 0 ? image_of(env, argv[0]) : NULL;
    if (!ib) {
        napi_throw_type_error(env, NULL, "readImage(image, out?): image must be a live DeviceImage");
        return NULL;
    }
    napi_valuetype t = napi_undefined;
    if (argc > 1) napi_typeof(env, argv[1], &t);
    return image_read_into(env, ib, argc > 1 ? argv[1] : NULL, argc > 1 && t != napi_undefined && t != napi_null);
}

typedef struct {
    image_box *ib;
    pinned_block *pin;
    float *dst;
    size_t n;
    int rc;
    char err[256];
    napi_ref self_ref, out_ref;
    napi_deferred deferred;
    napi_async_work work;
} read_work;
static void read_async_execute(napi_env env, void *data) {
    (void)env;
    read_work *w = (read_work *)data;
    /* image-only call: waits for the image's producers on this thread, then copies */
    w->rc = grv_image_read(w->ib->img, w->dst, w->n);
    if (w->rc != GRV_OK) snprintf(w->err, sizeof w->err, "%s", grv_image_last_error(w->ib->img));
}
static void read_async_complete(napi_env env, napi_status status, void *data) {
    read_work *w = (read_work *)data;
    napi_value out = NULL;
    if (status == napi_ok && w->rc == GRV_OK) napi_get_reference_value(env, w->out_ref, &out);
    if (out) {
        napi_resolve_deferred(env, w->deferred, out);
    } else {
        napi_value msg, err;
        napi_create_string_utf8(env, w->err[0] ? w->err : "readAsync failed", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, w->deferred, err);
    }
    napi_delete_async_work(env, w->work);
    pinned_release(w->pin);
    if (--w->ib->pending == 0 && w->ib->freed && w->ib->img) image_release(env, w->ib);
    napi_delete_reference(env, w->self_ref);
    napi_delete_reference(env, w->out_ref);
    free(w);
}
/* img.readAsync(out) -> Promise<out>: a pool thread waits for the image's producers, copies, and the promise
 * settles when the pixels have landed.  `out` must live in allocPinned() memory: the DMA writes it while JS runs.
 * Until then the image is busy: rendering into it (or reading it again) throws. */
static napi_value im_read_async(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1], self, promise, name;
    image_box *ib = this_image(env, info, &argc, argv, &self);
    if (!ib) return NULL;
    const size_t n = (size_t)ib->w * ib->h * 4;
    float *dst = NULL;
    if (ib->pending > 0) {
        napi_throw_error(env, NULL, "DeviceImage.readAsync: a readAsync of this image is still pending (await it first)");
        return NULL;
    }
    if (argc < 1 || !read_target(env, argv[0], 1, n, &dst, "DeviceImage.readAsync")) {
        if (argc < 1) napi_throw_type_error(env, NULL, "DeviceImage.readAsync(out): out is required");
        return NULL;
    }
    pinned_block *pin = pinned_acquire(dst, n * sizeof(float));
    if (!pin) {
        napi_throw_type_error(env, NULL, "DeviceImage.readAsync: out must live in allocPinned() memory (use read(out) otherwise)");
        return NULL;
    }
    read_work *w = (read_work *)calloc(1, sizeof *w);
    if (!w) {
        pinned_release(pin);
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    w->ib = ib;
    w->pin = pin;
    w->dst = dst;
    w->n = n;
    if (napi_create_promise(env, &w->deferred, &promise) != napi_ok ||
        napi_create_reference(env, self, 1, &w->self_ref) != napi_ok ||
        napi_create_reference(env, argv[0], 1, &w->out_ref) != napi_ok ||
        napi_create_string_utf8(env, "gravitas.read", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, NULL, name, read_async_execute, read_async_complete, w, &w->work) != napi_ok ||
        napi_queue_async_work(env, w->work) != napi_ok) {
        pinned_release(pin);
        if (w->self_ref) napi_delete_reference(env, w->self_ref);
        if (w->out_ref) napi_delete_reference(env, w->out_ref);
        free(w);
        napi_throw_error(env, NULL, "cannot queue async work");
        return NULL;
    }
    ib->pending++;
    return promise;
}
static napi_value stats_object(napi_env env, const GrvFrameStats *st) {
    napi_value out, tc;
    if (napi_create_object(env, &out) != napi_ok || napi_create_array_with_length(env, 5, &tc) != napi_ok) return NULL;
    napi_set_named_property(env, out, "rays", mk_f64(env, (double)st->rays));
    napi_set_named_property(env, out, "acceptedSteps", mk_f64(env, (double)st->accepted_steps));
    napi_set_named_property(env, out, "rkfTries", mk_f64(env, (double)st->rkf_tries));
    napi_set_named_property(env, out, "crossings", mk_f64(env, (double)st->crossings));
    napi_set_named_property(env, out, "maxDrift", mk_f64(env, st->max_drift));
    napi_set_named_property(env, out, "launches", mk_f64(env, (double)st->launches));
    napi_set_named_property(env, out, "integrateMs", mk_f64(env, (double)st->integrate_ms));
    for (uint32_t k = 0; k < 5; k++) napi_set_element(env, tc, k, mk_f64(env, (double)st->term_count[k]));
    napi_set_named_property(env, out, "termCount", tc);
    return out;
}
static napi_value im_stats(napi_env env, napi_callback_info info) {
    size_t argc = 0;
    napi_value self;
    image_box *ib = this_image(env, info, &argc, NULL, &self);
    if (!ib) return NULL;
    GrvFrameStats st;
    if (grv_image_frame_stats(ib->img, &st) != GRV_OK) {
        napi_throw_error(env, NULL, grv_image_last_error(ib->img));
        return NULL;
    }
    return stats_object(env, &st);
}
static napi_value im_wait(napi_env env, napi_callback_info info) {
    size_t argc = 0;
    napi_value self;
    image_box *ib = this_image(env, info, &argc, NULL, &self);
    if (!ib) return NULL;
    if (ib->pending > 0) {
        napi_throw_error(env, NULL, "DeviceImage.wait: a readAsync of this image is pending -- await its promise instead");
        return NULL;
    }
    if (grv_image_wait(ib->img) != GRV_OK) {
        napi_throw_error(env, NULL, grv_image_last_error(ib->img));
        return NULL;
    }
    return self;
}
static napi_value im_ready(napi_env env, napi_callback_info info) {
    size_t argc = 0;
    napi_value self, v;
    image_box *ib = this_image(env, info, &argc, NULL, &self);
    if (!ib) return NULL;
    const int q = ib->pending > 0 ? 0 : grv_image_query(ib->img);
    if (q < 0) {
        napi_throw_error(env, NULL, grv_image_last_error(ib->img));
        return NULL;
    }
    napi_get_boolean(env, q == 1, &v);
    return v;
}
static napi_value im_free(napi_env env, napi_callback_info info) {
    size_t argc = 0;
    napi_value self;
    image_box *ib = NULL;
    NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));
    if (napi_unwrap(env, self, (void **)&ib) == napi_ok && ib && ib->img) {
        if (ib->pending > 0) ib->freed = 1; /* the last pending read destroys it */
        else image_release(env, ib);
    }
    return NULL;
}

/* renderFrame({..., keepOnDevice | image}): the f64 frame into a device image, queued, nothing waited for */
static napi_value render_frame_to_image(napi_env env, engine_box *b, napi_value opts, const GrvCamera *cam,
                                        const GrvRenderParams *p) {
    napi_value obj;
    image_box *ib = target_image(env, b, opts, p->width, p->height, &obj, "renderFrame");
    if (!ib) return NULL;
    if (grv_render_frame_image(b->h, cam, p, ib->img) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    return queued_result(env, obj, p->width, p->height);
}

/* renderShaderFrame({kernel: "glsl" | "wgsl", width, height, maxSteps?, arith?: "fast"|"strict"|"packed",
 *                    eye?/target?/up?/fovY? (wgsl camera), spin?, mass?, zoom?, time?, features?, image | keepOnDevice,
 *                    out?})
 * ONE f32 march launch -- the GLSL fragment march (fragment.glsl.ts:40-334, BASELINE configs[1]) or the WGSL
 * compute march (compute.wgsl.ts:147-258, configs[3]) -- without the renderers' post chain.  With an image
 * the call queues and returns {image, queued: true}; otherwise -> {rgba: Float32Array, acceptedSteps}. */
static napi_value m_render_shader_frame(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    napi_valuetype t;
    if (argc < 1 || napi_typeof(env, argv[0], &t) != napi_ok || t != napi_object) {
        napi_throw_type_error(env, NULL, "renderShaderFrame expects an options object");
        return NULL;
    }
    const uint32_t w = (uint32_t)obj_f64(env, argv[0], "width", 256), h = (uint32_t)obj_f64(env, argv[0], "height", 256);
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 27)) {
        napi_throw_range_error(env, NULL, "renderShaderFrame: width/height out of range");
        return NULL;
    }
    char kernel[16] = "glsl", arith[16] = "fast";
    obj_str(env, argv[0], "kernel", kernel, sizeof kernel);
    if (!kernel[0]) strcpy(kernel, "glsl");
    obj_str(env, argv[0], "arith", arith, sizeof arith);
    const int is_glsl = strcmp(kernel, "glsl") == 0;
    if (!is_glsl && strcmp(kernel, "wgsl") != 0) {
        napi_throw_type_error(env, NULL, "renderShaderFrame: kernel must be \"glsl\" or \"wgsl\"");
        return NULL;
    }
    const int ar = strcmp(arith, "strict") == 0 ? GRV_ARITH_STRICT : strcmp(arith, "packed") == 0 ? GRV_ARITH_FAST_PACKED : GRV_ARITH_FAST;
    const double mass = obj_f64(env, argv[0], "mass", b->mass), spin = obj_f64(env, argv[0], "spin", b->spin);
    GrvGlslParams gp;
    GrvWgslParams wp;
    if (is_glsl) {
        if (ar == GRV_ARITH_FAST_PACKED) {
            napi_throw_type_error(env, NULL, "renderShaderFrame: arith \"packed\" is the two-rays-per-lane form of the WGSL march");
            return NULL;
        }
        grv_glsl_params_default(w, h, mass, spin, &gp);
        gp.zoom = (float)obj_f64(env, argv[0], "zoom", gp.zoom);
        gp.time = (float)obj_f64(env, argv[0], "time", 0.0);
        gp.max_ray_steps = (int32_t)obj_f64(env, argv[0], "maxSteps", gp.max_ray_steps);
        gp.features = (uint32_t)obj_f64(env, argv[0], "features", (double)gp.features);
        double mouse[3] = {gp.mouse[0], gp.mouse[1], 0.0};
        obj_vec3(env, argv[0], "mouse", mouse);
        gp.mouse[0] = (float)mouse[0];
        gp.mouse[1] = (float)mouse[1];
        gp.arith = ar;
    } else {
        GrvCamera cam;
        double eye[3] = {0.0, 0.0, 60.0}, target[3] = {0.0, 0.0, 0.0}, up[3] = {0.0, 1.0, 0.0};
        obj_vec3(env, argv[0], "eye", eye);
        obj_vec3(env, argv[0], "target", target);
        obj_vec3(env, argv[0], "up", up);
        grv_camera_look_at(eye, target, up, obj_f64(env, argv[0], "fovY", 60.0) * (3.14159265358979323846 / 180.0),
                           (double)w / (double)h, &cam);
        grv_wgsl_params_default(w, h, &cam, mass, spin, &wp);
        wp.max_steps = (int32_t)obj_f64(env, argv[0], "maxSteps", (double)wp.max_steps);
        wp.arith = ar;
    }
    if (wants_image(env, argv[0])) {
        napi_value obj;
        image_box *ib = target_image(env, b, argv[0], w, h, &obj, "renderShaderFrame");
        if (!ib) return NULL;
        const int rc = is_glsl ? grv_render_frame_glsl_image(b->h, &gp, ib->img) : grv_render_frame_wgsl_image(b->h, &wp, ib->img);
        if (rc != GRV_OK) {
            napi_throw_error(env, NULL, grv_last_error(b->h));
            return NULL;
        }
        return queued_result(env, obj, w, h);
    }
    /* host form: through a temporary image (one D2H) */
    grv_image *img = NULL;
    if (grv_image_create(b->h, w, h, &img) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    int rc = is_glsl ? grv_render_frame_glsl_image(b->h, &gp, img) : grv_render_frame_wgsl_image(b->h, &wp, img);
    if (rc != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        grv_image_destroy(img);
        return NULL;
    }
    const size_t n = (size_t)w * h * 4;
    float *dst = NULL;
    bool has = false;
    napi_value outv = NULL, ta, res;
    if (napi_has_named_property(env, argv[0], "out", &has) == napi_ok && has) napi_get_named_property(env, argv[0], "out", &outv);
    ta = read_target(env, outv, has, n, &dst, "renderShaderFrame");
    GrvFrameStats st;
    memset(&st, 0, sizeof st);
    if (ta) {
        rc = grv_image_read(img, dst, n);
        if (rc == GRV_OK) rc = grv_image_frame_stats(img, &st);
        if (rc != GRV_OK) napi_throw_error(env, NULL, grv_image_last_error(img));
    }
    grv_image_destroy(img);
    if (!ta || rc != GRV_OK) return NULL;
    NAPI_OK(napi_create_object(env, &res));
    napi_set_named_property(env, res, "rgba", ta);
    napi_set_named_property(env, res, "width", mk_f64(env, w));
    napi_set_named_property(env, res, "height", mk_f64(env, h));
    napi_set_named_property(env, res, "acceptedSteps", mk_f64(env, (double)st.accepted_steps));
    return res;
}

/* postBloom(scene: DeviceImage, out: DeviceImage, {intensity?, threshold?, blurPasses?, halfStorage?, fast?}) -> out
 * BloomManager.applyBloomToTexture (src/rendering/bloom.ts:443-583) between two device images */
static napi_value m_post_bloom(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    image_box *src = argc > 0 ? image_of(env, argv[0]) : NULL, *dst = argc > 1 ? image_of(env, argv[1]) : NULL;
    if (!src || !dst) {
        napi_throw_type_error(env, NULL, "postBloom(scene, out, opts?): scene and out must be live DeviceImages");
        return NULL;
    }
    if (dst->pending > 0) {
        napi_throw_error(env, NULL, "postBloom: a readAsync of `out` is still pending (await it first)");
        return NULL;
    }
    GrvBloomParams p;
    grv_bloom_params_default(src->w, src->h, &p);
    napi_valuetype t;
    if (argc > 2 && napi_typeof(env, argv[2], &t) == napi_ok && t == napi_object) {
        p.intensity = (float)obj_f64(env, argv[2], "intensity", p.intensity);
        p.threshold = (float)obj_f64(env, argv[2], "threshold", p.threshold);
        p.blur_passes = (int32_t)obj_f64(env, argv[2], "blurPasses", p.blur_passes);
        p.half_storage = (int32_t)obj_f64(env, argv[2], "halfStorage", p.half_storage);
        if (obj_flag(env, argv[2], "fast")) p.arith = GRV_ARITH_FAST;
    }
    if (grv_post_bloom_image(b->h, &p, src->img, dst->img) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    return argv[1];
}
/* postTaa(current, history, out, {blendFactor?, cameraMoving?, halfStorage?, fast?}) -> out
 * ReprojectionManager.resolve (src/rendering/reprojection.ts:196-262) */
static napi_value m_post_taa(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    image_box *cur = argc > 0 ? image_of(env, argv[0]) : NULL, *hist = argc > 1 ? image_of(env, argv[1]) : NULL,
              *dst = argc > 2 ? image_of(env, argv[2]) : NULL;
    if (!cur || !hist || !dst) {
        napi_throw_type_error(env, NULL, "postTaa(current, history, out, opts?): three live DeviceImages");
        return NULL;
    }
    if (dst->pending > 0) {
        napi_throw_error(env, NULL, "postTaa: a readAsync of `out` is still pending (await it first)");
        return NULL;
    }
    GrvTaaParams p;
    memset(&p, 0, sizeof p);
    p.width = cur->w;
    p.height = cur->h;
    p.blend_factor = 0.75f;
    p.half_storage = 1;
    p.arith = GRV_ARITH_STRICT;
    napi_valuetype t;
    if (argc > 3 && napi_typeof(env, argv[3], &t) == napi_ok && t == napi_object) {
        p.blend_factor = (float)obj_f64(env, argv[3], "blendFactor", p.blend_factor);
        p.camera_moving = obj_flag(env, argv[3], "cameraMoving");
        p.half_storage = (int32_t)obj_f64(env, argv[3], "halfStorage", p.half_storage);
        if (obj_flag(env, argv[3], "fast")) p.arith = GRV_ARITH_FAST;
    }
    if (grv_post_taa_resolve_image(b->h, &p, cur->img, hist->img, dst->img) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    return argv[2];
}

/* statsAccumulate(on), frameStats() -> {rays, acceptedSteps, ...}, frameStatsReset(), synchronize():
 * grv_stats_accumulate / grv_frame_stats / grv_frame_stats_reset of the synchronous handle.  With
 * accumulation on, a loop of queued frames needs no read-back: frameStats() after the loop waits for the
 * device and returns the sums (what bench.py does around its timed region). */
static napi_value m_stats_accumulate(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    bool on = true;
    if (argc > 0) napi_coerce_to_bool(env, argv[0], &argv[0]), napi_get_value_bool(env, argv[0], &on);
    grv_stats_accumulate(b->h, on ? 1 : 0);
    return NULL;
}
static napi_value m_frame_stats(napi_env env, napi_callback_info info) {
    size_t argc = 0;
    engine_box *b = unwrap(env, info, &argc, NULL);
    if (!b) return NULL;
    GrvFrameStats st;
    if (grv_frame_stats(b->h, NULL, &st) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    return stats_object(env, &st);
}
static napi_value m_frame_stats_reset(napi_env env, napi_callback_info info) {
    size_t argc = 0;
    engine_box *b = unwrap(env, info, &argc, NULL);
    if (!b) return NULL;
    if (grv_frame_stats_reset(b->h, NULL) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    return NULL;
}
static napi_value m_synchronize(napi_env env, napi_callback_info info) {
    size_t argc = 0;
    engine_box *b = unwrap(env, info, &argc, NULL);
    if (!b) return NULL;
    if (grv_engine_synchronize(b->h) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    return NULL;
}

/* allocPinned(bytes) -> ArrayBuffer over page-locked host memory (grv_host_alloc): a
 * `new Float32Array(buf)` passed as renderFrame({out}) receives the frame in one DMA, with no copy
 * on the JS side.  Freed when the ArrayBuffer is collected. */
static void pinned_finalize(napi_env env, void *data, void *hint) {
    (void)env;
    (void)data;
    pinned_block *blk = (pinned_block *)hint;
    pthread_mutex_lock(&g_pinned_mu);
    blk->dead = 1;
    if (blk->refs == 0) pinned_drop_locked(blk); /* else: the last queued work that writes into it frees it */
    pthread_mutex_unlock(&g_pinned_mu);
}
static napi_value f_alloc_pinned(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1], buf;
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    const double bytes = argc > 0 ? arg_f64(env, argv[0]) : 0.0;
    if (!(bytes >= 1.0 && bytes <= 68719476736.0)) {
        napi_throw_range_error(env, NULL, "allocPinned: size out of range");
        return NULL;
    }
    void *p = grv_host_alloc((size_t)bytes);
    pinned_block *blk = p ? (pinned_block *)calloc(1, sizeof *blk) : NULL;
    if (!p || !blk) {
        grv_host_free(p);
        napi_throw_error(env, NULL, "allocPinned: hipHostMalloc failed (no HIP device?)");
        return NULL;
    }
    blk->p = (uint8_t *)p;
    blk->bytes = (size_t)bytes;
    pthread_mutex_lock(&g_pinned_mu);
    blk->next = g_pinned;
    g_pinned = blk;
    pthread_mutex_unlock(&g_pinned_mu);
    if (napi_create_external_arraybuffer(env, p, (size_t)bytes, pinned_finalize, blk, &buf) != napi_ok) {
        pthread_mutex_lock(&g_pinned_mu);
        pinned_drop_locked(blk);
        pthread_mutex_unlock(&g_pinned_mu);
        napi_throw_error(env, NULL, "allocPinned: cannot wrap the allocation");
        return NULL;
    }
    return buf;
}

/* renderWebGPUFrame(cameraUniforms: Float32Array(88), physicsParams: Float32Array(8),
 *                   {maxSteps?, arith?}) -> Float32Array[w*h*4]
 * WebGPURenderer.render (src/rendering/webgpu/renderer.ts:280-411) with the uniform blocks exactly
 * as writeCameraUniforms / writePhysicsParams fill them; history and frame counter live in the engine. */
static napi_value m_render_webgpu_frame(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    float *blocks[2] = {NULL, NULL};
    const size_t want[2] = {88, 8};
    for (int k = 0; k < 2; k++) {
        napi_typedarray_type ty;
        size_t len = 0;
        void *data = NULL;
        bool is_ta = false;
        if (argc <= (size_t)k || napi_is_typedarray(env, argv[k], &is_ta) != napi_ok || !is_ta ||
            napi_get_typedarray_info(env, argv[k], &ty, &len, &data, NULL, NULL) != napi_ok ||
            ty != napi_float32_array || len < want[k]) {
            napi_throw_type_error(env, NULL, "renderWebGPUFrame(Float32Array(88), Float32Array(8), opts?)");
            return NULL;
        }
        blocks[k] = (float *)data;
    }
    int32_t max_steps = 150, arith = GRV_ARITH_STRICT;
    napi_valuetype t;
    if (argc > 2 && napi_typeof(env, argv[2], &t) == napi_ok && t == napi_object) {
        max_steps = (int32_t)obj_f64(env, argv[2], "maxSteps", 150.0);
        bool has = false;
        char buf[16] = {0};
        size_t len = 0;
        napi_value v;
        if (napi_has_named_property(env, argv[2], "arith", &has) == napi_ok && has &&
            napi_get_named_property(env, argv[2], "arith", &v) == napi_ok &&
            napi_get_value_string_utf8(env, v, buf, sizeof buf, &len) == napi_ok)
            arith = strcmp(buf, "fast") == 0 ? GRV_ARITH_FAST : GRV_ARITH_STRICT;
    }
    if (argc > 2 && napi_typeof(env, argv[2], &t) == napi_ok && t == napi_object && wants_image(env, argv[2])) {
        napi_value obj; /* present into a device image: queued, nothing waited for */
        const uint32_t w = (uint32_t)blocks[1][2], h = (uint32_t)blocks[1][3];
        if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 27)) {
            napi_throw_range_error(env, NULL, "renderWebGPUFrame: resolution out of range");
            return NULL;
        }
        image_box *ib = target_image(env, b, argv[2], w, h, &obj, "renderWebGPUFrame");
        if (!ib) return NULL;
        if (grv_webgpu_render_image(b->h, blocks[0], blocks[1], max_steps, arith, ib->img) != GRV_OK) {
            napi_throw_error(env, NULL, grv_last_error(b->h));
            return NULL;
        }
        return queued_result(env, obj, w, h);
    }
    const size_t n = (size_t)(uint32_t)blocks[1][2] * (uint32_t)blocks[1][3] * 4;
    napi_value ab, ta;
    void *dst;
    NAPI_OK(napi_create_arraybuffer(env, n * sizeof(float), &dst, &ab));
    if (grv_webgpu_render_host(b->h, blocks[0], blocks[1], max_steps, arith, (float *)dst) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, n, ab, 0, &ta));
    return ta;
}

/* renderWebGLFrame({width, height, spin?, zoom?, mouse?, time?, maxRaySteps?, bloom?, cameraMoving?,
 *                   arith?}) -> Float32Array[w*h*4]: WebGLRenderer.render's scene + TAA + bloom chain
 * (src/rendering/webgl/renderer.ts:173-422) with the reference's default uniforms. */
static napi_value m_render_webgl_frame(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    engine_box *b = unwrap(env, info, &argc, argv);
    if (!b) return NULL;
    napi_valuetype t;
    if (argc < 1 || napi_typeof(env, argv[0], &t) != napi_ok || t != napi_object) {
        napi_throw_type_error(env, NULL, "renderWebGLFrame expects an options object");
        return NULL;
    }
    const uint32_t w = (uint32_t)obj_f64(env, argv[0], "width", 256), h = (uint32_t)obj_f64(env, argv[0], "height", 256);
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 27)) {
        napi_throw_range_error(env, NULL, "renderWebGLFrame: width/height out of range");
        return NULL;
    }
    GrvGlslParams p;
    grv_glsl_params_default(w, h, obj_f64(env, argv[0], "mass", 1.0), obj_f64(env, argv[0], "spin", 0.5), &p);
    p.zoom = (float)obj_f64(env, argv[0], "zoom", p.zoom);
    p.time = (float)obj_f64(env, argv[0], "time", 0.0);
    p.max_ray_steps = (int32_t)obj_f64(env, argv[0], "maxRaySteps", p.max_ray_steps);
    p.features = (uint32_t)obj_f64(env, argv[0], "features", (double)p.features);
    double mouse[3] = {p.mouse[0], p.mouse[1], 0.0};
    obj_vec3(env, argv[0], "mouse", mouse);
    p.mouse[0] = (float)mouse[0];
    p.mouse[1] = (float)mouse[1];
    p.arith = obj_f64(env, argv[0], "fast", 0.0) != 0.0 ? GRV_ARITH_FAST : GRV_ARITH_STRICT;
    const int bloom = obj_f64(env, argv[0], "bloom", 1.0) != 0.0;
    const int moving = obj_f64(env, argv[0], "cameraMoving", 0.0) != 0.0;
    if (wants_image(env, argv[0])) {
        napi_value obj;
        image_box *ib = target_image(env, b, argv[0], w, h, &obj, "renderWebGLFrame");
        if (!ib) return NULL;
        if (grv_webgl_render_image(b->h, &p, bloom, moving, ib->img) != GRV_OK) {
            napi_throw_error(env, NULL, grv_last_error(b->h));
            return NULL;
        }
        return queued_result(env, obj, w, h);
    }
    const size_t n = (size_t)w * h * 4;
    napi_value ab, ta;
    void *dst;
    NAPI_OK(napi_create_arraybuffer(env, n * sizeof(float), &dst, &ab));
    if (grv_webgl_render_host(b->h, &p, bloom, moving, (float *)dst) != GRV_OK) {
        napi_throw_error(env, NULL, grv_last_error(b->h));
        return NULL;
    }
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, n, ab, 0, &ta));
    return ta;
}


static napi_value m_free(napi_env env, napi_callback_info info) { /* wasm-bindgen's .free() */
    size_t argc = 0;
    napi_value self;
    engine_box *box = NULL;
    NAPI_OK(napi_get_cb_info(env, info, &argc, NULL, &self, NULL));
    box = engine_of(env, self);
    if (box && box->h) {
        if (box->async_pending > 0) {
            /* works are queued on the pool: the synchronous handle goes now (the object is unusable
             * from here on), the async handles when the last work completes */
            grv_engine_destroy(box->h);
            box->h = NULL;
            if (box->multi) grv_multi_destroy(box->multi);
            box->multi = NULL;
            box->freed = 1;
        } else {
            box_destroy_handles(box);
        }
        slot_release(box);
    }
    return NULL;
}

/* default export: init() -> { memory: { buffer } }  (physics-bridge.ts:87-88) */
static napi_value f_init(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value mem, out, buf = get_arena(env);
    NAPI_OK(napi_create_object(env, &mem));
    NAPI_OK(napi_set_named_property(env, mem, "buffer", buf));
    NAPI_OK(napi_create_object(env, &out));
    NAPI_OK(napi_set_named_property(env, out, "memory", mem));
    napi_deferred d;
    napi_value promise;
    NAPI_OK(napi_create_promise(env, &d, &promise));
    napi_resolve_deferred(env, d, out);
    return promise;
}

static napi_value f_init_hooks(napi_env env, napi_callback_info info) { /* lib.rs:30-33 */
    (void)env;
    (void)info;
    return NULL;
}

#define METHOD(n, f) {n, NULL, f, NULL, NULL, NULL, napi_default, NULL}

static napi_value module_init(napi_env env, napi_value exports) {
    napi_property_descriptor props[] = {
        METHOD("update_params", m_update_params),
        METHOD("compute_horizon", m_compute_horizon),
        METHOD("compute_isco", m_compute_isco),
        METHOD("compute_photon_sphere", m_compute_photon_sphere),
        METHOD("compute_dilation", m_compute_dilation),
        METHOD("compute_g_factor", m_compute_g_factor),
        METHOD("integrate_ray_relativistic", m_integrate_ray),
        METHOD("integratePhotonGeodesic", m_integrate_ray),
        METHOD("set_ray_arith", m_set_ray_arith),
        METHOD("generate_spectrum_lut", m_generate_spectrum_lut),
        METHOD("generate_disk_lut", m_generate_disk_lut),
        METHOD("compute_shadow_curve", m_compute_shadow_curve),
        METHOD("compute_shadow_radius", m_compute_shadow_radius),
        METHOD("compute_disk_flux", m_compute_disk_flux),
        METHOD("set_camera_state", m_set_camera_state),
        METHOD("set_auto_spin", m_set_auto_spin),
        METHOD("tick_sab", m_tick_sab),
        METHOD("get_sab_ptr", m_get_sab_ptr),
        METHOD("attach_sab", m_attach_sab),
        METHOD("get_sab_layout", m_get_sab_layout),
        METHOD("compute_shadow_shift", m_compute_shadow_shift),
        METHOD("get_disk_lut_ptr", m_get_disk_lut_ptr),
        METHOD("generate_embedding_mesh", m_generate_embedding_mesh),
        METHOD("generate_ergosphere_mesh", m_generate_ergosphere_mesh),
        METHOD("compute_kretschner", m_compute_kretschner),
        METHOD("generate_curvature_field", m_generate_curvature_field),
        METHOD("compute_light_cone_tilt", m_compute_light_cone_tilt),
        METHOD("generate_tilt_field", m_generate_tilt_field),
        METHOD("compute_frame_drag_omega", m_compute_frame_drag_omega),
        METHOD("generate_frame_drag_field", m_generate_frame_drag_field),
        METHOD("compute_flamm_height", m_compute_flamm_height),
        METHOD("compute_proper_distance", m_compute_proper_distance),
        METHOD("renderFrame", m_render_frame),
        METHOD("render_frame", m_render_frame),
        METHOD("renderFrameAsync", m_render_frame_async),
        METHOD("integrate_batch", m_integrate_batch),
        METHOD("integrateBatch", m_integrate_batch),
        METHOD("integrateBatchAsync", m_integrate_batch_async),
        METHOD("renderWebGPUFrame", m_render_webgpu_frame),
        METHOD("renderWebGLFrame", m_render_webgl_frame),
        METHOD("createImage", m_create_image),
        METHOD("renderShaderFrame", m_render_shader_frame),
        METHOD("readImage", m_read_image),
        METHOD("postBloom", m_post_bloom),
        METHOD("postTaa", m_post_taa),
        METHOD("statsAccumulate", m_stats_accumulate),
        METHOD("frameStats", m_frame_stats),
        METHOD("frameStatsReset", m_frame_stats_reset),
        METHOD("synchronize", m_synchronize),
        METHOD("free", m_free),
    };
    napi_property_descriptor iprops[] = {
        METHOD("read", im_read),
        METHOD("readAsync", im_read_async),
        METHOD("stats", im_stats),
        METHOD("wait", im_wait),
        METHOD("ready", im_ready),
        METHOD("free", im_free),
    };
    napi_value icls;
    NAPI_OK(napi_define_class(env, "DeviceImage", NAPI_AUTO_LENGTH, image_ctor, NULL, sizeof iprops / sizeof iprops[0],
                              iprops, &icls));
    NAPI_OK(napi_create_reference(env, icls, 1, &g_image_ctor));
    NAPI_OK(napi_set_named_property(env, exports, "DeviceImage", icls));
    napi_value cls, fn;
    NAPI_OK(napi_define_class(env, "PhysicsEngine", NAPI_AUTO_LENGTH, engine_new, NULL,
                              sizeof props / sizeof props[0], props, &cls));
    NAPI_OK(napi_create_reference(env, cls, 1, &g_engine_ctor));
    NAPI_OK(napi_set_named_property(env, exports, "PhysicsEngine", cls));
    NAPI_OK(napi_create_function(env, "init", NAPI_AUTO_LENGTH, f_init, NULL, &fn));
    NAPI_OK(napi_set_named_property(env, exports, "default", fn));
    NAPI_OK(napi_set_named_property(env, exports, "init", fn));
    NAPI_OK(napi_create_function(env, "init_hooks", NAPI_AUTO_LENGTH, f_init_hooks, NULL, &fn));
    NAPI_OK(napi_set_named_property(env, exports, "init_hooks", fn));
    NAPI_OK(napi_create_function(env, "allocPinned", NAPI_AUTO_LENGTH, f_alloc_pinned, NULL, &fn));
    NAPI_OK(napi_set_named_property(env, exports, "allocPinned", fn));
    return exports;
}

NAPI_MODULE(blackhole_physics, module_init)
