// engine_images.hip -- device-resident images of the C ABI (include/gravitas_abi.h, "device images").
//
// What the reference's per-frame callers hold between passes is a GPU texture, not host pixels:
// WebGPURenderer.render writes its compute pass into `computeTexture`, resolves it against the history
// textures and blits (src/rendering/webgpu/renderer.ts:280-411); the physics worker only ever moves
// the 8 KB SAB block (src/workers/physics.worker.ts:111-176).  A host above this ABI (the N-API addon)
// gets the same shape here: a frame is rendered INTO a grv_image that stays in HBM, the post chain
// consumes images, and pixels cross PCIe only when grv_image_read* is called.
//
// An image's producers are queued on its compute stream and the calls return at once.  Images with
// streams of their own are written concurrently: a frame loop that alternates two of them keeps two
// frames in flight (the engine then alternates its two ray workspaces, engine_internal.hpp WorkSet) --
// the drain of one 1080p march under the head of the next.  Images created with grv_image_create_shared
// are written in queue order on one stream.  Reads (D2H) run on a copy stream per image, so the DMA of
// frame i runs under the kernels of frame i+1 either way.  Every image also keeps the counters of the
// frame that last wrote it (a 96-byte copy queued behind the frame's last kernel), so a host reads a
// frame's accepted steps without synchronising anything but that image.
#include "engine_internal.hpp"

#include <atomic>

// a compute stream images can share (grv_image_create_shared): its kernels then run in queue order
struct ImageStream {
    hipStream_t s = nullptr;
    int device = 0;
    std::atomic<int> refs{1};
};

struct grv_image {
    int device = 0;
    uint32_t w = 0, h = 0;
    float *d = nullptr;            // [h][w][4] f32
    size_t bytes = 0;
    ImageStream *cs = nullptr;     // compute stream: producers of this image (own, or shared with other images)
    hipStream_t s = nullptr;       // == cs->s
    hipStream_t copy_s = nullptr;  // D2H reads of this image (created by the first read): never behind another image's kernels
    hipEvent_t ready = nullptr;    // end of the last producer queued on s
    hipEvent_t copied = nullptr;   // end of the last read queued on copy_s
    hipEvent_t consumed = nullptr; // end of the last reader kernel that sits on ANOTHER stream
    std::atomic<bool> ready_rec{false}, copied_rec{false};
    bool consumed_rec = false;
    grvhip::FrameStatsDev *h_stats = nullptr; // pinned: counters of the frame that last wrote the image
    bool has_stats = false;
    std::string err;
};

namespace {

using namespace grvhost;

int ifail(grv_image *img, int code, const char *fmt, ...) {
    char buf[384];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (img) img->err = buf;
    return code;
}

#define IMG_HIP(img, call)                                                                          \
    do {                                                                                            \
        hipError_t _st = (call);                                                                    \
        if (_st != hipSuccess)                                                                      \
            return ifail((img), _st == hipErrorOutOfMemory ? GRV_ERR_OOM : GRV_ERR_HIP,             \
                         "%s failed: %s", #call, hipGetErrorString(_st));                           \
    } while (0)

// a producer is about to write `img` on img->s: behind the readers other streams queued on it
int begin_write(grv_engine *e, grv_image *img) {
    if (img->consumed_rec) GRV_HIP(e, hipStreamWaitEvent(img->s, img->consumed, 0));
    if (img->copied_rec) GRV_HIP(e, hipStreamWaitEvent(img->s, img->copied, 0)); // a D2H still reading it
    return GRV_OK;
}
int end_write(grv_engine *e, grv_image *img) {
    GRV_HIP(e, hipEventRecord(img->ready, img->s));
    img->ready_rec = true;
    return GRV_OK;
}
// a kernel queued on `s` is about to read `src`
int begin_read(grv_engine *e, const grv_image *src, hipStream_t s) {
    if (src->ready_rec && src->s != s) GRV_HIP(e, hipStreamWaitEvent(s, src->ready, 0));
    return GRV_OK;
}
int end_read(grv_engine *e, grv_image *src, hipStream_t s) {
    if (src->s == s) return GRV_OK;
    // chained: the newest record then stands for every earlier reader as well
    if (src->consumed_rec) GRV_HIP(e, hipStreamWaitEvent(s, src->consumed, 0));
    GRV_HIP(e, hipEventRecord(src->consumed, s));
    src->consumed_rec = true;
    return GRV_OK;
}

// The renderer layer's state (history ping-pong, frame counter, bloom scratch) is one per engine and
// frames follow each other through it: calls arriving on different image streams are chained.
int chain_begin(grv_engine *e, hipStream_t s) {
    if (!e->chain_done) GRV_HIP(e, hipEventCreateWithFlags(&e->chain_done, hipEventDisableTiming));
    if (e->chain_rec) GRV_HIP(e, hipStreamWaitEvent(s, e->chain_done, 0));
    return GRV_OK;
}
int chain_end(grv_engine *e, hipStream_t s) {
    GRV_HIP(e, hipEventRecord(e->chain_done, s));
    e->chain_rec = true;
    return GRV_OK;
}

bool same_device(const grv_engine *e, const grv_image *img) { return e && img && e->device == img->device; }

// the frame that was just queued on img->s through engine `e`: keep its counters with the image
int snapshot_stats(grv_engine *e, grv_image *img) {
    GRV_HIP(e, hipMemcpyAsync(img->h_stats, e->d_stats, sizeof(FrameStatsDev), hipMemcpyDeviceToHost, img->s));
    img->has_stats = true;
    if (!e->stats_accum) {
        // the block's next user clears it behind stats_done[b]: move that event behind the copy
        const int b = (int)(e->d_stats - e->stats_blocks);
        GRV_HIP(e, hipEventRecord(e->stats_done[b], img->s));
    }
    return GRV_OK;
}

} // namespace

extern "C" {

static int image_create(grv_engine *e, uint32_t width, uint32_t height, grv_image *stream_of, grv_image **out) {
    if (!e) return GRV_ERR_INVALID;
    if (!out) return fail(e, GRV_ERR_INVALID, "null argument");
    *out = nullptr;
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 27))
        return fail(e, GRV_ERR_INVALID, "grv_image_create: %u x %u out of range", width, height);
    if (stream_of && stream_of->device != e->device)
        return fail(e, GRV_ERR_INVALID, "grv_image_create_shared: the stream's image lives on device %d, the engine on %d",
                    stream_of->device, e->device);
    GRV_HIP(e, hipSetDevice(e->device));
    grv_image *img = new (std::nothrow) grv_image();
    if (!img) return fail(e, GRV_ERR_OOM, "grv_image_create: out of host memory");
    img->device = e->device;
    img->w = width;
    img->h = height;
    img->bytes = (size_t)width * height * 4 * sizeof(float);
    hipError_t st = hipMalloc(reinterpret_cast<void **>(&img->d), img->bytes);
    if (st == hipSuccess) {
        if (stream_of) {
            img->cs = stream_of->cs;
            img->cs->refs.fetch_add(1);
        } else {
            img->cs = new (std::nothrow) ImageStream();
            if (!img->cs) st = hipErrorOutOfMemory;
            else {
                img->cs->device = e->device;
                st = hipStreamCreateWithFlags(&img->cs->s, hipStreamNonBlocking);
            }
        }
    }
    if (st == hipSuccess) img->s = img->cs->s;
    // blocking-sync events: a host thread waiting for an image sleeps instead of spinning
    if (st == hipSuccess) st = hipEventCreateWithFlags(&img->ready, hipEventDisableTiming | hipEventBlockingSync);
    if (st == hipSuccess) st = hipEventCreateWithFlags(&img->copied, hipEventDisableTiming | hipEventBlockingSync);
    if (st == hipSuccess) st = hipEventCreateWithFlags(&img->consumed, hipEventDisableTiming | hipEventBlockingSync);
    if (st == hipSuccess) st = hipHostMalloc(reinterpret_cast<void **>(&img->h_stats), sizeof(FrameStatsDev), hipHostMallocDefault);
    if (st != hipSuccess) {
        grv_image_destroy(img);
        return fail(e, st == hipErrorOutOfMemory ? GRV_ERR_OOM : GRV_ERR_HIP, "grv_image_create(%u x %u): %s", width,
                    height, hipGetErrorString(st));
    }
    std::memset(img->h_stats, 0, sizeof(FrameStatsDev));
    // A new image is black, and its compute stream has carried work before the first frame does: the runtime sets a
    // stream's hardware queue up at its first use, and a first use in the middle of a burst of frames held every
    // synchronising call of the process -- a worker's control-plane call on its own high-priority stream included --
    // until the frames in flight had finished (measured: 41-57 ms once per new image stream, 1 ms afterwards;
    // profiles/EXPERIMENTS.md T, addendum).
    st = hipMemsetAsync(img->d, 0, img->bytes, img->s);
    if (st == hipSuccess) st = hipStreamSynchronize(img->s);
    if (st != hipSuccess) {
        grv_image_destroy(img);
        return fail(e, GRV_ERR_HIP, "grv_image_create(%u x %u): %s", width, height, hipGetErrorString(st));
    }
    *out = img;
    return GRV_OK;
}

int grv_image_create(grv_engine *e, uint32_t width, uint32_t height, grv_image **out) {
    return image_create(e, width, height, nullptr, out);
}
int grv_image_create_shared(grv_engine *e, uint32_t width, uint32_t height, grv_image *stream_of, grv_image **out) {
    if (e && !stream_of) return fail(e, GRV_ERR_INVALID, "null argument");
    return image_create(e, width, height, stream_of, out);
}

void grv_image_destroy(grv_image *img) {
    if (!img) return;
    (void)hipSetDevice(img->device);
    // everything that touches THIS image (not the later work of images sharing its stream)
    if (img->ready_rec) (void)hipEventSynchronize(img->ready);
    if (img->copied_rec) (void)hipEventSynchronize(img->copied);
    if (img->consumed_rec) (void)hipEventSynchronize(img->consumed);
    if (img->d) (void)hipFree(img->d);
    if (img->copy_s) (void)hipStreamDestroy(img->copy_s);
    if (img->cs && img->cs->refs.fetch_sub(1) == 1) {
        if (img->cs->s) {
            (void)hipStreamSynchronize(img->cs->s);
            (void)hipStreamDestroy(img->cs->s);
        }
        delete img->cs;
    }
    if (img->ready) (void)hipEventDestroy(img->ready);
    if (img->copied) (void)hipEventDestroy(img->copied);
    if (img->consumed) (void)hipEventDestroy(img->consumed);
    if (img->h_stats) (void)hipHostFree(img->h_stats);
    delete img;
}

uint32_t grv_image_width(const grv_image *img) { return img ? img->w : 0u; }
uint32_t grv_image_height(const grv_image *img) { return img ? img->h : 0u; }
size_t grv_image_bytes(const grv_image *img) { return img ? img->bytes : 0u; }
float *grv_image_data(grv_image *img) { return img ? img->d : nullptr; }
void *grv_image_stream(grv_image *img) { return img ? static_cast<void *>(img->s) : nullptr; }
const char *grv_image_last_error(const grv_image *img) { return img ? img->err.c_str() : "null image"; }

int grv_render_frame_image(grv_engine *e, const GrvCamera *cam, const GrvRenderParams *p, grv_image *img) {
    if (!e) return GRV_ERR_INVALID;
    if (!cam || !p || !img) return fail(e, GRV_ERR_INVALID, "null argument");
    if (!same_device(e, img)) return fail(e, GRV_ERR_INVALID, "image lives on device %d, engine on %d", img->device, e->device);
    if (p->tile_world > 1) return fail(e, GRV_ERR_INVALID, "an image holds a whole frame (tile_world must be 0 or 1)");
    if (p->width != img->w || p->height != img->h)
        return fail(e, GRV_ERR_INVALID, "frame %u x %u into an image of %u x %u", p->width, p->height, img->w, img->h);
    int rc = begin_write(e, img);
    if (rc != GRV_OK) return rc;
    GrvFrameBuffers fb{};
    fb.rgba = img->d;
    rc = grv_render_frame_device(e, cam, p, &fb, img->s);
    if (rc != GRV_OK) {
        (void)end_write(e, img); // kernels of the failed frame may be queued: a later wait / read / destroy covers them
        return rc;
    }
    rc = snapshot_stats(e, img);
    const int rc2 = end_write(e, img);
    return rc != GRV_OK ? rc : rc2;
}

int grv_render_frame_glsl_image(grv_engine *e, const GrvGlslParams *p, grv_image *img) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !img) return fail(e, GRV_ERR_INVALID, "null argument");
    if (!same_device(e, img)) return fail(e, GRV_ERR_INVALID, "image lives on device %d, engine on %d", img->device, e->device);
    if (p->tile_world > 1) return fail(e, GRV_ERR_INVALID, "an image holds a whole frame (tile_world must be 0 or 1)");
    if (p->width != img->w || p->height != img->h)
        return fail(e, GRV_ERR_INVALID, "frame %u x %u into an image of %u x %u", p->width, p->height, img->w, img->h);
    int rc = begin_write(e, img);
    if (rc != GRV_OK) return rc;
    rc = grv_render_frame_glsl(e, p, img->d, nullptr, nullptr, img->s);
    if (rc != GRV_OK) {
        (void)end_write(e, img); // kernels of the failed frame may be queued: a later wait / read / destroy covers them
        return rc;
    }
    rc = snapshot_stats(e, img);
    const int rc2 = end_write(e, img);
    return rc != GRV_OK ? rc : rc2;
}

int grv_render_frame_wgsl_image(grv_engine *e, const GrvWgslParams *p, grv_image *img) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !img) return fail(e, GRV_ERR_INVALID, "null argument");
    if (!same_device(e, img)) return fail(e, GRV_ERR_INVALID, "image lives on device %d, engine on %d", img->device, e->device);
    if (p->tile_world > 1) return fail(e, GRV_ERR_INVALID, "an image holds a whole frame (tile_world must be 0 or 1)");
    if (p->width != img->w || p->height != img->h)
        return fail(e, GRV_ERR_INVALID, "frame %u x %u into an image of %u x %u", p->width, p->height, img->w, img->h);
    int rc = begin_write(e, img);
    if (rc != GRV_OK) return rc;
    rc = grv_render_frame_wgsl(e, p, img->d, nullptr, nullptr, img->s);
    if (rc != GRV_OK) {
        (void)end_write(e, img); // kernels of the failed frame may be queued: a later wait / read / destroy covers them
        return rc;
    }
    rc = snapshot_stats(e, img);
    const int rc2 = end_write(e, img);
    return rc != GRV_OK ? rc : rc2;
}

int grv_webgl_render_image(grv_engine *e, const GrvGlslParams *p, int32_t bloom_enabled, int32_t camera_moving,
                           grv_image *screen) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !screen) return fail(e, GRV_ERR_INVALID, "null argument");
    if (!same_device(e, screen)) return fail(e, GRV_ERR_INVALID, "image lives on another device");
    if (p->width != screen->w || p->height != screen->h)
        return fail(e, GRV_ERR_INVALID, "frame %u x %u into an image of %u x %u", p->width, p->height, screen->w, screen->h);
    int rc = begin_write(e, screen);
    if (rc == GRV_OK) rc = chain_begin(e, screen->s);
    if (rc != GRV_OK) return rc;
    rc = grv_webgl_render(e, p, bloom_enabled, camera_moving, screen->d, screen->s);
    // (chained even after a failure: kernels of the chain may already be queued)
    const int rc2 = chain_end(e, screen->s);
    if (rc == GRV_OK) rc = rc2;
    if (rc == GRV_OK) rc = snapshot_stats(e, screen);
    const int rc3 = end_write(e, screen); // on the error paths too: a later wait / read / destroy covers what is queued
    return rc != GRV_OK ? rc : rc3;
}

int grv_webgpu_render_image(grv_engine *e, const float camera_uniforms[88], const float physics_params[8],
                            int32_t max_steps, int32_t arith, grv_image *screen) {
    if (!e) return GRV_ERR_INVALID;
    if (!camera_uniforms || !physics_params || !screen) return fail(e, GRV_ERR_INVALID, "null argument");
    if (!same_device(e, screen)) return fail(e, GRV_ERR_INVALID, "image lives on another device");
    if ((uint32_t)physics_params[2] != screen->w || (uint32_t)physics_params[3] != screen->h)
        return fail(e, GRV_ERR_INVALID, "frame %u x %u into an image of %u x %u", (uint32_t)physics_params[2],
                    (uint32_t)physics_params[3], screen->w, screen->h);
    int rc = begin_write(e, screen);
    if (rc == GRV_OK) rc = chain_begin(e, screen->s);
    if (rc != GRV_OK) return rc;
    rc = grv_webgpu_render(e, camera_uniforms, physics_params, max_steps, arith, screen->d, screen->s);
    const int rc2 = chain_end(e, screen->s);
    if (rc == GRV_OK) rc = rc2;
    if (rc == GRV_OK) rc = snapshot_stats(e, screen);
    const int rc3 = end_write(e, screen); // on the error paths too: a later wait / read / destroy covers what is queued
    return rc != GRV_OK ? rc : rc3;
}

int grv_post_bloom_image(grv_engine *e, const GrvBloomParams *p, grv_image *scene, grv_image *out) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !scene || !out) return fail(e, GRV_ERR_INVALID, "null argument");
    if (scene == out) return fail(e, GRV_ERR_INVALID, "bloom: out aliases the scene");
    if (!same_device(e, scene) || !same_device(e, out)) return fail(e, GRV_ERR_INVALID, "image lives on another device");
    if (p->width != scene->w || p->height != scene->h || out->w != scene->w || out->h != scene->h)
        return fail(e, GRV_ERR_INVALID, "bloom: image sizes differ from the parameters");
    int rc = begin_write(e, out);
    if (rc == GRV_OK) rc = begin_read(e, scene, out->s);
    if (rc == GRV_OK) rc = chain_begin(e, out->s); // the bloom scratch targets are one per engine
    if (rc != GRV_OK) return rc;
    rc = grv_post_bloom(e, p, scene->d, out->d, out->s);
    const int rc2 = chain_end(e, out->s);
    const int rc3 = end_read(e, scene, out->s);
    out->has_stats = false;
    const int rc4 = end_write(e, out); // on the error paths too
    if (rc != GRV_OK) return rc;
    if (rc2 != GRV_OK) return rc2;
    return rc3 != GRV_OK ? rc3 : rc4;
}

int grv_post_taa_resolve_image(grv_engine *e, const GrvTaaParams *p, grv_image *current, grv_image *history,
                               grv_image *out) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !current || !history || !out) return fail(e, GRV_ERR_INVALID, "null argument");
    if (out == current || out == history) return fail(e, GRV_ERR_INVALID, "taa: out aliases an input");
    if (!same_device(e, current) || !same_device(e, history) || !same_device(e, out))
        return fail(e, GRV_ERR_INVALID, "image lives on another device");
    for (const grv_image *q : {current, history, out})
        if (q->w != p->width || q->h != p->height) return fail(e, GRV_ERR_INVALID, "taa: image sizes differ from the parameters");
    int rc = begin_write(e, out);
    if (rc == GRV_OK) rc = begin_read(e, current, out->s);
    if (rc == GRV_OK) rc = begin_read(e, history, out->s);
    if (rc != GRV_OK) return rc;
    rc = grv_post_taa_resolve(e, p, current->d, history->d, out->d, out->s);
    const int rc2 = end_read(e, current, out->s);
    const int rc3 = end_read(e, history, out->s);
    out->has_stats = false;
    const int rc4 = end_write(e, out); // on the error paths too
    if (rc != GRV_OK) return rc;
    if (rc2 != GRV_OK) return rc2;
    return rc3 != GRV_OK ? rc3 : rc4;
}

// ---- reads: these touch the image only (its stream, its events), never an engine: a host may
// wait for / read one image on one thread while another thread queues frames on the engine ----

int grv_image_read_async(grv_image *img, float *host, size_t elems) {
    if (!img) return GRV_ERR_INVALID;
    if (!host) return ifail(img, GRV_ERR_INVALID, "null argument");
    if (elems > img->bytes / sizeof(float)) return ifail(img, GRV_ERR_INVALID, "read of %zu floats from an image of %zu", elems, img->bytes / sizeof(float));
    IMG_HIP(img, hipSetDevice(img->device));
    // on the image's own copy stream, behind its last producer: the compute stream (perhaps shared with
    // other images) goes on with the next frame's kernels while the DMA runs
    if (!img->copy_s) IMG_HIP(img, hipStreamCreateWithFlags(&img->copy_s, hipStreamNonBlocking));
    if (img->ready_rec) IMG_HIP(img, hipStreamWaitEvent(img->copy_s, img->ready, 0));
    IMG_HIP(img, hipMemcpyAsync(host, img->d, elems * sizeof(float), hipMemcpyDeviceToHost, img->copy_s));
    IMG_HIP(img, hipEventRecord(img->copied, img->copy_s));
    img->copied_rec = true;
    return GRV_OK;
}

int grv_image_wait(grv_image *img) {
    if (!img) return GRV_ERR_INVALID;
    IMG_HIP(img, hipSetDevice(img->device));
    if (img->ready_rec) IMG_HIP(img, hipEventSynchronize(img->ready));
    if (img->copied_rec) IMG_HIP(img, hipEventSynchronize(img->copied));
    return GRV_OK;
}

// The blocking read waits for the producers on the HOST and only then queues the copy: its copy stream
// never holds a barrier packet.  (The runtime multiplexes streams onto a few hardware queues; a copy stream
// that shares one with a compute stream would otherwise park its "wait for the frame" barrier in front of
// that stream's next kernels -- measured: every other frame of a rotation started a copy late.)
int grv_image_read(grv_image *img, float *host, size_t elems) {
    if (!img) return GRV_ERR_INVALID;
    if (!host) return ifail(img, GRV_ERR_INVALID, "null argument");
    if (elems > img->bytes / sizeof(float)) return ifail(img, GRV_ERR_INVALID, "read of %zu floats from an image of %zu", elems, img->bytes / sizeof(float));
    IMG_HIP(img, hipSetDevice(img->device));
    if (img->ready_rec) IMG_HIP(img, hipEventSynchronize(img->ready));
    if (!img->copy_s) IMG_HIP(img, hipStreamCreateWithFlags(&img->copy_s, hipStreamNonBlocking));
    IMG_HIP(img, hipMemcpyAsync(host, img->d, elems * sizeof(float), hipMemcpyDeviceToHost, img->copy_s));
    IMG_HIP(img, hipEventRecord(img->copied, img->copy_s));
    img->copied_rec = true;
    IMG_HIP(img, hipEventSynchronize(img->copied));
    return GRV_OK;
}

int grv_image_query(grv_image *img) {
    if (!img) return -GRV_ERR_INVALID;
    if (hipSetDevice(img->device) != hipSuccess) return -GRV_ERR_HIP;
    for (int k = 0; k < 2; ++k) {
        if (!(k == 0 ? img->ready_rec : img->copied_rec)) continue;
        const hipError_t st = hipEventQuery(k == 0 ? img->ready : img->copied);
        if (st == hipErrorNotReady) return 0;
        if (st != hipSuccess) {
            img->err = std::string("hipEventQuery: ") + hipGetErrorString(st);
            return -GRV_ERR_HIP;
        }
    }
    return 1;
}

int grv_image_frame_stats(grv_image *img, GrvFrameStats *stats) {
    if (!img) return GRV_ERR_INVALID;
    if (!stats) return ifail(img, GRV_ERR_INVALID, "null argument");
    if (!img->has_stats) return ifail(img, GRV_ERR_INVALID, "no frame was rendered into this image (or a post pass wrote it last)");
    IMG_HIP(img, hipSetDevice(img->device));
    IMG_HIP(img, hipEventSynchronize(img->ready)); // the counters' copy sits before it on the compute stream
    std::memset(stats, 0, sizeof *stats);
    const FrameStatsDev &d = *img->h_stats;
    stats->rays = d.rays;
    stats->accepted_steps = stats_total_steps(d);
    stats->rkf_tries = d.rkf_tries;
    for (int k = 0; k < 5; ++k) stats->term_count[k] = d.term_count[k];
    stats->crossings = d.crossings;
    double md;
    std::memcpy(&md, &d.max_drift_bits, sizeof md);
    stats->max_drift = md;
    return GRV_OK;
}

} // extern "C"
