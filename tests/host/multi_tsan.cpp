// multi_tsan.cpp -- the multi-GPU host logic (blackhole-simulation_amd/csrc/multi_core.hpp: RankThreads, the
// exchange buffers of both frame parities, the frame skeleton with its events) under ThreadSanitizer on the CPU,
// over a mock Api whose streams are REAL threads that execute their queues in order and whose copies and
// "kernels" touch REAL memory.  An event that is missing between two streams is therefore a data race the
// detector reports, and a slot that is reused too early shows up as a wrong pixel: every frame renders
// value(frame, x, y) and every assembled image is compared with it on the caller's stream.
//
// Build + run: oracle/sanitize_host.sh (clang++ -fsanitize=thread), or plain (no sanitizer) from
// tests/test_multi_host_logic.py.  usage: multi_tsan [frames_per_combination]
// Exercised: G = 2, 4, 8 ranks (+ G = 1 under the self-exchange hook), peer-copy and RCCL-shaped transports,
// RGBA32F and RGBA16F exchange, four caller streams (frames of both parities in flight), a changing frame size (buffer
// regrowth), and the three injected faults of grv_multi_test_inject_fault with the frame after each.
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>

#include "../../blackhole-simulation_amd/csrc/multi_core.hpp"

namespace {

// ---- a stream: one thread, one in-order queue ---------------------------------------------------------------
struct MockStream {
    std::mutex mu;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    std::thread th;
    MockStream() : th([this] { loop(); }) {}
    ~MockStream() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void push(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(std::move(f));
        }
        cv.notify_all();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(mu);
        idle.wait(lk, [&] { return q.empty() && !busy; });
    }
    void loop() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            f();
            {
                std::lock_guard<std::mutex> lk(mu);
                busy = false;
            }
            idle.notify_all();
        }
    }
};

// ---- an event: HIP semantics -- a wait refers to the record that was the latest when the wait was queued -----
struct MockEvent {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;
};

// ---- one send / receive pair of a group --------------------------------------------------------------------
struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    bool src_ready = false, done = false;
};

inline uint16_t to_half_bits(float x) { // any deterministic narrowing will do for the mock; keep 16 high bits
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return (uint16_t)(u >> 16);
}
inline float from_half_bits(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float x;
    std::memcpy(&x, &u, 4);
    return x;
}

uint32_t tile_pitch(uint32_t width, uint32_t world) { // engine.hip tile_pitch: first pitch >= tiles across coprime with G
    uint32_t p = (width + 63u) / 64u;
    if (world <= 1) return p;
    for (;; ++p) {
        uint32_t a = p, b = world;
        while (b) {
            const uint32_t t = a % b;
            a = b;
            b = t;
        }
        if (a == 1u) return p;
    }
}
uint32_t tiles_local(uint32_t width, uint32_t height, int G, int r) {
    const uint32_t total = tile_pitch(width, (uint32_t)G) * ((height + 63u) / 64u);
    return total > (uint32_t)r ? (total - (uint32_t)r + (uint32_t)G - 1u) / (uint32_t)G : 0u;
}

struct MockApi {
    using Stream = MockStream *;
    using Event = MockEvent *;
    std::string msg;
    const char *error_text() const { return msg.c_str(); }
    int set_device(int) { return 0; }
    int device_synchronize() {
        for (auto *s : all_streams) s->sync();
        return 0;
    }
    int malloc(void **p, size_t bytes) {
        *p = std::calloc(1, bytes ? bytes : 1);
        return *p ? 0 : 4;
    }
    void free(void *p) { std::free(p); }
    int stream_wait_event(Stream s, Event e) {
        uint64_t target;
        {
            std::lock_guard<std::mutex> lk(e->mu);
            target = e->recorded;
        }
        s->push([e, target] {
            std::unique_lock<std::mutex> lk(e->mu);
            e->cv.wait(lk, [&] { return e->completed >= target; });
        });
        return 0;
    }
    int event_record(Event e, Stream s) {
        uint64_t gen;
        {
            std::lock_guard<std::mutex> lk(e->mu);
            gen = ++e->recorded;
        }
        s->push([e, gen] {
            {
                std::lock_guard<std::mutex> lk(e->mu);
                if (e->completed < gen) e->completed = gen;
            }
            e->cv.notify_all();
        });
        return 0;
    }
    int copy_to_rank0(void *dst, int, const void *src, int, size_t bytes, Stream s) {
        s->push([=] { std::memcpy(dst, src, bytes); });
        return 0;
    }
    int copy_on_device(void *dst, const void *src, size_t bytes, Stream s) {
        s->push([=] { std::memcpy(dst, src, bytes); });
        return 0;
    }
    int pack_half(const float *src, void *dst, size_t n_px, Stream s) {
        s->push([=] {
            uint16_t *d = static_cast<uint16_t *>(dst);
            for (size_t i = 0; i < n_px * 4; ++i) d[i] = to_half_bits(src[i]);
        });
        return 0;
    }
    int widen_half(const void *src, float *dst, size_t n_px, Stream s) {
        s->push([=] {
            const uint16_t *h = static_cast<const uint16_t *>(src);
            for (size_t i = 0; i < n_px * 4; ++i) dst[i] = from_half_bits(h[i]);
        });
        return 0;
    }
    int quantize(float *img, size_t n_px, Stream s) {
        s->push([=] {
            for (size_t i = 0; i < n_px * 4; ++i) img[i] = from_half_bits(to_half_bits(img[i]));
        });
        return 0;
    }
    int unpack_tiles(uint32_t width, uint32_t height, int G, int r, const void *slot, float *image, bool half, Stream s) {
        s->push([=] {
            const uint32_t pitch = tile_pitch(width, (uint32_t)G), n = tiles_local(width, height, G, r);
            for (uint32_t tl = 0; tl < n; ++tl) {
                const uint32_t tile = tl * (uint32_t)G + (uint32_t)r, tx = tile % pitch, ty = tile / pitch;
                for (uint32_t py = 0; py < 64; ++py)
                    for (uint32_t px = 0; px < 64; ++px) {
                        const uint32_t X = tx * 64 + px, Y = ty * 64 + py;
                        if (X >= width || Y >= height) continue;
                        const size_t k = (size_t)tl * 4096u + py * 64u + px;
                        float *d = image + ((size_t)Y * width + X) * 4;
                        if (half) {
                            const uint16_t *h = static_cast<const uint16_t *>(slot) + k * 4;
                            for (int c = 0; c < 4; ++c) d[c] = from_half_bits(h[c]);
                        } else {
                            std::memcpy(d, static_cast<const float *>(slot) + k * 4, 16);
                        }
                    }
            }
        });
        return 0;
    }
    size_t share_pixels(uint32_t width, uint32_t height, int G, int r) {
        return G <= 1 ? (size_t)width * height : (size_t)tiles_local(width, height, G, r) * 4096u;
    }
    size_t slot_pixels(uint32_t width, uint32_t height, int G) {
        const size_t total = (size_t)tile_pitch(width, (uint32_t)G) * ((height + 63u) / 64u);
        return (total + (size_t)G - 1) / (size_t)G * 4096u;
    }
    // RCCL-shaped group: a send blocks its stream until the matching receive has copied the data
    struct Pending {
        const void *src = nullptr;
        void *dst = nullptr;
        size_t bytes = 0;
        Stream ss = nullptr, rs = nullptr;
    };
    std::vector<Pending> group;
    bool group_open = false;
    int groups_closed = 0;
    int group_start() {
        if (group_open) {
            msg = "group already open";
            return 3;
        }
        group_open = true;
        group.clear();
        return 0;
    }
    int send(const void *src, size_t n, bool half, int r, Stream s) {
        if (!group_open) return 3;
        Pending p;
        p.src = src;
        p.bytes = n * (half ? 2u : 4u);
        p.ss = s;
        group.push_back(p);
        (void)r;
        return 0;
    }
    int recv(void *dst, size_t n, bool half, int, Stream s) {
        if (!group_open || group.empty() || group.back().dst) return 3;
        group.back().dst = dst;
        group.back().rs = s;
        assert(group.back().bytes == n * (half ? 2u : 4u));
        return 0;
    }
    int group_end() {
        if (!group_open) return 3;
        group_open = false;
        ++groups_closed;
        for (const Pending &p : group) {
            if (!p.dst) continue; // a send whose receive never came (failed mid-pair): dropped with the group
            auto rv = std::make_shared<Rendezvous>();
            p.ss->push([rv] {
                std::unique_lock<std::mutex> lk(rv->mu);
                rv->src_ready = true;
                rv->cv.notify_all();
                rv->cv.wait(lk, [&] { return rv->done; });
            });
            const Pending q = p;
            p.rs->push([rv, q] {
                std::unique_lock<std::mutex> lk(rv->mu);
                rv->cv.wait(lk, [&] { return rv->src_ready; });
                std::memcpy(q.dst, q.src, q.bytes);
                rv->done = true;
                rv->cv.notify_all();
            });
        }
        group.clear();
        return 0;
    }
    std::vector<MockStream *> all_streams;
};

float value(uint64_t frame, uint32_t x, uint32_t y, int c) {
    // exactly representable in the mock's 16-bit narrowing: small integers
    return (float)((frame * 7u + x * 3u + y * 5u + (uint32_t)c) % 251u);
}

struct Harness {
    grvmulti::Core<MockApi> core;
    std::vector<std::unique_ptr<MockStream>> streams;
    std::vector<std::unique_ptr<MockEvent>> events;
    MockStream *stream() {
        streams.emplace_back(new MockStream());
        core.api.all_streams.push_back(streams.back().get());
        return streams.back().get();
    }
    MockEvent *event() {
        events.emplace_back(new MockEvent());
        return events.back().get();
    }
    Harness(int G, int transport, int format, bool self_exchange) {
        core.G = G;
        core.transport = transport;
        core.format = format;
        core.self_exchange = self_exchange;
        core.dev.assign(G, 0);
        for (int r = 0; r < G; ++r) core.dev[r] = r;
        core.rank.resize(G);
        for (int r = 0; r < G; ++r)
            for (int b = 0; b < 2; ++b) {
                core.rank[r].s[b] = stream();
                core.rank[r].arrived[b] = event();
            }
        for (int b = 0; b < 2; ++b) {
            core.rs[b] = stream();
            core.unpacked[b] = event();
        }
        core.caller_ready = event();
        core.threads = new grvmulti::RankThreads(G);
    }
    ~Harness() {
        delete core.threads;
        core.api.device_synchronize();
        core.drop_buffers();
    }
};

std::atomic<long> g_bad_pixels{0};
std::atomic<long> g_frames_checked{0};

int run_combo(int G, int transport, int format, bool self_exchange, int frames) {
    // four caller streams / images in rotation: frame i waits (on the host) for frame i - 4 only, so frames i and
    // i - 2 -- the two users of one parity's slots and events -- really are in flight together
    constexpr int NC = 4;
    std::vector<float> image[NC]; // (outlive the harness: its streams write into them until they are joined)
    Harness H(G, transport, format, self_exchange);
    auto &core = H.core;
    MockStream *caller[NC] = {H.stream(), H.stream(), H.stream(), H.stream()};
    int faults_seen = 0, faults_wanted = 0;
    uint64_t rendered = 0; // frames that went through (== core.frame)
    for (int i = 0; i < frames; ++i) {
        const uint32_t W = (i / 40) % 2 ? 200u : 136u, Hh = (i / 40) % 2 ? 130u : 70u; // regrows the buffers now and then
        const int cb = i % NC;
        // the caller owns the image until its own stream has checked it: wait for that stream before resizing it
        caller[cb]->sync();
        image[cb].assign((size_t)W * Hh * 4, -1.0f);
        float *img = image[cb].data();
        int fk = grvmulti::FAULT_NONE, fr = 0;
        if (i % 23 == 11) {
            fk = transport == grvmulti::TRANSPORT_RCCL ? (i % 2 ? grvmulti::FAULT_SEND : grvmulti::FAULT_RENDER)
                                                       : (i % 2 ? grvmulti::FAULT_PEER_COPY : grvmulti::FAULT_RENDER);
            // a rank that holds tiles of this frame (small frames leave the last ranks of G = 8 empty-handed)
            const uint32_t total_tiles = tile_pitch(W, (uint32_t)G) * ((Hh + 63u) / 64u);
            const int holders = (int)(total_tiles < (uint32_t)G ? total_tiles : (uint32_t)G);
            fr = holders > 1 ? 1 + (i % (holders - 1)) : 0;
            if ((fk == grvmulti::FAULT_SEND || fk == grvmulti::FAULT_PEER_COPY) && G == 1 && !self_exchange) fk = grvmulti::FAULT_RENDER;
            core.fault_kind = fk;
            core.fault_rank = fr;
            ++faults_wanted;
        }
        const uint64_t fid = rendered;
        const int rc = core.run_frame(
            W, Hh, img, caller[cb],
            [&, W, Hh, fid](int r, float *target, MockStream *s) -> int {
                const int Gn = core.G;
                const bool whole = (Gn == 1 && !core.self_exchange);
                s->push([=] {
                    if (whole || Gn == 1) { // row-major whole frame (one rank)
                        for (uint32_t y = 0; y < Hh; ++y)
                            for (uint32_t x = 0; x < W; ++x)
                                for (int c = 0; c < 4; ++c) target[((size_t)y * W + x) * 4 + c] = value(fid, x, y, c);
                        return;
                    }
                    const uint32_t pitch = tile_pitch(W, (uint32_t)Gn), n = tiles_local(W, Hh, Gn, r);
                    for (uint32_t tl = 0; tl < n; ++tl) {
                        const uint32_t tile = tl * (uint32_t)Gn + (uint32_t)r, tx = tile % pitch, ty = tile / pitch;
                        for (uint32_t py = 0; py < 64; ++py)
                            for (uint32_t px = 0; px < 64; ++px)
                                for (int c = 0; c < 4; ++c)
                                    target[((size_t)tl * 4096u + py * 64u + px) * 4 + c] = value(fid, tx * 64 + px, ty * 64 + py, c);
                    }
                });
                return 0;
            },
            [](int) { return std::string("mock render"); });
        if (fk != grvmulti::FAULT_NONE) {
            if (rc == 0) {
                std::fprintf(stderr, "G=%d t=%d: injected fault %d on rank %d went unnoticed\n", G, transport, fk, fr);
                return 1;
            }
            if (core.err.find("injected") == std::string::npos || core.api.group_open) {
                std::fprintf(stderr, "G=%d t=%d: fault %d: err='%s' group_open=%d\n", G, transport, fk, core.err.c_str(), (int)core.api.group_open);
                return 1;
            }
            ++faults_seen;
            continue; // the next loop iteration renders the following frame on the same handle
        }
        if (rc != 0) {
            std::fprintf(stderr, "G=%d t=%d frame %d failed: %s\n", G, transport, i, core.err.c_str());
            return 1;
        }
        ++rendered;
        // on the caller's stream, behind the frame: compare the assembled image with what was rendered
        caller[cb]->push([=] {
            long bad = 0;
            for (uint32_t y = 0; y < Hh; ++y)
                for (uint32_t x = 0; x < W; ++x)
                    for (int c = 0; c < 4; ++c)
                        if (img[((size_t)y * W + x) * 4 + c] != value(fid, x, y, c)) ++bad;
            g_bad_pixels += bad;
            g_frames_checked += 1;
        });
    }
    for (auto *c : caller) c->sync();
    if (faults_seen != faults_wanted) return 1;
    return 0;
}

} // namespace

int main(int argc, char **argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 1000;
    int failed = 0, combos = 0;
    for (int G : {2, 4, 8})
        for (int transport : {grvmulti::TRANSPORT_PEER_COPY, grvmulti::TRANSPORT_RCCL})
            for (int format : {grvmulti::FORMAT_RGBA32F, grvmulti::FORMAT_RGBA16F}) {
                failed += run_combo(G, transport, format, false, frames);
                ++combos;
            }
    // one rank walking the whole transport path (grv_multi_test_self_exchange), and the plain one-rank handle
    failed += run_combo(1, grvmulti::TRANSPORT_RCCL, grvmulti::FORMAT_RGBA32F, true, frames / 4 + 1);
    failed += run_combo(1, grvmulti::TRANSPORT_PEER_COPY, grvmulti::FORMAT_RGBA16F, true, frames / 4 + 1);
    failed += run_combo(1, grvmulti::TRANSPORT_PEER_COPY, grvmulti::FORMAT_RGBA32F, false, frames / 4 + 1);
    combos += 3;
    std::printf("{\"combinations\": %d, \"frames_per_combination\": %d, \"frames_checked\": %ld, \"bad_pixels\": %ld, \"failed\": %d}\n",
                combos, frames, g_frames_checked.load(), g_bad_pixels.load(), failed);
    return (failed || g_bad_pixels.load()) ? 1 : 0;
}
