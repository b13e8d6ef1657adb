// multi_core.hpp -- the host logic of the multi-GPU handle (include/gravitas_abi.h grv_multi), written against
// an `Api` policy instead of the HIP / RCCL runtimes so that the SAME code runs
//   * in the product, over HipApi (engine_multi.hip): streams, events, peer copies, ncclSend / ncclRecv;
//   * on the CPU box under ThreadSanitizer, over a mock Api whose streams are real threads executing their
//     queues in order and whose "kernels" touch real memory (tests/host/multi_tsan.cpp, oracle/sanitize_host.sh):
//     a missing event between two streams is then a data race TSan reports, and the rank threads themselves
//     (RankThreads) run under the detector with frames in flight.
//
// What lives here: one worker thread per rank (RankThreads), the exchange buffers of both frame parities, and
// the frame skeleton -- every rank queues its tile share (64x64 tiles dealt round-robin,
// physics-engine/_legacy_src/tiling.rs:38-56) and its push to rank 0, rank 0 queues the one exchange and the
// de-interleave into the caller's image; even and odd frames alternate streams, send buffers and receive slots.
//
// Api requirements (all `int` functions return 0 on success, else a GRV_ERR_* code after noting a message that
// Api::error_text() returns on the calling thread):
//   types   Stream, Event (cheap handles, default-constructible, comparable with `== Stream{}` / `== Event{}`)
//   device  set_device(dev), device_synchronize(), malloc(void **, bytes), free(void *)
//   order   stream_wait_event(Stream, Event), event_record(Event, Stream)
//   copies  copy_to_rank0(dst, dst_dev, src, src_dev, bytes, Stream), copy_on_device(dst, src, bytes, Stream)
//   kernels pack_half(src, dst, n_px, Stream), widen_half(src, dst, n_px, Stream), quantize(img, n_px, Stream),
//           unpack_tiles(width, height, G, rank, slot, image, half, Stream)
//   deal    share_pixels(width, height, G, rank), slot_pixels(width, height, G)
//   rccl    group_start(), group_end(), send(src, n_elems, half, rank, Stream), recv(dst, n_elems, half, from, Stream)
#pragma once

#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace grvmulti {

enum { OK = 0, ERR_INVALID = 1, ERR_HIP = 3 };                 // == GRV_OK, GRV_ERR_INVALID, GRV_ERR_HIP
enum { TRANSPORT_RCCL = 1, TRANSPORT_PEER_COPY = 2 };          // == GRV_TRANSPORT_*
enum { FORMAT_RGBA32F = 0, FORMAT_RGBA16F = 1 };               // == GRV_EXCHANGE_*
// grv_multi_test_inject_fault: where the NEXT frame fails (verification hook)
enum { FAULT_NONE = 0, FAULT_RENDER = 1, FAULT_SEND = 2, FAULT_PEER_COPY = 3 };

// One worker thread per rank.  run(job) hands `job(rank)` to every worker and returns when all of
// them have finished QUEUEING (the device work stays asynchronous); the first non-zero status wins.
class RankThreads {
  public:
    explicit RankThreads(int n) : n_(n), rc_(n, 0) {
        for (int r = 0; r < n; ++r) th_.emplace_back([this, r] { loop(r); });
    }
    ~RankThreads() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int run(const std::function<int(int)> &job) {
        std::unique_lock<std::mutex> lk(mu_);
        job_ = &job;
        pending_ = n_;
        ++gen_;
        cv_.notify_all();
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
        for (int rc : rc_)
            if (rc != 0) return rc;
        return 0;
    }

  private:
    void loop(int r) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<int(int)> *job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                job = job_;
            }
            const int rc = (*job)(r);
            {
                std::lock_guard<std::mutex> lk(mu_);
                rc_[r] = rc;
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::vector<int> rc_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<int(int)> *job_ = nullptr;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

template <class Api>
struct Core {
    using Stream = typename Api::Stream;
    using Event = typename Api::Event;

    Api api;
    int G = 0;
    int transport = TRANSPORT_PEER_COPY;
    bool self_exchange = false; // test hook: rank 0's own share travels through the transport too
    int format = FORMAT_RGBA32F; // what travels: RGBA f32 (16 B / pixel) or RGBA binary16 (8 B / pixel)
    std::vector<int> dev;
    std::string err;

    struct Rank {
        Stream s[2] = {Stream{}, Stream{}};
        float *send[2] = {nullptr, nullptr}; // packed tile-order RGBA f32 (ranks >= 1; rank 0 under self_exchange
                                             // and, RGBA16F, as its f32 render target)
        void *send16[2] = {nullptr, nullptr}; // RGBA16F: the share as it travels (ranks that exchange)
        Event arrived[2] = {Event{}, Event{}}; // this rank's tiles of frame parity b sit in rank 0's slot
    };
    std::vector<Rank> rank;
    size_t slot_px = 0; // pixels per receive slot / send buffer (max tiles of a rank * 4096)

    // rank 0 side
    float *recv[2] = {nullptr, nullptr}; // [G][slot_px][4] per parity (RGBA16F: [G][slot_px] x 8 B in the same allocation)
    Stream rs[2] = {Stream{}, Stream{}}; // exchange + unpack streams
    Event unpacked[2] = {Event{}, Event{}};
    bool unpacked_rec[2] = {false, false};
    Event caller_ready = Event{};

    RankThreads *threads = nullptr;
    uint64_t frame = 0;

    // verification hook (grv_multi_test_inject_fault): the next frame fails at this point of this rank
    int fault_kind = FAULT_NONE, fault_rank = 0;

    int fail(int code, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
#define GRVC_API(call)                                                                        \
    do {                                                                                      \
        const int _st = (call);                                                               \
        if (_st != OK) return fail(_st, "%s failed: %s", #call, api.error_text());            \
    } while (0)

    // Waits for the devices and frees every exchange buffer (the next frame allocates for the mode /
    // size then in force).
    int drop_buffers() {
        for (int r = 0; r < G; ++r) {
            GRVC_API(api.set_device(dev[r]));
            GRVC_API(api.device_synchronize());
            for (int b = 0; b < 2; ++b) {
                if (rank[r].send[b]) api.free(rank[r].send[b]);
                if (rank[r].send16[b]) api.free(rank[r].send16[b]);
                rank[r].send[b] = nullptr;
                rank[r].send16[b] = nullptr;
            }
        }
        GRVC_API(api.set_device(dev[0]));
        for (int b = 0; b < 2; ++b) {
            if (recv[b]) api.free(recv[b]);
            recv[b] = nullptr;
            unpacked_rec[b] = false;
        }
        slot_px = 0;
        return OK;
    }

    // Buffers sized for a width x height frame split over G ranks (grown on demand; growing waits for
    // the device -- frames of a fixed size never do).
    int ensure_buffers(uint32_t width, uint32_t height) {
        const size_t need = api.slot_pixels(width, height, G);
        if (need <= slot_px) return OK;
        const int rc = drop_buffers();
        if (rc != OK) return rc;
        const bool half = format == FORMAT_RGBA16F;
        const size_t wire = half ? 8u : 16u; // bytes per pixel as exchanged
        for (int b = 0; b < 2; ++b) GRVC_API(api.malloc(reinterpret_cast<void **>(&recv[b]), (size_t)G * need * wire));
        for (int r = 0; r < G; ++r) {
            const bool direct = (r == 0 && !self_exchange); // rank 0's share needs no transport
            if (direct && !half) continue;                  // ... and, RGBA f32, is rendered into its receive slot
            GRVC_API(api.set_device(dev[r]));
            for (int b = 0; b < 2; ++b) {
                GRVC_API(api.malloc(reinterpret_cast<void **>(&rank[r].send[b]), need * 16u));
                if (half && !direct) GRVC_API(api.malloc(&rank[r].send16[b], need * 8u));
            }
        }
        slot_px = need;
        return OK;
    }

    // The frame skeleton shared by the f64 frame and the f32 compute march.
    //   render(rank, target, stream): queue this rank's tile share into `target` (packed tile order, RGBA f32)
    //   on `stream`; returns a status, its message through render_error(rank).
    int run_frame(uint32_t width, uint32_t height, float *d_rgba, Stream caller,
                  const std::function<int(int, float *, Stream)> &render,
                  const std::function<std::string(int)> &render_error) {
        if (!d_rgba) return fail(ERR_INVALID, "null image");
        if (width == 0 || height == 0) return fail(ERR_INVALID, "empty frame");
        const bool half = format == FORMAT_RGBA16F;
        const int fk = fault_kind, fr = fault_rank; // one shot: consumed by this frame
        fault_kind = FAULT_NONE;
        if (G == 1 && !self_exchange) {
            // one rank: the whole frame is already row-major (GrvFrameBuffers), nothing to exchange
            GRVC_API(api.set_device(dev[0]));
            if (fk == FAULT_RENDER && fr == 0) return fail(ERR_HIP, "rank 0: injected render failure");
            const int rc = render(0, d_rgba, caller);
            if (rc != OK) return fail(rc, "rank 0: %s", render_error(0).c_str());
            // RGBA16F: the image a G-rank handle assembles is the half-rounded frame; so is this one
            if (half) GRVC_API(api.quantize(d_rgba, (size_t)width * height, caller));
            frame++;
            return OK;
        }
        int rc = ensure_buffers(width, height);
        if (rc != OK) return rc;
        const int b = (int)(frame & 1u);
        std::vector<size_t> n_px(G);
        for (int r = 0; r < G; ++r) n_px[r] = api.share_pixels(width, height, G, r);
        const bool rccl = transport == TRANSPORT_RCCL;
        const size_t wire = half ? 8u : 16u;
        std::vector<std::string> rank_err(G);

        // every rank: render its share (and, peer-copy transport, push it to rank 0) on its own thread
        rc = threads->run([&](int r) -> int {
            Rank &R = rank[r];
            auto bad = [&](int code, const char *what) {
                rank_err[r] = std::string(what) + ": " + api.error_text();
                return code;
            };
            int st = api.set_device(dev[r]);
            if (st != OK) return bad(st, "set_device");
            Stream s = R.s[b];
            // the receive slot of this parity is free once frame - 2 has been unpacked
#ifndef GRVMULTI_MUTANT_NO_SLOT_WAIT // (mutation switch of tests/host/multi_tsan.cpp: the detector must notice its absence)
            if (unpacked_rec[b] && (st = api.stream_wait_event(s, unpacked[b])) != OK) return bad(st, "stream_wait_event");
#endif
            // receive slot of rank r: f32 pixels, or (RGBA16F) 8-byte pixels in the same allocation
            char *slot = reinterpret_cast<char *>(recv[b]) + (size_t)r * slot_px * wire;
            const bool direct = (r == 0 && !self_exchange);
            float *target = (direct && !half) ? reinterpret_cast<float *>(slot) : R.send[b];
            if (n_px[r] > 0) {
                if (fk == FAULT_RENDER && fr == r) {
                    rank_err[r] = "injected render failure";
                    return (int)ERR_HIP;
                }
                st = render(r, target, s);
                if (st != OK) {
                    rank_err[r] = render_error(r);
                    return st;
                }
            }
            const void *wire_src = target;
            if (half && n_px[r] > 0) {
                // narrow the share to binary16 where it was rendered: rank 0's straight into its slot
                void *dst = direct ? static_cast<void *>(slot) : R.send16[b];
                if ((st = api.pack_half(target, dst, n_px[r], s)) != OK) return bad(st, "pack_half");
                wire_src = dst;
            }
            if (!direct && !rccl && n_px[r] > 0) {
                if (fk == FAULT_PEER_COPY && fr == r) {
                    rank_err[r] = "injected peer-copy failure";
                    return (int)ERR_HIP;
                }
                st = api.copy_to_rank0(slot, dev[0], wire_src, dev[r], n_px[r] * wire, s);
                if (st != OK) return bad(st, "copy_to_rank0");
            }
            // RCCL: the event marks "rendered"; the transfer itself is ordered by the receive on rank 0's stream
            if ((st = api.event_record(R.arrived[b], s)) != OK) return bad(st, "event_record");
            return (int)OK;
        });
        if (rc != OK) {
            for (int r = 0; r < G; ++r)
                if (!rank_err[r].empty()) return fail(rc, "rank %d: %s", r, rank_err[r].c_str());
            return fail(rc, "a rank failed to queue its share");
        }

        // rank 0: the one exchange, then the de-interleave into the caller's image
        GRVC_API(api.set_device(dev[0]));
        Stream xs = rs[b];
        if (rccl) {
            // one group: G-1 sends on the ranks' render streams (behind their kernels), G-1 receives
            // on rank 0's exchange stream -- concurrent point-to-point transfers, one xGMI link each
            GRVC_API(api.group_start());
            int gst = OK; // a failed call must not leave the group open: always reach group_end
            std::string gmsg;
            for (int r = (self_exchange ? 0 : 1); r < G && gst == OK; ++r) {
                if (n_px[r] == 0) continue;
                // four channels per pixel either way: float or half elements
                const void *src = half ? rank[r].send16[b] : static_cast<const void *>(rank[r].send[b]);
                if (fk == FAULT_SEND && fr == r) {
                    gst = ERR_HIP;
                    gmsg = "injected ncclSend failure";
                    break;
                }
                gst = api.send(src, n_px[r] * 4u, half, r, rank[r].s[b]);
                if (gst == OK) gst = api.recv(reinterpret_cast<char *>(recv[b]) + (size_t)r * slot_px * wire, n_px[r] * 4u, half, r, xs);
                if (gst != OK) gmsg = api.error_text();
            }
            const int gend = api.group_end();
            if (gst != OK) return fail(gst, "ncclSend / ncclRecv failed (rank group closed): %s", gmsg.c_str());
            if (gend != OK) return fail(gend, "group_end failed: %s", api.error_text());
            if (!self_exchange) GRVC_API(api.stream_wait_event(xs, rank[0].arrived[b]));
        } else {
#ifndef GRVMULTI_MUTANT_NO_ARRIVED_WAIT
            for (int r = 0; r < G; ++r) GRVC_API(api.stream_wait_event(xs, rank[r].arrived[b]));
#endif
        }
        // the caller's image may still be read by work the caller queued earlier
        GRVC_API(api.event_record(caller_ready, caller));
        GRVC_API(api.stream_wait_event(xs, caller_ready));
        if (G == 1) {
            // (self-exchange walk) one rank's share is the whole frame, already row-major
            if (half) GRVC_API(api.widen_half(recv[b], d_rgba, n_px[0], xs));
            else GRVC_API(api.copy_on_device(d_rgba, recv[b], n_px[0] * 16u, xs));
        } else {
            for (int r = 0; r < G; ++r) {
                if (n_px[r] == 0) continue;
                const char *slot = reinterpret_cast<const char *>(recv[b]) + (size_t)r * slot_px * wire;
                GRVC_API(api.unpack_tiles(width, height, G, r, slot, d_rgba, half, xs));
            }
        }
        GRVC_API(api.event_record(unpacked[b], xs));
        unpacked_rec[b] = true;
        GRVC_API(api.stream_wait_event(caller, unpacked[b])); // the image is complete in the caller's stream order
        frame++;
        return OK;
    }
#undef GRVC_API
};

} // namespace grvmulti
