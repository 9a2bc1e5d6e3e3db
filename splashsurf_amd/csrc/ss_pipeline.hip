// ss_pipeline.hip -- frame pipeline of libsplashsurf_hip.so (ss_pipeline_*, include/splashsurf_hip.h): a time series of host-resident frames
// through `depth` contexts of ONE device, one host thread per context.
//
// Why it exists.  The reference's time-series use is a loop of reconstruct_surface_inplace over the frames with one workspace (lib.rs:340-346,
// 466-470; the CLI's frame loop, reconstruct.rs).  Here a host-to-host frame is a chain -- pageable upload, ~45 kernels, the mesh back over PCIe, the
// u32 -> u64 widening of the triangle indices on host threads -- in which nothing of ONE frame overlaps (14.7 ms per S10M-tank
// frame of which 6.6 ms are kernels).  The overlap that exists is BETWEEN frames: while frame k's mesh crosses PCIe and is widened, frame k + 1
// uploads and computes on another context's stream.  bench.py measured that with two Python threads (`pcie_pipelined`); this file makes it an entry
// point of the C ABI, so that a Rust / C host gets it without writing the threads.
//
// Built on the library's public C ABI only (a context, a result and the host accessors per slot): no kernel, no device state of its own.
// Frames are handed back in submission order; frame t runs on slot t % depth.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/splashsurf_hip.h"

namespace {

struct Slot {
    ss_context* ctx = nullptr;
    ss_result* res = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    enum State { IDLE, QUEUED, RUNNING, DONE } state = IDLE;  // IDLE: never used, or its frame was handed back
    bool quit = false;
    int device = 0;
    // the frame
    bool f64 = false;
    const void* xyz = nullptr;
    uint64_t n = 0;
    ss_params_f32 p32;
    ss_params_f64 p64;
    uint32_t fetch = 0;
    uint64_t ticket = 0;
    // its outcome
    ss_status status = SS_OK;
    std::string err;
    double ms_reconstruct = 0.0, ms_fetch = 0.0;

    void run_frame() {
        const auto t0 = std::chrono::steady_clock::now();
        ss_status s = f64 ? ss_reconstruct_surface_inplace_f64(ctx, static_cast<const double*>(xyz), n, &p64, res)
                          : ss_reconstruct_surface_inplace_f32(ctx, static_cast<const float*>(xyz), n, &p32, res);
        const auto t1 = std::chrono::steady_clock::now();
        // the host mirrors the caller asked for: filled HERE, on this slot's thread and stream, while the next frame runs on another slot
        uint64_t cnt = 0;
        if (s == SS_OK && (fetch & SS_FETCH_VERTICES)) {
            if (f64) {
                const double* v = nullptr;
                s = ss_result_vertices_f64(res, &v, &cnt);
            } else {
                const float* v = nullptr;
                s = ss_result_vertices(res, &v, &cnt);
            }
        }
        if (s == SS_OK && (fetch & SS_FETCH_TRIANGLES_U64)) {
            const uint64_t* t = nullptr;
            s = ss_result_triangles(res, &t, &cnt);
        }
        if (s == SS_OK && (fetch & SS_FETCH_TRIANGLES_U32)) {
            const uint32_t* t = nullptr;
            s = ss_result_triangles_u32(res, &t, &cnt);
        }
        if (s == SS_OK && (fetch & SS_FETCH_DENSITIES)) {
            if (f64) {
                const double* r = nullptr;
                s = ss_result_particle_densities_f64(res, &r, &cnt);
            } else {
                const float* r = nullptr;
                s = ss_result_particle_densities(res, &r, &cnt);
            }
        }
        const auto t2 = std::chrono::steady_clock::now();
        status = s;
        err = s == SS_OK ? std::string() : std::string(ss_last_error(ctx));
        ms_reconstruct = std::chrono::duration<double, std::milli>(t1 - t0).count();
        ms_fetch = std::chrono::duration<double, std::milli>(t2 - t1).count();
    }

    void loop() {
        (void)hipSetDevice(device);  // (the current device is per host thread)
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return quit || state == QUEUED; });
            if (state != QUEUED) return;  // quit, nothing queued
            state = RUNNING;
            lk.unlock();
            try {
                run_frame();
            } catch (...) {  // (out of host memory inside the call: the frame fails, the thread lives)
                status = SS_ERR_UNKNOWN;
                err = "exception in the frame's host code (out of host memory?)";
            }
            lk.lock();
            state = DONE;
            cv.notify_all();
        }
    }
};

}  // namespace

struct ss_pipeline {
    int device = 0;
    int depth = 0;
    std::vector<std::unique_ptr<Slot>> slots;
    uint64_t next_submit = 0, next_return = 0;  // tickets: frames next_return .. next_submit - 1 are in flight
    std::string err;
};

namespace {

ss_status pfail(ss_pipeline* p, ss_status s, const std::string& msg) {
    if (p) p->err = msg;
    return s;
}

template <class R, class PRM>
ss_status submit(ss_pipeline* p, const R* xyz, uint64_t n, const PRM* prm, uint32_t fetch, uint64_t* ticket) {
    if (!p) return SS_ERR_INVALID_ARGUMENT;
    p->err.clear();
    if (!prm) return pfail(p, SS_ERR_INVALID_ARGUMENT, "null parameters");
    if (n > 0 && !xyz) return pfail(p, SS_ERR_INVALID_ARGUMENT, "null particle array");
    if (fetch & ~(uint32_t)(SS_FETCH_VERTICES | SS_FETCH_TRIANGLES_U64 | SS_FETCH_TRIANGLES_U32 | SS_FETCH_DENSITIES))
        return pfail(p, SS_ERR_INVALID_ARGUMENT, "unknown SS_FETCH_* bit");
    if (p->next_submit - p->next_return >= (uint64_t)p->depth)
        return pfail(p, SS_ERR_INVALID_ARGUMENT, "the pipeline is full: take a frame with ss_pipeline_next before submitting another");
    Slot& s = *p->slots[(size_t)(p->next_submit % (uint64_t)p->depth)];
    {
        std::lock_guard<std::mutex> lk(s.mu);
        // (IDLE by construction: the frame that used this slot before has the ticket next_submit - depth < next_return)
        s.f64 = sizeof(R) == 8;
        s.xyz = xyz;
        s.n = n;
        if constexpr (sizeof(R) == 8)
            s.p64 = *prm;
        else
            s.p32 = *prm;
        s.fetch = fetch;
        s.ticket = p->next_submit;
        s.state = Slot::QUEUED;
    }
    s.cv.notify_all();
    if (ticket) *ticket = p->next_submit;
    ++p->next_submit;
    return SS_OK;
}

}  // namespace

extern "C" {

ss_status ss_pipeline_create(int device_id, int depth, ss_pipeline** out) {
    if (!out) return SS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (depth < 1 || depth > SS_PIPELINE_MAX_DEPTH) return SS_ERR_INVALID_ARGUMENT;
    std::unique_ptr<ss_pipeline> p(new (std::nothrow) ss_pipeline());
    if (!p) return SS_ERR_UNKNOWN;
    p->device = device_id;
    p->depth = depth;
    ss_status s = SS_OK;
    for (int i = 0; i < depth && s == SS_OK; ++i) {
        std::unique_ptr<Slot> sl(new (std::nothrow) Slot());
        if (!sl) {
            s = SS_ERR_UNKNOWN;
            break;
        }
        sl->device = device_id;
        s = ss_context_create(device_id, &sl->ctx);
        if (s == SS_OK) s = ss_result_create(sl->ctx, &sl->res);
        p->slots.push_back(std::move(sl));
    }
    if (s == SS_OK) {
        try {
            for (auto& sl : p->slots) {
                Slot* raw = sl.get();
                raw->th = std::thread([raw] { raw->loop(); });
            }
        } catch (...) {  // (no thread to be had: nothing may leave a C entry point but a status)
            s = SS_ERR_UNKNOWN;
        }
    }
    if (s != SS_OK) {
        for (auto& sl : p->slots) {
            if (sl->th.joinable()) {
                {
                    std::lock_guard<std::mutex> lk(sl->mu);
                    sl->quit = true;
                }
                sl->cv.notify_all();
                sl->th.join();
            }
            if (sl->res) ss_result_free(sl->res);
            if (sl->ctx) ss_context_destroy(sl->ctx);
        }
        return s;
    }
    *out = p.release();
    return SS_OK;
}

void ss_pipeline_destroy(ss_pipeline* p) {
    if (!p) return;
    for (auto& sl : p->slots) {
        {
            std::unique_lock<std::mutex> lk(sl->mu);
            sl->cv.wait(lk, [&] { return sl->state == Slot::IDLE || sl->state == Slot::DONE; });  // a frame in flight finishes first
            sl->quit = true;
        }
        sl->cv.notify_all();
        if (sl->th.joinable()) sl->th.join();
    }
    for (auto& sl : p->slots) {
        if (sl->res) ss_result_free(sl->res);
        if (sl->ctx) ss_context_destroy(sl->ctx);
    }
    delete p;
}

const char* ss_pipeline_last_error(const ss_pipeline* p) { return p ? p->err.c_str() : "null pipeline"; }

int ss_pipeline_depth(const ss_pipeline* p) { return p ? p->depth : 0; }

int ss_pipeline_in_flight(const ss_pipeline* p) { return p ? (int)(p->next_submit - p->next_return) : 0; }

ss_context* ss_pipeline_context(ss_pipeline* p, int slot) { return (p && slot >= 0 && slot < p->depth) ? p->slots[(size_t)slot]->ctx : nullptr; }

ss_status ss_pipeline_set_option(ss_pipeline* p, int option, int value) {
    if (!p) return SS_ERR_INVALID_ARGUMENT;
    p->err.clear();
    if (p->next_submit != p->next_return) return pfail(p, SS_ERR_INVALID_ARGUMENT, "options can be set only while no frame is in flight");
    for (auto& sl : p->slots) {
        const ss_status s = ss_context_set_option(sl->ctx, option, value);
        if (s != SS_OK) return pfail(p, s, ss_last_error(sl->ctx));
    }
    return SS_OK;
}

ss_status ss_pipeline_submit_f32(ss_pipeline* p, const float* xyz, uint64_t n, const ss_params_f32* prm, uint32_t fetch, uint64_t* ticket) {
    return submit<float>(p, xyz, n, prm, fetch, ticket);
}
ss_status ss_pipeline_submit_f64(ss_pipeline* p, const double* xyz, uint64_t n, const ss_params_f64* prm, uint32_t fetch, uint64_t* ticket) {
    return submit<double>(p, xyz, n, prm, fetch, ticket);
}

ss_status ss_pipeline_next(ss_pipeline* p, ss_result** result, uint64_t* ticket) {
    if (!p) return SS_ERR_INVALID_ARGUMENT;
    p->err.clear();
    if (result) *result = nullptr;
    if (p->next_return == p->next_submit) return pfail(p, SS_ERR_INVALID_ARGUMENT, "no frame in flight");
    Slot& s = *p->slots[(size_t)(p->next_return % (uint64_t)p->depth)];
    ss_status st;
    {
        std::unique_lock<std::mutex> lk(s.mu);
        s.cv.wait(lk, [&] { return s.state == Slot::DONE; });
        s.state = Slot::IDLE;
        st = s.status;
        if (st != SS_OK) p->err = s.err;
    }
    if (ticket) *ticket = s.ticket;
    ++p->next_return;
    if (st == SS_OK && result) *result = s.res;
    return st;
}

int ss_pipeline_ready(ss_pipeline* p) {
    if (!p || p->next_return == p->next_submit) return 0;
    Slot& s = *p->slots[(size_t)(p->next_return % (uint64_t)p->depth)];
    std::lock_guard<std::mutex> lk(s.mu);
    return s.state == Slot::DONE ? 1 : 0;
}

ss_status ss_pipeline_frame_times(ss_pipeline* p, int slot, double* ms_reconstruct, double* ms_fetch) {
    if (!p || slot < 0 || slot >= p->depth) return SS_ERR_INVALID_ARGUMENT;
    Slot& s = *p->slots[(size_t)slot];
    std::lock_guard<std::mutex> lk(s.mu);
    if (s.state == Slot::QUEUED || s.state == Slot::RUNNING) return pfail(p, SS_ERR_INVALID_ARGUMENT, "the slot's frame is still in flight");
    if (ms_reconstruct) *ms_reconstruct = s.ms_reconstruct;
    if (ms_fetch) *ms_fetch = s.ms_fetch;
    return SS_OK;
}

}  // extern "C"
