// ss_dist.hip -- the multi-GPU reconstruction behind the C ABI: one process (or host thread) per GPU,
// every exchange inside the library -- RCCL point-to-point / collectives over xGMI, no host-language code on the data path.
//
// The reference is single-process; its unit of parallelism is the subdomain (dense_subdomains.rs:349-494 builds the
// decomposition, :1582-1598 iterates it with rayon).  Here the subdomain grid of ONE global domain is cut into `world`
// axis-aligned bricks of subdomains by recursive bisection of the owner histogram (balanced by particle count); rank r
// reconstructs brick r through the two-phase shard entry points of ss_api.hip (ss_shard_begin_* / ss_shard_finish):
//   1. global particle ids = concatenation of the ranks' inputs (defines the level set's summation order); global AABB;
//   2. owner histogram all-reduced, bricks derived identically on every rank (rcb_split);
//   3. sparse all-to-all #1 (grouped ncclSend/ncclRecv): (id, position) of every particle inside a brick grown by the ghost margin;
//   4. phase 1: binning + densities of the particles contained in the brick;
//   5. sparse all-to-all #2: (id, rho) from the owner to the ranks holding the particle as a ghost (copied, never recomputed);
//   6. phase 2: level set + marching cubes of the brick;
//   7. ss_dist_assemble: a vertex on a brick face belongs to the LOWEST rank whose brick holds its edge (the rule of
//      globalize_local_edge, dense_subdomains.rs:1260-1329); counts are all-gathered into global vertex / triangle offsets,
//      owners send (edge key, global id) of shared vertices to the other holders (sparse all-to-all #3; the receiving side is
//      the hash join of `stitching`, dense_subdomains.rs:1693-1733, as a radix sort + binary search), triangles are rewritten
//      to global ids.  The mesh is the concatenation over ranks of (owned vertices, triangles).
// splashsurf_amd/distributed.py is the host-side mirror of the same algorithm over torch.distributed (gloo in the CPU tests).
//
// Transports: RCCL (loaded with dlopen at first use, so single-GPU users need no RCCL), and an in-process group of host
// threads sharing one device (tests: the whole algorithm runs on a 1-GPU box; RCCL refuses two ranks on one device).
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/splashsurf_hip.h"
#include "ss_host.h"
#include "ss_kernels.h"
#include "ss_prims.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// transports
// ---------------------------------------------------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // SPLASH_RCCL_LIB: the RCCL build to bind, by path (a site's own build; the stand-in of the no-GPU tests, tests/emu/fake_rccl.cpp).  Otherwise
        // prefer a copy that is already in the process (PyTorch-ROCm bundles its own librccl.so): one RCCL per process
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (const char* named = getenv("SPLASH_RCCL_LIB")) {
            if (!(api.handle = dlopen(named, RTLD_NOW | RTLD_LOCAL))) {
                const char* why = dlerror();
                api.error = std::string("SPLASH_RCCL_LIB: ") + (why ? why : "dlopen failed");
                return;
            }
        }
        if (!api.handle)
            for (const char* n : names)
                if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
        if (!api.handle)
            for (const char* n : names)
                if ((api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!api.handle) {
            const char* why = dlerror();  // (a second call would return NULL: dlerror clears the state it reports)
            api.error = std::string("RCCL not found: ") + (why ? why : "dlopen failed");
            return;
        }
        bool ok = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(api.handle, name);
            if (!p) {
                ok = false;
                api.error = std::string("RCCL symbol missing: ") + name;
            }
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
        api.CommGetAsyncError = reinterpret_cast<decltype(api.CommGetAsyncError)>(sym("ncclCommGetAsyncError"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) api.handle = nullptr;
    });
    return &api;
}

// host threads sharing one device: barrier + tables of what every rank published
struct LocalGroup {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool failed = false;
    std::vector<std::vector<uint8_t>> host;          // allgather_host slots
    std::vector<const uint8_t*> send_ptr;            // exchange: device send buffers
    std::vector<std::vector<uint64_t>> send_off;     // exchange: byte offsets per destination (world + 1)
    std::vector<const uint32_t*> red_ptr;            // allreduce: device buffers
    // ss_comm_local_group_take_turns: the ranks share ONE device, so their kernels delay each other and a rank's stage timers show
    // the crowd, not the rank.  With take_turns a rank holds `turn` while it computes between two exchange steps (never while it
    // waits for a peer) and drains its stream before it hands the device on: its timers then read what the rank takes on a GPU of
    // its own, which is what bench.py --pseudo-ranks reports.
    std::mutex turn;
    bool take_turns = false;
    bool barrier(double timeout_s) {
        std::unique_lock<std::mutex> lk(m);
        if (failed) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || failed; });
        if (!ok) {
            failed = true;  // a rank never arrived: release everybody with an error instead of hanging
            cv.notify_all();
        }
        return ok && !failed;
    }
};

__global__ __launch_bounds__(256) void k_sum_peers_u32(const uint32_t* const* __restrict__ peers, int world, size_t n, uint32_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = 0;
    for (int q = 0; q < world; ++q) s += peers[q][i];
    out[i] = s;
}

}  // namespace

struct ss_comm {
    ss_context* ctx = nullptr;
    int rank = 0, world = 1;
    int kind = 0;  // 0 in-process group, 1 RCCL
    ncclComm_t nccl = nullptr;
    bool own_nccl = false;
    std::shared_ptr<LocalGroup> group;
    double timeout_s = 120.0;
    // scratch
    DevBuf small_dev, small_dev2, red_tmp, peers_dev;
    HostBuf small_host;
    // counts the host waits for (SSMailSlot, ss_prims.h): pinned host memory mapped into the device, polled instead of a device-to-host copy
    // and a stream synchronisation.  16 slots of {value, seq}, then COMM_MAIL_WORDS 64-bit words for small arrays (AABB, exchange bounds).
    unsigned long long* mail_host = nullptr;
    unsigned long long* mail_dev = nullptr;
    unsigned long long mail_seq = 0;
    DevBuf zeros;  // zeroed words of one step of the flow: scan states, counters (one memset per use)
    // partition feedback (ss_comm_set_balance_feedback): the cost per owned particle every rank measured in the previous call weighs the
    // owner histogram of the next one (a time series of frames is the real workload; the first call balances particle counts)
    bool feedback = false;
    std::vector<double> cost_per_particle;  // per rank, from the previous call (empty: none yet)
    std::vector<int64_t> prev_bricks;
    int prev_ns[3] = {0, 0, 0};
    double prev_origin[3] = {0.0, 0.0, 0.0};  // grid origin of the call the previous bricks belong to
    // state of the last ss_dist_reconstruct / ss_dist_assemble
    DevBuf xyz_in, hist, flags, offs, boxes_dev, mask, sendbuf, recvbuf, gids, L, owned, sort_tmp, keys_a, keys_b, vals_a, vals_b, owner, holder, gid_local, mine_off, tri64, vown, kown, err;
    bool is_f64 = false;
    std::vector<int64_t> bricks;  // world x 6 (lo[3], hi[3])
    int ns[3] = {0, 0, 0};
    int n_cubes = 0;
    std::vector<uint64_t> per_rank_owned, per_rank_held;
    ss_dist_info info;
    bool assembled = false;
};

namespace {

ss_status comm_fail(ss_comm* c, const std::string& msg) { return fail(c->ctx, SS_ERR_DEVICE, msg); }

// see LocalGroup::turn.  Never hold one across a barrier or a collective.
struct TurnGuard {
    ss_comm* c;
    const char* what;
    bool held;
    double t_start = 0.0;
    explicit TurnGuard(ss_comm* comm, const char* name) : c(comm), what(name), held(comm->kind == 0 && comm->world > 1 && comm->group && comm->group->take_turns) {
        if (held) {
            c->group->turn.lock();
            t_start = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
        }
    }
    void release() {
        if (!held) return;
        (void)hipStreamSynchronize(c->ctx->stream);  // the device is handed on idle
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_start;
        c->info.ms_own_turns += ms;
        static const bool trace = getenv("SPLASH_DIST_TRACE") != nullptr;  // per-section times of every rank on stderr
        if (trace) fprintf(stderr, "[dist-trace] rank %d %s %.3f ms\n", c->rank, what, ms);
        c->group->turn.unlock();
        held = false;
    }
    ~TurnGuard() { release(); }
    TurnGuard(const TurnGuard&) = delete;
    TurnGuard& operator=(const TurnGuard&) = delete;
};

// wait for the context's stream, but never forever: a peer that died leaves RCCL kernels spinning
ss_status wait_stream(ss_comm* c, const char* what) {
    ss_context* ctx = c->ctx;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        hipError_t e = hipStreamQuery(ctx->stream);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) {
            (void)hipGetLastError();
            return comm_fail(c, std::string("HIP error while waiting for ") + what + ": " + hipGetErrorString(e));
        }
        if (c->kind == 1 && c->nccl) {
            ncclResult_t ar = ncclSuccess;
            if (rccl_api()->CommGetAsyncError(c->nccl, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
                return comm_fail(c, std::string("RCCL asynchronous error during ") + what + ": " + rccl_api()->GetErrorString(ar));
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt > c->timeout_s) {
            // (a communicator the host handed in with ss_comm_adopt_rccl stays the host's: never aborted or destroyed here)
            if (c->kind == 1 && c->nccl && c->own_nccl) (void)rccl_api()->CommAbort(c->nccl), c->nccl = nullptr;
            return comm_fail(c, std::string("timeout (") + std::to_string((int)c->timeout_s) + " s, SPLASH_COMM_TIMEOUT_S) waiting for " + what +
                                    ": a peer rank did not take part in the exchange");
        }
        if (dt > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return SS_OK;
}

// ---- counts the host waits for ----
#define COMM_MAIL_SLOTS 16
#define COMM_MAIL_WORDS 96
ss_status comm_ensure_mail(ss_comm* c) {
    if (c->mail_host) return SS_OK;
    ss_context* ctx = c->ctx;
    void* h = nullptr;
    const size_t bytes = (COMM_MAIL_SLOTS * 2 + COMM_MAIL_WORDS) * sizeof(unsigned long long);
    SS_HIP(ctx, hipHostMalloc(&h, bytes, hipHostMallocMapped));
    memset(h, 0, bytes);
    void* d = nullptr;
    SS_HIP(ctx, hipHostGetDevicePointer(&d, h, 0));
    c->mail_host = reinterpret_cast<unsigned long long*>(h);
    c->mail_dev = reinterpret_cast<unsigned long long*>(d);
    return SS_OK;
}
SSMailSlot comm_slot(ss_comm* c, int k) { return SSMailSlot{c->mail_dev + 2 * k, ++c->mail_seq}; }
unsigned long long* comm_words_dev(ss_comm* c) { return c->mail_dev + 2 * COMM_MAIL_SLOTS; }
const volatile unsigned long long* comm_words_host(ss_comm* c) { return c->mail_host + 2 * COMM_MAIL_SLOTS; }
// polls the pinned word (the stream goes on with whatever is enqueued behind the posting kernel); a drained stream without the value, a stream
// or RCCL error or the communicator's timeout end the wait with an error
ss_status comm_mail_wait(ss_comm* c, const SSMailSlot& m, unsigned long long* value, const char* what) {
    ss_context* ctx = c->ctx;
    const int k = (int)((m.p - c->mail_dev) / 2);
    volatile unsigned long long* h = c->mail_host + 2 * k;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long it = 0;; ++it) {
        if (__atomic_load_n(&h[1], __ATOMIC_ACQUIRE) == m.seq) {
            if (value) *value = h[0];
            return SS_OK;
        }
        if ((it & 0xFFFu) == 0xFFFu) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(&h[1], __ATOMIC_ACQUIRE) == m.seq) {
                    if (value) *value = h[0];
                    return SS_OK;
                }
                return comm_fail(c, std::string("a count the host waits for never arrived (stream drained): ") + what);
            }
            if (q != hipErrorNotReady) {
                (void)hipGetLastError();
                return comm_fail(c, std::string("HIP error while waiting for ") + what + ": " + hipGetErrorString(q));
            }
            if (c->kind == 1 && c->nccl) {
                ncclResult_t ar = ncclSuccess;
                if (rccl_api()->CommGetAsyncError(c->nccl, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
                    return comm_fail(c, std::string("RCCL asynchronous error while waiting for ") + what + ": " + rccl_api()->GetErrorString(ar));
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s)
                return comm_fail(c, std::string("timed out waiting for ") + what);
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}
// n zeroed 32-bit words (one memset)
ss_status comm_zeros(ss_comm* c, size_t words, uint32_t** out) {
    ss_context* ctx = c->ctx;
    words = (words + 16 + 3) & ~(size_t)3;  // (whole 16-byte units: ss_round16, ss_host.h)
    SS_HIP(ctx, c->zeros.reserve(words * 4));
    SS_HIP(ctx, hipMemsetAsync(c->zeros.p, 0, words * 4, ctx->stream));
    *out = c->zeros.as<uint32_t>();
    return SS_OK;
}

#define SS_NCCL(c, call)                                                                                             \
    do {                                                                                                             \
        ncclResult_t _r = (call);                                                                                    \
        if (_r != ncclSuccess && _r != ncclInProgress)                                                               \
            return comm_fail(c, std::string("RCCL error: ") + rccl_api()->GetErrorString(_r) + " at " #call);        \
    } while (0)

// every rank contributes `bytes` of host data; out receives world * bytes (rank-major)
ss_status comm_allgather_host(ss_comm* c, const void* in, size_t bytes, void* out) {
    if (c->world == 1 && c->kind == 0) {  // (a one-rank RCCL communicator still goes through RCCL: the plumbing is exercised on a 1-GPU box)
        memcpy(out, in, bytes);
        return SS_OK;
    }
    ss_context* ctx = c->ctx;
    if (c->kind == 0) {
        LocalGroup& g = *c->group;
        {
            std::lock_guard<std::mutex> lk(g.m);
            g.host[c->rank].assign((const uint8_t*)in, (const uint8_t*)in + bytes);
        }
        if (!g.barrier(c->timeout_s)) return comm_fail(c, "in-process group: a rank did not reach the all-gather");
        for (int q = 0; q < c->world; ++q) {
            if (g.host[q].size() != bytes) return comm_fail(c, "in-process group: all-gather size mismatch");
            memcpy((uint8_t*)out + (size_t)q * bytes, g.host[q].data(), bytes);
        }
        if (!g.barrier(c->timeout_s)) return comm_fail(c, "in-process group: a rank did not leave the all-gather");
        return SS_OK;
    }
    SS_HIP(ctx, c->small_dev.reserve(bytes + 64));
    SS_HIP(ctx, c->small_dev2.reserve(bytes * (size_t)c->world + 64));
    SS_HIP(ctx, hipMemcpyAsync(c->small_dev.p, in, bytes, hipMemcpyHostToDevice, ctx->stream));
    SS_NCCL(c, rccl_api()->AllGather(c->small_dev.p, c->small_dev2.p, bytes, ncclInt8, c->nccl, ctx->stream));
    ss_status s = wait_stream(c, "ncclAllGather");
    if (s != SS_OK) return s;
    SS_HIP(ctx, hipMemcpy(out, c->small_dev2.p, bytes * (size_t)c->world, hipMemcpyDeviceToHost));
    return SS_OK;
}

// in-place sum of a device array of n uint32 over all ranks
ss_status comm_allreduce_sum_u32(ss_comm* c, uint32_t* dev, size_t n) {
    if ((c->world == 1 && c->kind == 0) || n == 0) return SS_OK;
    ss_context* ctx = c->ctx;
    if (c->kind == 1) {
        SS_NCCL(c, rccl_api()->AllReduce(dev, dev, n, ncclUint32, ncclSum, c->nccl, ctx->stream));
        return wait_stream(c, "ncclAllReduce");
    }
    LocalGroup& g = *c->group;
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    {
        std::lock_guard<std::mutex> lk(g.m);
        g.red_ptr[c->rank] = dev;
    }
    if (!g.barrier(c->timeout_s)) return comm_fail(c, "in-process group: a rank did not reach the all-reduce");
    SS_HIP(ctx, c->red_tmp.reserve(n * 4 + 64));
    SS_HIP(ctx, c->peers_dev.reserve((size_t)c->world * sizeof(void*) + 64));
    SS_HIP(ctx, hipMemcpyAsync(c->peers_dev.p, g.red_ptr.data(), (size_t)c->world * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_sum_peers_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, c->peers_dev.as<const uint32_t*>(), c->world, n,
                       c->red_tmp.as<uint32_t>());
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!g.barrier(c->timeout_s)) return comm_fail(c, "in-process group: a rank did not finish the all-reduce");  // everybody has read everybody
    SS_HIP(ctx, hipMemcpyAsync(dev, c->red_tmp.p, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

// Sparse all-to-all of device byte ranges: send[send_off[q] .. send_off[q+1]) goes to rank q and lands at
// recv[recv_off[r] ..) of the receiver, r = the sender.  RCCL: ONE group of ncclSend / ncclRecv (pairs with nothing to say are
// skipped on both sides, which both sides know from the all-gathered count matrix).
ss_status comm_exchange(ss_comm* c, const uint8_t* send, const uint64_t* send_off, uint8_t* recv, const uint64_t* recv_off) {
    ss_context* ctx = c->ctx;
    const int me = c->rank;
    const uint64_t self_bytes = send_off[me + 1] - send_off[me];
    if (self_bytes != recv_off[me + 1] - recv_off[me]) return comm_fail(c, "exchange: inconsistent self segment");
    if (self_bytes) SS_HIP(ctx, hipMemcpyAsync(recv + recv_off[me], send + send_off[me], self_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (c->world == 1) return SS_OK;
    if (c->kind == 1) {
        SS_NCCL(c, rccl_api()->GroupStart());
        for (int q = 0; q < c->world; ++q) {
            if (q == me) continue;
            const uint64_t sb = send_off[q + 1] - send_off[q], rb = recv_off[q + 1] - recv_off[q];
            if (sb) SS_NCCL(c, rccl_api()->Send(send + send_off[q], sb, ncclInt8, q, c->nccl, ctx->stream));
            if (rb) SS_NCCL(c, rccl_api()->Recv(recv + recv_off[q], rb, ncclInt8, q, c->nccl, ctx->stream));
        }
        SS_NCCL(c, rccl_api()->GroupEnd());
        return wait_stream(c, "grouped ncclSend/ncclRecv");
    }
    LocalGroup& g = *c->group;
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the send buffer is complete
    {
        std::lock_guard<std::mutex> lk(g.m);
        g.send_ptr[me] = send;
        g.send_off[me].assign(send_off, send_off + c->world + 1);
    }
    if (!g.barrier(c->timeout_s)) return comm_fail(c, "in-process group: a rank did not reach the exchange");
    for (int q = 0; q < c->world; ++q) {
        if (q == me) continue;
        const uint64_t rb = recv_off[q + 1] - recv_off[q];
        const uint64_t sb = g.send_off[q][me + 1] - g.send_off[q][me];
        if (rb != sb) return comm_fail(c, "in-process group: exchange size mismatch");
        if (rb) SS_HIP(ctx, hipMemcpyAsync(recv + recv_off[q], g.send_ptr[q] + g.send_off[q][me], rb, hipMemcpyDeviceToDevice, ctx->stream));
    }
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!g.barrier(c->timeout_s)) return comm_fail(c, "in-process group: a rank did not finish the exchange");
    return SS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// partition: recursive coordinate bisection of the subdomain grid (same rule as distributed.py: bricks_from_histogram)
// ---------------------------------------------------------------------------------------------------------------------
void rcb_split(const std::vector<double>& hist, const int ns[3], const int lo[3], const int hi[3], int r0, int r1, const double pref[3], double tol,
               std::vector<int64_t>& out) {
    const int k = r1 - r0;
    auto put = [&](int r, const int a[3], const int b[3]) {
        for (int d = 0; d < 3; ++d) {
            out[(size_t)r * 6 + d] = a[d];
            out[(size_t)r * 6 + 3 + d] = b[d];
        }
    };
    if (k == 1) {
        put(r0, lo, hi);
        return;
    }
    const int k1 = k / 2;
    const double frac = (double)k1 / (double)k;
    const int ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    std::vector<double> marg[3];
    double total = 0.0;
    for (int d = 0; d < 3; ++d) marg[d].assign((size_t)std::max(ext[d], 0), 0.0);
    for (int x = lo[0]; x < hi[0]; ++x)
        for (int y = lo[1]; y < hi[1]; ++y)
            for (int z = lo[2]; z < hi[2]; ++z) {
                const double v = hist[((size_t)x * ns[1] + y) * ns[2] + z];
                marg[0][x - lo[0]] += v;
                marg[1][y - lo[1]] += v;
                marg[2][z - lo[2]] += v;
                total += v;
            }
    struct Cand { double err; long long area; int axis, cut; };
    std::vector<Cand> cands;
    for (int a = 0; a < 3; ++a) {
        if (ext[a] < 2) continue;
        int cut;
        double err = 0.0;
        if (total > 0.0) {
            double cum = 0.0, best = 1e300;
            cut = 1;
            for (int c = 1; c < ext[a]; ++c) {
                cum += marg[a][c - 1];
                const double e = std::fabs(cum - total * frac);
                if (e < best) {
                    best = e;
                    cut = c;
                }
            }
            err = best / total;
        } else {
            cut = std::min(std::max((int)std::floor(ext[a] * frac + 0.5), 1), ext[a] - 1);
        }
        long long area = 1;
        for (int d = 0; d < 3; ++d)
            if (d != a) area *= ext[d];
        cands.push_back({err, area, a, cut});
    }
    if (cands.empty()) {  // a single subdomain for several ranks: the first gets it, the others get empty bricks
        put(r0, lo, hi);
        int elo[3] = {hi[0], lo[1], lo[2]};
        for (int r = r0 + 1; r < r1; ++r) put(r, elo, hi);
        return;
    }
    double best = 1e300;
    for (const Cand& c : cands) best = std::min(best, c.err);
    const Cand* pick = nullptr;
    for (const Cand& c : cands) {
        if (c.err > best + tol) continue;
        if (!pick || c.area < pick->area || (c.area == pick->area && (pref[c.axis] < pref[pick->axis] || (pref[c.axis] == pref[pick->axis] && (c.err < pick->err || (c.err == pick->err && c.axis < pick->axis))))))
            pick = &c;
    }
    int mid_hi[3] = {hi[0], hi[1], hi[2]}, mid_lo[3] = {lo[0], lo[1], lo[2]};
    mid_hi[pick->axis] = lo[pick->axis] + pick->cut;
    mid_lo[pick->axis] = lo[pick->axis] + pick->cut;
    rcb_split(hist, ns, lo, mid_hi, r0, r0 + k1, pref, tol, out);
    rcb_split(hist, ns, mid_lo, hi, r0 + k1, r1, pref, tol, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------
struct DistBox {  // coordinate box of a brick grown by the ghost margin (conservative; the engine applies the exact rule)
    double lo[3], hi[3];
    int empty;
};

template <class R>
__global__ __launch_bounds__(256) void k_owner_hist(const R* __restrict__ xyz, uint64_t n, R g0, R g1, R g2, R sub_size, int ns0, int ns1, int ns2,
                                                    uint32_t* __restrict__ hist) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const R g[3] = {g0, g1, g2};
    const int ns[3] = {ns0, ns1, ns2};
    int s[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int v = (int)ss_floor((xyz[3 * i + d] - g[d]) / sub_size);
        s[d] = max(0, min(ns[d] - 1, v));
    }
    // consecutive particles mostly share their subdomain: one atomic per distinct bin of a wave instead of one per particle
    const uint32_t bin = (uint32_t)(((size_t)s[0] * ns1 + s[1]) * ns2 + s[2]);
    unsigned long long todo = __ballot(true);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lbin = (uint32_t)__shfl((int)bin, leader);
        const unsigned long long same = __ballot(bin == lbin) & todo;
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lbin], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

// mask[i] = the destinations (bit q <=> rank q, among `active`) whose box holds particle i (and i is owned, if `owned` is given):
// one pass over the particles for all destinations
struct DistBoxes {  // world <= 64
    DistBox box[64];
};
template <class R>
__global__ __launch_bounds__(256) void k_box_masks(const R* __restrict__ xyz, uint64_t n, const DistBoxes* __restrict__ boxes, unsigned long long active,
                                                   const uint32_t* __restrict__ owned, unsigned long long* __restrict__ mask) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long m = 0;
    if (!owned || owned[i]) {
        const double x = (double)xyz[3 * i], y = (double)xyz[3 * i + 1], z = (double)xyz[3 * i + 2];
        for (unsigned long long todo = active; todo; todo &= todo - 1) {
            const int q = __ffsll((long long)todo) - 1;
            const DistBox& b = boxes->box[q];
            if (x >= b.lo[0] && x <= b.hi[0] && y >= b.lo[1] && y <= b.hi[1] && z >= b.lo[2] && z <= b.hi[2]) m |= 1ull << q;
        }
    }
    mask[i] = m;
}

// "vertex v is this rank's to number" as a functor over the vertex index (entry n of a scan reads 0, so that the scan ends with the count)
struct VertexFlag {  // owner[v] == me && rank q also holds the vertex (q < 0: any)
    const uint32_t* owner;
    const unsigned long long* holder;
    uint64_t n;
    uint32_t me;
    int q;
    __host__ __device__ uint32_t operator()(uint64_t v) const {
        if (v >= n) return 0u;
        return (owner[v] == me && (q < 0 || ((holder[v] >> q) & 1ull))) ? 1u : 0u;
    }
};

// "the ranks element i goes to" as a 64-bit mask: what the exchanges are packed from in ONE pass over the elements for all
// destinations (a scan and a pack kernel per destination cost 0.25 ms per exchange and rank on S40M-tank at 8 ranks)
struct MaskDests {
    const unsigned long long* mask;
    __device__ unsigned long long operator()(uint64_t i) const { return mask[i]; }
};
struct VertexDests {  // the other holders of a vertex this rank owns
    const uint32_t* owner;
    const unsigned long long* holder;
    uint32_t me;
    __device__ unsigned long long operator()(uint64_t v) const { return owner[v] == me ? holder[v] : 0ull; }
};

// counts[slot * n_wg + workgroup] = elements of this workgroup's 256 that go to the slot-th destination of `active`
template <class Dests>
__global__ __launch_bounds__(256) void k_dest_counts(uint64_t n, Dests dests, unsigned long long active, uint32_t n_wg, uint32_t* __restrict__ counts) {
    __shared__ uint32_t s_cnt[64];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) s_cnt[tid] = 0u;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * 256u + (uint64_t)tid;
    const unsigned long long m = (i < n) ? (dests(i) & active) : 0ull;
    int slot = 0;
    for (unsigned long long a = active; a; a &= a - 1ull, ++slot) {
        const int q = __ffsll((long long)a) - 1;
        const uint32_t cq = (uint32_t)__popcll(__ballot((m >> q) & 1ull));
        if (lane == 0 && cq) atomicAdd(&s_cnt[slot], cq);
    }
    __syncthreads();
    if (tid < slot) counts[(size_t)tid * n_wg + blockIdx.x] = s_cnt[tid];
}

// rows[offs[slot * n_wg + workgroup] + rank inside the workgroup] = (id0 + i or ids[i], payload[i]): the rows of a destination are
// consecutive and in ascending element order, the destinations follow each other in the order of their ranks
template <class Dests>
__global__ __launch_bounds__(256) void k_pack_all(uint64_t n, Dests dests, unsigned long long active, uint32_t n_wg, const uint32_t* __restrict__ offs, uint64_t id0,
                                                  const unsigned long long* __restrict__ ids, const uint32_t* __restrict__ payload, int payload_words,
                                                  uint32_t* __restrict__ rows) {
    __shared__ uint32_t s_wave[64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + (uint64_t)tid;
    const unsigned long long m = (i < n) ? (dests(i) & active) : 0ull;
    int slot = 0;
    for (unsigned long long a = active; a; a &= a - 1ull, ++slot) {
        const int q = __ffsll((long long)a) - 1;
        const uint32_t cq = (uint32_t)__popcll(__ballot((m >> q) & 1ull));
        if (lane == 0) s_wave[slot][wave] = cq;
    }
    __syncthreads();
    if (!m) return;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long id = ids ? ids[i] : (unsigned long long)(id0 + i);
    slot = 0;
    for (unsigned long long a = active; a; a &= a - 1ull, ++slot) {
        const int q = __ffsll((long long)a) - 1;
        const bool bit = (m >> q) & 1ull;
        // (the lanes that left above hold nothing for anybody: the ballot of the remaining ones is the ballot of all)
        const unsigned long long bal = __ballot(bit);
        if (!bit) continue;
        uint32_t row = offs[(size_t)slot * n_wg + blockIdx.x] + (uint32_t)__popcll(bal & below);
        for (int w = 0; w < wave; ++w) row += s_wave[slot][w];
        uint32_t* dst = rows + (size_t)row * (size_t)(2 + payload_words);
        dst[0] = (uint32_t)id;
        dst[1] = (uint32_t)(id >> 32);
        for (int w = 0; w < payload_words; ++w) dst[2 + w] = payload[(size_t)i * payload_words + w];
    }
}

__global__ __launch_bounds__(256) void k_unpack_rows(uint64_t n, const uint32_t* __restrict__ rows, int payload_words, unsigned long long* __restrict__ ids,
                                                     uint32_t* __restrict__ payload) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = rows + (size_t)i * (size_t)(2 + payload_words);
    ids[i] = (unsigned long long)src[0] | ((unsigned long long)src[1] << 32);
    for (int w = 0; w < payload_words; ++w) payload[(size_t)i * payload_words + w] = src[2 + w];
}

__device__ inline long long dist_lower_bound(const unsigned long long* __restrict__ a, long long n, unsigned long long key) {
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// received (id, rho) rows -> rho[position of id among the held ids]
__global__ __launch_bounds__(256) void k_scatter_density(uint64_t n_rows, const uint32_t* __restrict__ rows, int payload_words, const unsigned long long* __restrict__ gids,
                                                         uint64_t n_held, uint32_t* __restrict__ rho_words, uint32_t* __restrict__ err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t* src = rows + (size_t)i * (size_t)(2 + payload_words);
    const unsigned long long id = (unsigned long long)src[0] | ((unsigned long long)src[1] << 32);
    const long long pos = dist_lower_bound(gids, (long long)n_held, id);
    if (pos >= (long long)n_held || gids[pos] != id) {
        atomicOr(err, 1u);  // a density arrived for a particle this rank does not hold
        return;
    }
    for (int w = 0; w < payload_words; ++w) rho_words[(size_t)pos * payload_words + w] = src[2 + w];
}

struct DistBricks {  // closed boxes of grid points of every rank's brick; world <= 64
    int lo[64][3], hi[64][3];
    int empty[64];
    int world;
};

// owner[v] = lowest rank whose brick holds the edge of vertex v; holder[v] = bit mask of all such ranks
__global__ __launch_bounds__(256) void k_vertex_owner(uint64_t nv, const unsigned long long* __restrict__ keys, DistBricks B, unsigned long long np1, unsigned long long np2,
                                                      uint32_t* __restrict__ owner, unsigned long long* __restrict__ holder) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const unsigned long long k = keys[v];  // ((gi*NPy + gj)*NPz + gk)*3 + axis
    const int axis = (int)(k % 3ull);
    const unsigned long long p = k / 3ull;
    const int g[3] = {(int)(p / (np2 * np1)), (int)((p / np2) % np1), (int)(p % np2)};
    unsigned long long mask = 0;
    uint32_t own = 0xFFFFFFFFu;
    for (int q = B.world - 1; q >= 0; --q) {
        if (B.empty[q]) continue;
        bool in = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) in = in && g[d] >= B.lo[q][d] && g[d] + (axis == d ? 1 : 0) <= B.hi[q][d];
        if (in) {
            mask |= 1ull << q;
            own = (uint32_t)q;
        }
    }
    owner[v] = own;
    holder[v] = mask;
}

__global__ __launch_bounds__(256) void k_owned_gids(uint64_t nv, const uint32_t* __restrict__ mine, const uint32_t* __restrict__ mine_off, unsigned long long voff,
                                                    unsigned long long* __restrict__ gid_local) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nv) gid_local[v] = mine[v] ? voff + mine_off[v] : ~0ull;
}

// vertices owned elsewhere: global id from the sorted (key, id) pairs the owners sent
__global__ __launch_bounds__(256) void k_resolve_shared(uint64_t nv, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ mine,
                                                        const unsigned long long* __restrict__ rkeys, const unsigned long long* __restrict__ rids, uint64_t n_recv,
                                                        unsigned long long* __restrict__ gid_local, uint32_t* __restrict__ err) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv || mine[v]) return;
    const long long pos = dist_lower_bound(rkeys, (long long)n_recv, keys[v]);
    if (pos >= (long long)n_recv || rkeys[pos] != keys[v]) {
        atomicOr(err, 2u);  // the owner rank did not emit this face vertex: level sets differ between ranks
        return;
    }
    gid_local[v] = rids[pos];
}

__global__ __launch_bounds__(256) void k_global_triangles(uint64_t n3, const uint32_t* __restrict__ tri32, const unsigned long long* __restrict__ gid_local,
                                                          unsigned long long* __restrict__ tri64) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) tri64[i] = gid_local[tri32[i]];
}

template <class R>
__global__ __launch_bounds__(256) void k_compact_owned(uint64_t nv, const uint32_t* __restrict__ mine, const uint32_t* __restrict__ mine_off, const R* __restrict__ vertices,
                                                       const unsigned long long* __restrict__ keys, R* __restrict__ vown, unsigned long long* __restrict__ kown) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv || !mine[v]) return;
    const size_t o = mine_off[v];
    vown[3 * o] = vertices[3 * v];
    vown[3 * o + 1] = vertices[3 * v + 1];
    vown[3 * o + 2] = vertices[3 * v + 2];
    kown[o] = keys[v];
}

inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// n 32-bit words of the device, widened into the mapped host words, then the mail (one thread: the release of the post orders its stores)
__global__ void k_post_words(const uint32_t* __restrict__ src, int n, unsigned long long* __restrict__ host_words, SSMailSlot m) {
    for (int i = 0; i < n; ++i) __hip_atomic_store(host_words + i, (unsigned long long)src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    ss_mail_post(m, (unsigned long long)n);
}

// (destination, workgroup) counts -> offsets; the first offset of every destination is its boundary in the send buffer
struct PackCountsIn {
    const uint32_t* counts;
    __device__ uint32_t operator()(uint32_t i) const { return counts[i]; }
};
struct PackOffsOut {
    uint32_t* offs;
    uint32_t* bounds;
    uint32_t n_wg;
    __device__ void operator()(uint32_t i, uint32_t, uint32_t excl) const {
        offs[i] = excl;
        if (i % n_wg == 0u) bounds[i / n_wg] = excl;
    }
};
// "this rank numbers vertex v": flag -> (flag array, rank among the flagged); the scan posts the count
struct MineOut {
    uint32_t* mine;
    uint32_t* mine_off;
    __device__ void operator()(uint32_t v, uint32_t x, uint32_t excl) const {
        mine[v] = x;
        mine_off[v] = excl;
    }
};
struct VertexFlagIn {
    VertexFlag f;
    __device__ uint32_t operator()(uint32_t v) const { return f((uint64_t)v); }
};
// "this rank computed the density of held particle i" (rho > 0: exactly the particles its brick owns); the scan posts their number
template <class R>
struct OwnedIn {
    const R* rho;
    __device__ uint32_t operator()(uint32_t i) const { return rho[i] > R(0.0) ? 1u : 0u; }
};
struct OwnedOut {
    uint32_t* owned;
    __device__ void operator()(uint32_t i, uint32_t x, uint32_t) const { owned[i] = x; }
};

// ---- join of the received (edge key, global id) rows: stable LSD sort of the 64-bit keys with the library's own 32-bit pair sort, low word
// first, then the high word's significant bits (ss_prims.h) ----
__global__ __launch_bounds__(256) void k_row_word(uint64_t n, const uint32_t* __restrict__ rows, int row_words, int word, const uint32_t* __restrict__ perm, uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rows[(size_t)(perm ? perm[i] : (uint32_t)i) * (size_t)row_words + (size_t)word];
}
__global__ __launch_bounds__(256) void k_rows_by_perm(uint64_t n, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ perm, unsigned long long* __restrict__ keys,
                                                      unsigned long long* __restrict__ ids) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = rows + (size_t)perm[i] * 4u;
    keys[i] = (unsigned long long)src[0] | ((unsigned long long)src[1] << 32);
    ids[i] = (unsigned long long)src[2] | ((unsigned long long)src[3] << 32);
}

template <class R> struct DistTypes;
template <> struct DistTypes<float> {
    using params = ss_params_f32; using grid = ss_grid_f32; using shard = ss_shard_f32;
    static ss_status grid_for_domain(const params* p, const float* a, const float* b, grid* g, grid* sg, float* m) { return ss_grid_for_domain_f32(p, a, b, g, sg, m); }
    static ss_status begin(ss_context* c, const float* x, uint64_t n, const params* p, const shard* s, ss_result* r) { return ss_shard_begin_f32(c, x, n, p, s, r); }
};
template <> struct DistTypes<double> {
    using params = ss_params_f64; using grid = ss_grid_f64; using shard = ss_shard_f64;
    static ss_status grid_for_domain(const params* p, const double* a, const double* b, grid* g, grid* sg, double* m) { return ss_grid_for_domain_f64(p, a, b, g, sg, m); }
    static ss_status begin(ss_context* c, const double* x, uint64_t n, const params* p, const shard* s, ss_result* r) { return ss_shard_begin_f64(c, x, n, p, s, r); }
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Packs, for every destination rank in `active`, the elements whose destination mask names it into consecutive rows of c->sendbuf
// (destinations in rank order, rows in ascending element order), exchanges them and leaves the received rows (ordered by source
// rank) in c->recvbuf.  One counting pass, one scan over (destination, workgroup) and one packing pass serve all destinations; the
// destinations' boundaries reach the host through a mail slot (no copy, no stream synchronisation).  `turn`: the caller's turn on a shared
// device (it may have launched the kernel that produces the masks inside it); released here before the ranks meet.
template <class Dests>
ss_status pack_and_exchange(ss_comm* c, TurnGuard& turn, uint64_t n, uint64_t id0, const unsigned long long* ids, const uint32_t* payload, int payload_words, Dests dests,
                            unsigned long long active, uint64_t* n_recv_rows, uint64_t* bytes_sent) {
    ss_context* ctx = c->ctx;
    hipStream_t st = ctx->stream;
    const int world = c->world, me = c->rank;
    const size_t row_bytes = (size_t)(2 + payload_words) * 4;
    if (!n) active = 0ull;
    std::vector<uint32_t> cnt((size_t)world, 0u);
    const int n_active = __builtin_popcountll(active);
    const uint64_t n_wg64 = (n + 255) / 256;
    if (n_wg64 * (uint64_t)std::max(n_active, 1) >= (1ull << 31)) return fail(ctx, SS_ERR_UNSUPPORTED, "exchange: too many (destination, workgroup) pairs for this build");
    const uint32_t n_wg = (uint32_t)n_wg64;
    const size_t n_pairs = (size_t)n_active * n_wg;
    std::vector<uint64_t> bound((size_t)n_active + 1, 0u);
    uint32_t* offs = nullptr;
    ss_status s = comm_ensure_mail(c);
    if (s != SS_OK) return s;
    if (n_active) {
        static_assert(COMM_MAIL_WORDS >= 66, "one boundary per destination plus the total");
        SS_HIP(ctx, c->offs.reserve((2 * n_pairs + 2) * 4 + 64));
        uint32_t* counts = c->offs.as<uint32_t>();
        offs = counts + n_pairs + 1;
        uint32_t* z = nullptr;  // zeroed: the scan's state, then the destinations' boundaries + the total
        s = comm_zeros(c, ss_scan_state_words(n_pairs) + (size_t)n_active + 2, &z);
        if (s != SS_OK) return s;
        uint32_t* bounds_dev = z + ss_scan_state_words(n_pairs);
        hipLaunchKernelGGL(k_dest_counts<Dests>, dim3(n_wg), dim3(256), 0, st, n, dests, active, n_wg, counts);
        ss_chained_scan<uint32_t, SSOpPlus>(PackCountsIn{counts}, PackOffsOut{offs, bounds_dev, n_wg}, (uint32_t)n_pairs, z, bounds_dev + n_active, SSMailSlot{}, st);
        const SSMailSlot m = comm_slot(c, 0);
        hipLaunchKernelGGL(k_post_words, dim3(1), dim3(1), 0, st, bounds_dev, n_active + 1, comm_words_dev(c), m);
        s = comm_mail_wait(c, m, nullptr, "the boundaries of an exchange");
        if (s != SS_OK) return s;
        const volatile unsigned long long* hw = comm_words_host(c);
        for (int sl = 0; sl <= n_active; ++sl) bound[(size_t)sl] = hw[sl];
    }
    std::vector<uint64_t> send_off((size_t)world + 1, 0), send_rows((size_t)world, 0);
    {
        int sl = 0;
        for (int q = 0; q < world; ++q) {
            if ((active >> q) & 1ull) {
                cnt[q] = (uint32_t)(bound[(size_t)sl + 1] - bound[(size_t)sl]);
                ++sl;
            }
            send_rows[q] = cnt[q];
            send_off[q + 1] = send_off[q] + (uint64_t)cnt[q] * row_bytes;
        }
    }
    SS_HIP(ctx, c->sendbuf.reserve(send_off[world] + 64));
    if (n_active && send_off[world])
        hipLaunchKernelGGL(k_pack_all<Dests>, dim3(n_wg), dim3(256), 0, st, n, dests, active, n_wg, offs, id0, ids, payload, payload_words, c->sendbuf.as<uint32_t>());
    turn.release();
    // matrix[r][q] = rows rank r sends to rank q
    std::vector<uint64_t> matrix((size_t)world * world, 0);
    s = comm_allgather_host(c, send_rows.data(), (size_t)world * 8, matrix.data());
    if (s != SS_OK) return s;
    ++c->info.n_collectives;
    std::vector<uint64_t> recv_off((size_t)world + 1, 0);
    for (int r = 0; r < world; ++r) recv_off[r + 1] = recv_off[r] + matrix[(size_t)r * world + me] * row_bytes;
    SS_HIP(ctx, c->recvbuf.reserve(recv_off[world] + 64));
    s = comm_exchange(c, c->sendbuf.as<uint8_t>(), send_off.data(), c->recvbuf.as<uint8_t>(), recv_off.data());
    if (s != SS_OK) return s;
    ++c->info.n_collectives;
    *n_recv_rows = recv_off[world] / row_bytes;
    if (bytes_sent) *bytes_sent += send_off[world] - (send_off[me + 1] - send_off[me]);
    {   // the busiest of this rank's point-to-point links in this exchange (ss_dist_info::bytes_link_max)
        uint64_t busiest = 0;
        for (int q = 0; q < world; ++q)
            if (q != me) busiest = std::max(busiest, std::max(send_off[q + 1] - send_off[q], recv_off[q + 1] - recv_off[q]));
        c->info.bytes_link_max += busiest;
    }
    return SS_OK;
}

template <class R>
ss_status dist_reconstruct(ss_comm* c, const R* xyz_in, uint64_t n_local, const typename DistTypes<R>::params* prm, ss_result* res) {
    using T = DistTypes<R>;
    if (!c || !prm || !res) return SS_ERR_INVALID_ARGUMENT;
    ss_context* ctx = c->ctx;
    if (res->ctx != ctx) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "result belongs to a different context than the communicator");
    if (c->world > 64) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 64 ranks are not supported by this build");
    if (prm->has_particle_aabb) return fail(ctx, SS_ERR_UNSUPPORTED, "particle_aabb cannot be combined with the multi-GPU reconstruction");
    if (n_local >= (1ull << 31)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^31-1 particles per rank");
    ctx->err.clear();
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int world = c->world, me = c->rank;
    c->assembled = false;
    c->is_f64 = sizeof(R) == 8;
    memset(&c->info, 0, sizeof(c->info));
    c->info.rank = me;
    c->info.world = world;
    double t0 = now_ms();

    // ---- 1. global ids, global AABB ----
    const R* d_xyz = xyz_in;
    if (n_local && !is_device_pointer(xyz_in)) {
        SS_HIP(ctx, c->xyz_in.reserve(n_local * 3 * sizeof(R) + 64));
        SS_HIP(ctx, hipMemcpyAsync(c->xyz_in.p, xyz_in, n_local * 3 * sizeof(R), hipMemcpyHostToDevice, st));
        d_xyz = c->xyz_in.as<R>();
    }
    struct Head { uint64_t n; double lo[3], hi[3]; uint64_t fb, not_finite; } mine_head, zero_head;
    memset(&zero_head, 0, sizeof(zero_head));
    mine_head = zero_head;
    mine_head.n = n_local;
    // every rank must hold the SAME partition-feedback state (flag, previous bricks, measured costs): ranks that disagree would cut different bricks
    // and exchange with the wrong peers.  A digest of the state travels with the head and is compared below.
    {
        uint64_t hsh = c->feedback ? 0x9E3779B97F4A7C15ull : 1ull;
        auto mix = [&](uint64_t v) { hsh = (hsh ^ v) * 0x100000001B3ull; hsh ^= hsh >> 29; };
        if (c->feedback) {
            for (double v : c->cost_per_particle) { uint64_t b; memcpy(&b, &v, 8); mix(b); }
            for (int64_t v : c->prev_bricks) mix((uint64_t)v);
            for (int d = 0; d < 3; ++d) { mix((uint64_t)c->prev_ns[d]); uint64_t b; memcpy(&b, &c->prev_origin[d], 8); mix(b); }
        }
        mine_head.fb = hsh;
    }
    ss_status s = comm_ensure_mail(c);
    if (s != SS_OK) return s;
    TurnGuard turn(c, "aabb");
    if (n_local) {
        // the six values land in the communicator's mapped host words, announced through a mail slot (no copy, no stream synchronisation)
        SS_HIP(ctx, ctx->aabb_partial.reserve(SS_AABB_PARTIAL_WORDS * sizeof(R)));
        const SSMailSlot m = comm_slot(c, 1);
        ss_launch_aabb<R>(d_xyz, (uint32_t)n_local, ctx->aabb_partial.as<R>(), reinterpret_cast<R*>(comm_words_dev(c) + 80), m, st);
        unsigned long long not_finite = 0;
        s = comm_mail_wait(c, m, &not_finite, "the bounding box of the local particles");
        if (s != SS_OK) return s;
        mine_head.not_finite = not_finite;  // (travels with the head: every rank must refuse the step together)
        const volatile R* h6 = reinterpret_cast<const volatile R*>(comm_words_host(c) + 80);
        for (int d = 0; d < 3; ++d) {
            mine_head.lo[d] = (double)h6[d];
            mine_head.hi[d] = (double)h6[3 + d];
        }
    }
    turn.release();
    std::vector<Head> heads((size_t)world);
    s = comm_allgather_host(c, &mine_head, sizeof(Head), heads.data());
    if (s != SS_OK) return s;
    ++c->info.n_collectives;
    uint64_t id0 = 0, n_total = 0;
    R dmin[3] = {0, 0, 0}, dmax[3] = {0, 0, 0};
    bool any = false;
    double pref[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < world; ++q)
        if (heads[q].not_finite)
            return fail(ctx, SS_ERR_INVALID_ARGUMENT, "particle coordinates must be finite (a rank's input holds a NaN or an infinity)", q);
    for (int q = 0; q < world; ++q)
        if (heads[q].fb != mine_head.fb)
            return fail(ctx, SS_ERR_INVALID_ARGUMENT, "the ranks disagree on the partition-feedback state (ss_comm_set_balance_feedback must be called on every rank, at the same step)", q);
    for (int q = 0; q < world; ++q) {
        if (q < me) id0 += heads[q].n;
        n_total += heads[q].n;
        if (!heads[q].n) continue;
        for (int d = 0; d < 3; ++d) {
            const R a = (R)heads[q].lo[d], b = (R)heads[q].hi[d];  // exact: they were R values
            dmin[d] = any ? ss_min(dmin[d], a) : a;
            dmax[d] = any ? ss_max(dmax[d], b) : b;
        }
        any = true;
    }
    for (int q = 0; q < world && any; ++q) {  // how far every rank's input extends along each axis: the tie-break of the bisection
        if (!heads[q].n) continue;
        for (int d = 0; d < 3; ++d) {
            const double span = std::max((double)dmax[d] - (double)dmin[d], 1e-300);
            pref[d] = std::max(pref[d], std::round(1000.0 * (heads[q].hi[d] - heads[q].lo[d]) / span) / 1000.0);
        }
    }
    c->info.n_total = n_total;
    typename T::grid grid, subgrid;
    R margin = 0;
    s = T::grid_for_domain(prm, dmin, dmax, &grid, &subgrid, &margin);
    if (s != SS_OK) return fail(ctx, s, "grid construction failed for the global domain");
    const int ns[3] = {(int)subgrid.n_cells[0], (int)subgrid.n_cells[1], (int)subgrid.n_cells[2]};
    const double nsub_d = (double)ns[0] * ns[1] * ns[2];
    if (nsub_d > 2.0e9) return fail(ctx, SS_ERR_UNSUPPORTED, "subdomain grid too large for the partition histogram");
    const size_t nsub = (size_t)nsub_d;
    for (int d = 0; d < 3; ++d) c->ns[d] = ns[d];
    c->n_cubes = (int)prm->subdomain_num_cubes_per_dim;

    // ---- 2. bricks from the all-reduced owner histogram ----
    SS_HIP(ctx, c->hist.reserve(nsub * 4 + 64));
    {
        TurnGuard hist_turn(c, "hist");
        SS_HIP(ctx, hipMemsetAsync(c->hist.p, 0, ss_round16(nsub * 4), st));  // (reserved with 64 bytes to spare)
        if (n_local)
            hipLaunchKernelGGL(k_owner_hist<R>, grid_for(n_local), dim3(256), 0, st, d_xyz, n_local, grid.aabb_min[0], grid.aabb_min[1], grid.aabb_min[2], subgrid.cell_size, ns[0],
                               ns[1], ns[2], c->hist.as<uint32_t>());
    }
    s = comm_allreduce_sum_u32(c, c->hist.as<uint32_t>(), nsub);
    if (s != SS_OK) return s;
    ++c->info.n_collectives;
    std::vector<uint32_t> h_hist(nsub);
    SS_HIP(ctx, hipMemcpyAsync(h_hist.data(), c->hist.p, nsub * 4, hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    std::vector<double> hist_d(nsub);
    for (size_t i = 0; i < nsub; ++i) hist_d[i] = (double)h_hist[i];
    // partition feedback: a subdomain's particles weigh what a particle cost the rank that owned the subdomain in the previous call (identical on every
    // rank: the costs were all-gathered as integers); subdomains outside the previous bricks (the domain moved) keep weight 1
    bool weighted = false;
    // (a domain whose origin moved: the previous bricks name other regions of space -- their weights are dropped)
    const bool same_origin = c->prev_origin[0] == (double)grid.aabb_min[0] && c->prev_origin[1] == (double)grid.aabb_min[1] && c->prev_origin[2] == (double)grid.aabb_min[2];
    if (c->feedback && same_origin && (int)c->cost_per_particle.size() == world && c->prev_bricks.size() == (size_t)world * 6 && c->prev_ns[0] == ns[0] && c->prev_ns[1] == ns[1] &&
        c->prev_ns[2] == ns[2]) {
        double mean = 0.0;
        int cnt = 0;
        for (int q = 0; q < world; ++q)
            if (c->cost_per_particle[q] > 0.0) mean += c->cost_per_particle[q], ++cnt;
        if (cnt) {
            weighted = true;
            mean /= (double)cnt;
            for (int q = 0; q < world; ++q) {
                const double w = c->cost_per_particle[q] > 0.0 ? std::min(4.0, std::max(0.25, c->cost_per_particle[q] / mean)) : 1.0;
                const int64_t* lo = &c->prev_bricks[(size_t)q * 6];
                const int64_t* hi = lo + 3;
                for (int64_t x = lo[0]; x < hi[0]; ++x)
                    for (int64_t y = lo[1]; y < hi[1]; ++y)
                        for (int64_t z = lo[2]; z < hi[2]; ++z) hist_d[((size_t)x * ns[1] + y) * ns[2] + z] *= w;
            }
        }
    }
    c->bricks.assign((size_t)world * 6, 0);
    {
        const int lo0[3] = {0, 0, 0};
        rcb_split(hist_d, ns, lo0, ns, 0, world, pref, 0.02, c->bricks);
    }
    if (weighted) {
        // hysteresis: bricks are whole subdomains, so a cut plane that flips between two frames moves several per cent of a rank's work and
        // regrows every buffer of both ranks -- the previous bricks stay unless the weighted histogram predicts a clearly better maximum
        auto worst = [&](const std::vector<int64_t>& b) {
            double mx = 0.0;
            for (int q = 0; q < world; ++q) {
                double t = 0.0;
                for (int64_t x = b[(size_t)q * 6]; x < b[(size_t)q * 6 + 3]; ++x)
                    for (int64_t y = b[(size_t)q * 6 + 1]; y < b[(size_t)q * 6 + 4]; ++y)
                        for (int64_t z = b[(size_t)q * 6 + 2]; z < b[(size_t)q * 6 + 5]; ++z) t += hist_d[((size_t)x * ns[1] + y) * ns[2] + z];
                mx = std::max(mx, t);
            }
            return mx;
        };
        if (!(worst(c->bricks) < 0.97 * worst(c->prev_bricks))) c->bricks = c->prev_bricks;
    }
    const int64_t* my_lo = &c->bricks[(size_t)me * 6];
    const int64_t* my_hi = my_lo + 3;
    for (int d = 0; d < 3; ++d) {
        c->info.brick_lo[d] = my_lo[d];
        c->info.brick_hi[d] = my_hi[d];
    }
    double amax = 1.0;
    for (int d = 0; d < 3; ++d) amax = std::max(amax, std::max(std::fabs((double)grid.aabb_min[d]), std::fabs((double)dmax[d])));
    const double pad = (double)margin * 1.001 + 1e-6 * amax;
    std::vector<DistBox> boxes((size_t)world);
    for (int q = 0; q < world; ++q) {
        DistBox& b = boxes[q];
        b.empty = 0;
        for (int d = 0; d < 3; ++d) {
            const int64_t a = c->bricks[(size_t)q * 6 + d], e = c->bricks[(size_t)q * 6 + 3 + d];
            if (e <= a) b.empty = 1;
            b.lo[d] = (double)grid.aabb_min[d] + (double)a * (double)subgrid.cell_size - pad;
            b.hi[d] = (double)grid.aabb_min[d] + (double)e * (double)subgrid.cell_size + pad;
        }
    }
    c->info.ms_partition = now_ms() - t0;
    t0 = now_ms();

    // ---- 3. positions to every rank that needs the particle (owner or ghost) ----
    uint64_t n_held = 0;
    const int pos_words = 3 * (int)(sizeof(R) / 4);
    // which boxes can hold anything of this rank's input, then ONE pass over the particles for all of them
    DistBoxes h_boxes;
    memset(&h_boxes, 0, sizeof(h_boxes));
    for (int q = 0; q < world; ++q) h_boxes.box[q] = boxes[q];
    SS_HIP(ctx, c->boxes_dev.reserve(sizeof(DistBoxes)));
    SS_HIP(ctx, hipMemcpyAsync(c->boxes_dev.p, &h_boxes, sizeof(DistBoxes), hipMemcpyHostToDevice, st));
    unsigned long long pos_active = 0;
    for (int q = 0; q < world && n_local; ++q) {
        if (boxes[q].empty) continue;
        bool overlap = true;
        for (int d = 0; d < 3; ++d)  // nothing of this rank's input can lie in a box that misses its bounding box
            if (mine_head.hi[d] < boxes[q].lo[d] || mine_head.lo[d] > boxes[q].hi[d]) overlap = false;
        if (overlap) pos_active |= 1ull << q;
    }
    SS_HIP(ctx, c->mask.reserve((n_local + 1) * 8 + 64));
    {
        TurnGuard pack_turn(c, "pack_pos");  // destination masks, counting, scan and packing are this rank's own work; released before the ranks meet
        if (pos_active)
            hipLaunchKernelGGL(k_box_masks<R>, grid_for(n_local), dim3(256), 0, st, d_xyz, n_local, c->boxes_dev.as<DistBoxes>(), pos_active, (const uint32_t*)nullptr,
                               c->mask.as<unsigned long long>());
        s = pack_and_exchange(c, pack_turn, n_local, id0, nullptr, reinterpret_cast<const uint32_t*>(d_xyz), pos_words, MaskDests{c->mask.as<unsigned long long>()}, pos_active, &n_held,
                              &c->info.bytes_sent_positions);
        if (s != SS_OK) return s;
    }
    if (n_held >= (1ull << 31)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^31-1 particles held by one rank");
    // Rows arrive ascending from every source rank and the ranks' id ranges ascend, so the concatenation by source rank IS the
    // ascending global-id order the engine needs (no sort).
    SS_HIP(ctx, c->gids.reserve(n_held * 8 + 64));
    SS_HIP(ctx, c->L.reserve(n_held * 3 * sizeof(R) + 64));
    TurnGuard phase1_turn(c, "unpack_phase1");  // unpacking, phase 1, the owned flags, the density masks and their packing are this rank's own work
    if (n_held)
        hipLaunchKernelGGL(k_unpack_rows, grid_for(n_held), dim3(256), 0, st, n_held, c->recvbuf.as<uint32_t>(), pos_words, c->gids.as<unsigned long long>(), c->L.as<uint32_t>());
    c->info.n_held = n_held;
    c->info.ms_position_exchange = now_ms() - t0;

    // ---- 4. phase 1: binning + densities of the particles contained in this brick ----
    typename T::shard shard;
    for (int d = 0; d < 3; ++d) {
        shard.domain_min[d] = dmin[d];
        shard.domain_max[d] = dmax[d];
        shard.sub_lo[d] = my_lo[d];
        shard.sub_hi[d] = my_hi[d];
    }
    t0 = now_ms();
    s = T::begin(ctx, c->L.as<R>(), n_held, prm, &shard, res);
    if (s != SS_OK) return s;
    c->info.ms_phase1 = now_ms() - t0;  // (host time: the phase's last kernels are still in flight, the stream is not drained here)
    t0 = now_ms();

    // ---- 5. halo densities: owners -> ranks holding the particle as a ghost ----
    SS_HIP(ctx, c->owned.reserve((n_held + 1) * 4 + 64));
    const SSMailSlot m_owned = comm_slot(c, 2);  // read at the end of the call: nothing of the flow depends on the count
    {
        uint32_t* z = nullptr;
        s = comm_zeros(c, ss_scan_state_words(n_held), &z);
        if (s != SS_OK) return s;
        ss_chained_scan<uint32_t, SSOpPlus>(OwnedIn<R>{res->rho.as<R>()}, OwnedOut{c->owned.as<uint32_t>()}, (uint32_t)n_held, z, (uint32_t*)nullptr, m_owned, st);
    }
    uint64_t n_rho_rows = 0;
    const int rho_words = (int)(sizeof(R) / 4);
    unsigned long long rho_active = 0;
    for (int q = 0; q < world && n_held; ++q) {
        if (q == me || boxes[q].empty || boxes[me].empty) continue;
        bool overlap = true;
        for (int d = 0; d < 3; ++d)  // held particles lie inside this rank's grown box: only overlapping boxes can want them
            if (boxes[me].hi[d] < boxes[q].lo[d] || boxes[me].lo[d] > boxes[q].hi[d]) overlap = false;
        if (overlap) rho_active |= 1ull << q;
    }
    SS_HIP(ctx, c->mask.reserve((n_held + 1) * 8 + 64));
    if (rho_active)
        hipLaunchKernelGGL(k_box_masks<R>, grid_for(n_held), dim3(256), 0, st, c->L.as<R>(), n_held, c->boxes_dev.as<DistBoxes>(), rho_active, c->owned.as<uint32_t>(),
                           c->mask.as<unsigned long long>());
    s = pack_and_exchange(c, phase1_turn, n_held, 0, c->gids.as<unsigned long long>(), reinterpret_cast<const uint32_t*>(res->rho.as<R>()), rho_words,
                          MaskDests{c->mask.as<unsigned long long>()}, rho_active, &n_rho_rows, &c->info.bytes_sent_densities);
    if (s != SS_OK) return s;
    TurnGuard phase2_turn(c, "scatter_phase2");  // scattering the received densities and phase 2
    SS_HIP(ctx, c->err.reserve(64));
    SS_HIP(ctx, hipMemsetAsync(c->err.p, 0, 4, st));
    if (n_rho_rows)
        hipLaunchKernelGGL(k_scatter_density, grid_for(n_rho_rows), dim3(256), 0, st, n_rho_rows, c->recvbuf.as<uint32_t>(), rho_words, c->gids.as<unsigned long long>(), n_held,
                           res->rho.as<uint32_t>(), c->err.as<uint32_t>());
    const SSMailSlot m_err = comm_slot(c, 3);  // (checked after phase 2: the stream is not held up for it; a stray density is dropped by the kernel)
    hipLaunchKernelGGL(k_post_words, dim3(1), dim3(1), 0, st, c->err.as<uint32_t>(), 1, comm_words_dev(c) + 72, m_err);
    res->hrho = false;
    c->info.ms_density_exchange = now_ms() - t0;

    // ---- 6. phase 2: level set + marching cubes of the brick ----
    t0 = now_ms();
    s = ss_shard_finish(ctx, res);
    if (s != SS_OK) return s;
    c->info.ms_phase2 = now_ms() - t0;
    phase2_turn.release();
    {
        unsigned long long v = 0;
        s = comm_mail_wait(c, m_err, nullptr, "the error flag of the density exchange");
        if (s != SS_OK) return s;
        if (comm_words_host(c)[72]) return fail(ctx, SS_ERR_UNKNOWN, "halo density exchange: a density arrived for a particle this rank does not hold");
        if (n_held) {
            s = comm_mail_wait(c, m_owned, &v, "the number of owned particles");
            if (s != SS_OK) return s;
        }
        c->info.n_owned = v;
    }
    {
        ss_stats st_;
        if (ss_result_stats(res, &st_) == SS_OK) c->info.ms_device = st_.ms_total;
    }

    // load balance of the partition, identical on every rank; with it travels what the step cost this rank (microseconds of device time): the
    // weights of the next call's partition (ss_comm_set_balance_feedback)
    uint64_t triple[3] = {c->info.n_owned, c->info.n_held, (uint64_t)std::llround(std::max(0.0, c->info.ms_device) * 1000.0)};
    std::vector<uint64_t> triples((size_t)world * 3);
    s = comm_allgather_host(c, triple, sizeof(triple), triples.data());
    if (s != SS_OK) return s;
    ++c->info.n_collectives;
    c->per_rank_owned.assign((size_t)world, 0);
    c->per_rank_held.assign((size_t)world, 0);
    std::vector<double> cpp((size_t)world, 0.0);
    for (int q = 0; q < world; ++q) {
        c->per_rank_owned[q] = triples[(size_t)q * 3];
        c->per_rank_held[q] = triples[(size_t)q * 3 + 1];
        // relative to the previous weights: the measured cost per particle of a brick is the product of all corrections so far
        cpp[q] = triples[(size_t)q * 3] ? (double)triples[(size_t)q * 3 + 2] / (double)triples[(size_t)q * 3] : 0.0;
    }
    c->cost_per_particle = cpp;
    c->prev_bricks = c->bricks;
    for (int d = 0; d < 3; ++d) c->prev_ns[d] = ns[d];
    for (int d = 0; d < 3; ++d) c->prev_origin[d] = (double)grid.aabb_min[d];
    return SS_OK;
}

template <class R>
ss_status dist_assemble(ss_comm* c, ss_result* res) {
    ss_context* ctx = c->ctx;
    if (!res->valid || res->phase != 2 || res->global_strategy) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "ss_dist_assemble needs the result of ss_dist_reconstruct");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int world = c->world, me = c->rank;
    const uint64_t nv = res->n_vertices, nt = res->n_triangles;
    const double t0 = now_ms();
    DistBricks B;
    memset(&B, 0, sizeof(B));
    B.world = world;
    const int n = c->n_cubes;
    for (int q = 0; q < world; ++q) {
        B.empty[q] = 0;
        for (int d = 0; d < 3; ++d) {
            const int64_t a = c->bricks[(size_t)q * 6 + d], e = c->bricks[(size_t)q * 6 + 3 + d];
            if (e <= a) B.empty[q] = 1;
            B.lo[q][d] = (int)(a * n);
            B.hi[q][d] = (int)(e * n);
        }
    }
    const unsigned long long np1 = (unsigned long long)c->ns[1] * n + 1ull, np2 = (unsigned long long)c->ns[2] * n + 1ull;
    SS_HIP(ctx, c->owner.reserve(nv * 4 + 64));
    SS_HIP(ctx, c->holder.reserve(nv * 8 + 64));
    SS_HIP(ctx, c->gid_local.reserve(nv * 8 + 64));
    SS_HIP(ctx, c->mine_off.reserve((nv + 1) * 4 + 64));
    SS_HIP(ctx, c->vals_a.reserve((nv + 1) * 4 + 64));  // mine flags
    const unsigned long long* keys = res->vkeys.as<unsigned long long>();
    if (nv >= SS_SCAN_MAX_N) return fail(ctx, SS_ERR_UNSUPPORTED, "too many vertices on one rank for the assembly");
    ss_status s = comm_ensure_mail(c);
    if (s != SS_OK) return s;
    TurnGuard turn(c, "asm_mine");
    if (nv) hipLaunchKernelGGL(k_vertex_owner, grid_for(nv), dim3(256), 0, st, nv, keys, B, np1, np2, c->owner.as<uint32_t>(), c->holder.as<unsigned long long>());
    uint32_t* mine = c->vals_a.as<uint32_t>();
    // "mine" flags and their ranks in one dispatch (the flag is the scan's input); the number of owned vertices arrives by mail
    const SSMailSlot m_mine = comm_slot(c, 4);
    {
        uint32_t* z = nullptr;
        s = comm_zeros(c, ss_scan_state_words(nv), &z);
        if (s != SS_OK) return s;
        ss_chained_scan<uint32_t, SSOpPlus>(VertexFlagIn{VertexFlag{c->owner.as<uint32_t>(), c->holder.as<unsigned long long>(), nv, (uint32_t)me, -1}},
                                            MineOut{mine, c->mine_off.as<uint32_t>()}, (uint32_t)nv, z, (uint32_t*)nullptr, m_mine, st);
    }
    unsigned long long v_owned = 0;
    s = comm_mail_wait(c, m_mine, &v_owned, "the number of vertices this rank numbers");
    if (s != SS_OK) return s;
    const uint32_t n_owned = (uint32_t)v_owned;
    turn.release();
    uint64_t cnt[2] = {n_owned, nt};
    std::vector<uint64_t> all((size_t)world * 2);
    s = comm_allgather_host(c, cnt, sizeof(cnt), all.data());
    if (s != SS_OK) return s;
    ++c->info.n_collectives;
    uint64_t voff = 0, toff = 0, vtot = 0, ttot = 0;
    for (int q = 0; q < world; ++q) {
        if (q < me) {
            voff += all[(size_t)q * 2];
            toff += all[(size_t)q * 2 + 1];
        }
        vtot += all[(size_t)q * 2];
        ttot += all[(size_t)q * 2 + 1];
    }
    // owners -> the other ranks holding the edge: (key, global id)
    uint64_t n_rows = 0;
    unsigned long long gid_active = 0ull;
    for (int q = 0; q < world && nv; ++q) {
        if (q == me || B.empty[q] || B.empty[me]) continue;
        bool touch = true;
        for (int d = 0; d < 3; ++d)  // closed point boxes that do not even touch share no edge
            if (B.hi[me][d] < B.lo[q][d] || B.lo[me][d] > B.hi[q][d]) touch = false;
        if (touch) gid_active |= 1ull << q;
    }
    {
        TurnGuard gid_turn(c, "asm_gid_pack");
        if (nv) hipLaunchKernelGGL(k_owned_gids, grid_for(nv), dim3(256), 0, st, nv, mine, c->mine_off.as<uint32_t>(), (unsigned long long)voff, c->gid_local.as<unsigned long long>());
        s = pack_and_exchange(c, gid_turn, nv, 0, keys, reinterpret_cast<const uint32_t*>(c->gid_local.as<unsigned long long>()), 2,
                              VertexDests{c->owner.as<uint32_t>(), c->holder.as<unsigned long long>(), (uint32_t)me}, gid_active, &n_rows, &c->info.bytes_sent_assembly);
        if (s != SS_OK) return s;
    }
    if (n_rows >= (1ull << 30)) return fail(ctx, SS_ERR_UNSUPPORTED, "too many shared face vertices received by one rank");
    TurnGuard join_turn(c, "asm_join");
    SS_HIP(ctx, c->err.reserve(64));
    SS_HIP(ctx, hipMemsetAsync(c->err.p, 0, 4, st));
    if (n_rows || nv) {
        // received rows (key lo, key hi, id lo, id hi): stable LSD sort of the 64-bit keys -- the low word with the library's 32-bit pair sort, then the
        // significant bits of the high word -- and every vertex owned elsewhere looks its key up
        SS_HIP(ctx, c->keys_b.reserve((n_rows + 1) * 8 + 64));
        SS_HIP(ctx, c->vals_b.reserve((n_rows + 1) * 8 + 64));
        unsigned long long* rk_sorted = c->keys_b.as<unsigned long long>();
        unsigned long long* ri_sorted = c->vals_b.as<unsigned long long>();
        if (n_rows) {
            const uint32_t nr = (uint32_t)n_rows;
            const uint32_t* rows = c->recvbuf.as<uint32_t>();
            unsigned hi_bits = 0;
            {   // keys are below 3 * (number of grid points)
                const long double kmax = 3.0L * (long double)((unsigned long long)c->ns[0] * n + 1ull) * (long double)np1 * (long double)np2;
                while (hi_bits < 32 && std::ldexp(1.0L, 32 + (int)hi_bits) < kmax) ++hi_bits;
            }
            const size_t work_words = (ss_radix_sort_work_words(nr, 32) + ss_radix_sort_work_words(nr, hi_bits ? hi_bits : 1) + 3) & ~(size_t)3;  // (whole 16-byte units: ss_round16)
            SS_HIP(ctx, c->sort_tmp.reserve(((size_t)nr + 16) * 4 * 4 + work_words * 4 + 64));
            uint32_t* k0 = c->sort_tmp.as<uint32_t>();
            uint32_t* k1 = k0 + ((size_t)nr + 16);
            uint32_t* v0 = k1 + ((size_t)nr + 16);
            uint32_t* v1 = v0 + ((size_t)nr + 16);
            uint32_t* work = v1 + ((size_t)nr + 16);
            SS_HIP(ctx, hipMemsetAsync(work, 0, work_words * 4, st));
            hipLaunchKernelGGL(k_row_word, grid_for(nr), dim3(256), 0, st, (uint64_t)nr, rows, 4, 0, (const uint32_t*)nullptr, k0);
            uint32_t* kk[2] = {k0, k1};
            uint32_t* vv[2] = {v0, v1};
            int r = ss_radix_sort_pairs(kk, vv, nr, 32, true, work, true, st);
            const uint32_t* perm = vv[r];
            if (hi_bits) {
                // the high words in the order of the low words; the permutation so far travels as the values
                uint32_t* kk2[2] = {kk[r ^ 1], kk[r]};  // (the sorted low words are not needed any more)
                uint32_t* vv2[2] = {vv[r], vv[r ^ 1]};
                hipLaunchKernelGGL(k_row_word, grid_for(nr), dim3(256), 0, st, (uint64_t)nr, rows, 4, 1, perm, kk2[0]);
                const int r2 = ss_radix_sort_pairs(kk2, vv2, nr, hi_bits, false, work + ss_radix_sort_work_words(nr, 32), true, st);
                perm = vv2[r2];
            }
            hipLaunchKernelGGL(k_rows_by_perm, grid_for(nr), dim3(256), 0, st, (uint64_t)nr, rows, perm, rk_sorted, ri_sorted);
        }
        if (nv)
            hipLaunchKernelGGL(k_resolve_shared, grid_for(nv), dim3(256), 0, st, nv, keys, mine, rk_sorted, ri_sorted, n_rows, c->gid_local.as<unsigned long long>(),
                               c->err.as<uint32_t>());
    }
    SS_HIP(ctx, c->tri64.reserve(nt * 24 + 64));
    if (nt) hipLaunchKernelGGL(k_global_triangles, grid_for(nt * 3), dim3(256), 0, st, nt * 3, res->tri32.as<uint32_t>(), c->gid_local.as<unsigned long long>(), c->tri64.as<unsigned long long>());
    SS_HIP(ctx, c->vown.reserve((size_t)n_owned * 3 * sizeof(R) + 64));
    SS_HIP(ctx, c->kown.reserve((size_t)n_owned * 8 + 64));
    if (nv) hipLaunchKernelGGL(k_compact_owned<R>, grid_for(nv), dim3(256), 0, st, nv, mine, c->mine_off.as<uint32_t>(), res->vertices.as<R>(), keys, c->vown.as<R>(), c->kown.as<unsigned long long>());
    const SSMailSlot m_err = comm_slot(c, 5);  // the last kernel of the assembly: its mail is the end of the step
    hipLaunchKernelGGL(k_post_words, dim3(1), dim3(1), 0, st, c->err.as<uint32_t>(), 1, comm_words_dev(c) + 73, m_err);
    s = comm_mail_wait(c, m_err, nullptr, "the end of the mesh assembly");
    if (s != SS_OK) return s;
    if (comm_words_host(c)[73]) return fail(ctx, SS_ERR_UNKNOWN, "mesh assembly: a shared face vertex was not emitted by its owner rank (level sets differ between ranks)");
    c->info.n_vertices_owned = n_owned;
    c->info.vertex_offset = voff;
    c->info.n_vertices_total = vtot;
    c->info.n_triangles = nt;
    c->info.triangle_offset = toff;
    c->info.n_triangles_total = ttot;
    c->info.ms_assembly = now_ms() - t0;
    c->assembled = true;
    return SS_OK;
}

ss_status copy_to_caller(ss_comm* c, const void* src, void* dst, size_t bytes) {
    ss_context* ctx = c->ctx;
    if (!bytes) return SS_OK;
    if (!dst) return SS_ERR_INVALID_ARGUMENT;
    SS_HIP(ctx, hipSetDevice(ctx->device));
    SS_HIP(ctx, hipMemcpyAsync(dst, src, bytes, is_device_pointer(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

void comm_release(ss_comm* c) {
    for (DevBuf* b : {&c->small_dev, &c->small_dev2, &c->red_tmp, &c->peers_dev, &c->xyz_in, &c->hist, &c->flags, &c->offs, &c->boxes_dev, &c->mask, &c->sendbuf, &c->recvbuf, &c->gids, &c->L, &c->owned,
                      &c->sort_tmp, &c->keys_a, &c->keys_b, &c->vals_a, &c->vals_b, &c->owner, &c->holder, &c->gid_local, &c->mine_off, &c->tri64, &c->vown, &c->kown, &c->err, &c->zeros})
        b->release();
    c->small_host.release();
    if (c->mail_host) (void)hipHostFree(c->mail_host);
    c->mail_host = c->mail_dev = nullptr;
}

double env_timeout() {
    const char* e = getenv("SPLASH_COMM_TIMEOUT_S");
    const double v = e ? atof(e) : 120.0;
    return v > 0.0 ? v : 120.0;
}

}  // namespace

// =====================================================================================================================
// C ABI
// =====================================================================================================================
extern "C" {

ss_status ss_comm_unique_id(uint8_t id[SS_COMM_ID_BYTES]) {
    if (!id) return SS_ERR_INVALID_ARGUMENT;
    RcclApi* api = rccl_api();
    if (!api->handle) return SS_ERR_DEVICE;
    ncclUniqueId u;
    if (api->GetUniqueId(&u) != ncclSuccess) return SS_ERR_DEVICE;
    static_assert(sizeof(u) == SS_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, SS_COMM_ID_BYTES);
    return SS_OK;
}

ss_status ss_comm_create_rccl(ss_context* ctx, const uint8_t id[SS_COMM_ID_BYTES], int rank, int world, ss_comm** out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return SS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(ctx, SS_ERR_DEVICE, api->error.empty() ? "RCCL not available" : api->error);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, SS_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t r = api->CommInitRank(&comm, world, u, rank);
    if (r != ncclSuccess) return fail(ctx, SS_ERR_DEVICE, std::string("ncclCommInitRank failed: ") + api->GetErrorString(r));
    ss_comm* c = new (std::nothrow) ss_comm();
    if (!c) return SS_ERR_UNKNOWN;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->kind = 1;
    c->nccl = comm;
    c->own_nccl = true;
    c->timeout_s = env_timeout();
    *out = c;
    return SS_OK;
}

ss_status ss_comm_adopt_rccl(ss_context* ctx, void* nccl_comm, int rank, int world, ss_comm** out) {
    if (!ctx || !nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return SS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    RcclApi* api = rccl_api();
    if (!api->handle) return fail(ctx, SS_ERR_DEVICE, api->error.empty() ? "RCCL not available" : api->error);
    ss_comm* c = new (std::nothrow) ss_comm();
    if (!c) return SS_ERR_UNKNOWN;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->kind = 1;
    c->nccl = reinterpret_cast<ncclComm_t>(nccl_comm);
    c->own_nccl = false;
    c->timeout_s = env_timeout();
    *out = c;
    return SS_OK;
}

ss_status ss_comm_create_local_group(ss_context* const* ctxs, int world, ss_comm** out) {
    if (!ctxs || !out || world < 1 || world > 64) return SS_ERR_INVALID_ARGUMENT;
    auto g = std::make_shared<LocalGroup>();
    g->world = world;
    g->host.resize((size_t)world);
    g->send_ptr.assign((size_t)world, nullptr);
    g->send_off.resize((size_t)world);
    g->red_ptr.assign((size_t)world, nullptr);
    for (int q = 0; q < world; ++q) {
        if (!ctxs[q]) return SS_ERR_INVALID_ARGUMENT;
        ss_comm* c = new (std::nothrow) ss_comm();
        if (!c) return SS_ERR_UNKNOWN;
        c->ctx = ctxs[q];
        c->rank = q;
        c->world = world;
        c->kind = 0;
        c->group = g;
        c->timeout_s = env_timeout();
        out[q] = c;
    }
    return SS_OK;
}

ss_status ss_comm_local_group_take_turns(ss_comm* c, int on) {
    if (!c || c->kind != 0 || !c->group) return SS_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->group->m);
    c->group->take_turns = on != 0;
    return SS_OK;
}

ss_status ss_comm_set_balance_feedback(ss_comm* c, int on) {
    if (!c) return SS_ERR_INVALID_ARGUMENT;
    c->feedback = on != 0;
    if (!c->feedback) c->cost_per_particle.clear();
    return SS_OK;
}

void ss_comm_destroy(ss_comm* c) {
    if (!c) return;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    comm_release(c);
    if (c->kind == 1 && c->nccl && c->own_nccl) (void)rccl_api()->CommDestroy(c->nccl);
    delete c;
}

ss_status ss_dist_reconstruct_f32(ss_comm* comm, const float* xyz_local, uint64_t n_local, const ss_params_f32* params, ss_result* inout) {
    return dist_reconstruct<float>(comm, xyz_local, n_local, params, inout);
}
ss_status ss_dist_reconstruct_f64(ss_comm* comm, const double* xyz_local, uint64_t n_local, const ss_params_f64* params, ss_result* inout) {
    return dist_reconstruct<double>(comm, xyz_local, n_local, params, inout);
}

ss_status ss_dist_assemble(ss_comm* comm, ss_result* inout) {
    if (!comm || !inout) return SS_ERR_INVALID_ARGUMENT;
    if (inout->ctx != comm->ctx) return fail(comm->ctx, SS_ERR_INVALID_ARGUMENT, "result belongs to a different context than the communicator");
    comm->ctx->err.clear();
    return inout->is_f64 ? dist_assemble<double>(comm, inout) : dist_assemble<float>(comm, inout);
}

ss_status ss_dist_partition(const uint32_t* histogram, const int64_t n_subdomains[3], int world, const double axis_preference[3], int64_t* bricks) {
    if (!histogram || !n_subdomains || !bricks || world < 1 || world > 64) return SS_ERR_INVALID_ARGUMENT;
    const int ns[3] = {(int)n_subdomains[0], (int)n_subdomains[1], (int)n_subdomains[2]};
    if (ns[0] < 1 || ns[1] < 1 || ns[2] < 1 || (double)ns[0] * ns[1] * ns[2] > 2.0e9) return SS_ERR_INVALID_ARGUMENT;
    const size_t nsub = (size_t)ns[0] * ns[1] * ns[2];
    std::vector<double> h(nsub);
    for (size_t i = 0; i < nsub; ++i) h[i] = (double)histogram[i];
    const double pref[3] = {axis_preference ? axis_preference[0] : 0.0, axis_preference ? axis_preference[1] : 0.0, axis_preference ? axis_preference[2] : 0.0};
    std::vector<int64_t> out((size_t)world * 6, 0);
    const int lo0[3] = {0, 0, 0};
    rcb_split(h, ns, lo0, ns, 0, world, pref, 0.02, out);
    for (size_t i = 0; i < out.size(); ++i) bricks[i] = out[i];
    return SS_OK;
}

ss_status ss_dist_get_info(const ss_comm* comm, ss_dist_info* out) {
    if (!comm || !out) return SS_ERR_INVALID_ARGUMENT;
    *out = comm->info;
    return SS_OK;
}

ss_status ss_dist_get_partition(const ss_comm* comm, int64_t* bricks, uint64_t* owned, uint64_t* held) {
    if (!comm || comm->bricks.size() != (size_t)comm->world * 6) return SS_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; bricks && i < comm->bricks.size(); ++i) bricks[i] = comm->bricks[i];
    for (int q = 0; q < comm->world; ++q) {
        if (owned) owned[q] = q < (int)comm->per_rank_owned.size() ? comm->per_rank_owned[q] : 0;
        if (held) held[q] = q < (int)comm->per_rank_held.size() ? comm->per_rank_held[q] : 0;
    }
    return SS_OK;
}

ss_status ss_dist_copy_global_ids(ss_comm* comm, uint64_t* dst) {
    if (!comm) return SS_ERR_INVALID_ARGUMENT;
    return copy_to_caller(comm, comm->gids.p, dst, (size_t)comm->info.n_held * 8);
}
ss_status ss_dist_copy_vertices(ss_comm* comm, void* dst) {
    if (!comm || !comm->assembled) return SS_ERR_INVALID_ARGUMENT;
    return copy_to_caller(comm, comm->vown.p, dst, (size_t)comm->info.n_vertices_owned * 3 * (comm->is_f64 ? 8 : 4));
}
ss_status ss_dist_copy_vertex_keys(ss_comm* comm, uint64_t* dst) {
    if (!comm || !comm->assembled) return SS_ERR_INVALID_ARGUMENT;
    return copy_to_caller(comm, comm->kown.p, dst, (size_t)comm->info.n_vertices_owned * 8);
}
ss_status ss_dist_copy_triangles(ss_comm* comm, uint64_t* dst) {
    if (!comm || !comm->assembled) return SS_ERR_INVALID_ARGUMENT;
    return copy_to_caller(comm, comm->tri64.p, dst, (size_t)comm->info.n_triangles * 24);
}

}  // extern "C"
