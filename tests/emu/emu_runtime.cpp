// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE (see include/hip/hip_runtime.h): fiber scheduler of the CPU execution model and the
// handful of HIP runtime calls the library makes, all synchronous.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <sys/mman.h>
#include <time.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#endif
#endif

namespace emu {

thread_local Lane* cur = nullptr;

enum : int { ST_READY = 0, ST_WAIT_WAVE = 1, ST_WAIT_BLOCK = 2, ST_DONE = 3 };

extern "C" void emu_ctx_switch(void** save_sp, void* new_sp);
asm(R"(
    .text
    .globl emu_ctx_switch
    .type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_ctx_switch, .-emu_ctx_switch
)");

static constexpr size_t STACK_BYTES = 256 * 1024;
static constexpr int MAX_THREADS = 1024;

// everything one OS thread needs to run workgroups: fiber stacks, lanes, the scheduler's own context
struct Worker {
    char* stacks = nullptr;
    Lane lanes[MAX_THREADS];
    Block blk;
    void* sched_sp = nullptr;
    const void* sched_stack_bottom = nullptr;  // (AddressSanitizer build: the OS thread's stack, learnt at the first switch into a fiber)
    size_t sched_stack_size = 0;
    void (*thunk)(void*) = nullptr;
    void* ctx = nullptr;
    std::vector<char> dyn;
    ~Worker() {
        if (stacks) munmap(stacks, STACK_BYTES * MAX_THREADS);
    }
};
static thread_local Worker* tl_worker = nullptr;
static thread_local Worker* tl_running = nullptr;  // the worker whose fibers run on this OS thread

// AddressSanitizer build (build_emu.py --asan): the sanitizer is told about every change of stack, so that its stack checks follow the fibers
static void to_scheduler() {
    Worker* w = tl_running;
    Lane* me = cur;
#ifdef EMU_ASAN
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(me->state == ST_DONE ? nullptr : &fake, w->sched_stack_bottom, w->sched_stack_size);
#endif
    emu_ctx_switch(&me->sp, w->sched_sp);
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

static void fiber_main() {
    Worker* w = tl_running;
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &w->sched_stack_bottom, &w->sched_stack_size);
#endif
    w->thunk(w->ctx);
    cur->state = ST_DONE;
    to_scheduler();
    abort();  // a finished fiber is never resumed
}

// scheduler side of a switch into lane L
static inline void resume_lane(Worker* w, Lane* L) {
    cur = L;
#ifdef EMU_ASAN
    void* fake = nullptr;
    const char* bottom = w->stacks + STACK_BYTES * (size_t)L->flat;
    __sanitizer_start_switch_fiber(&fake, bottom, STACK_BYTES);
#endif
    emu_ctx_switch(&w->sched_sp, L->sp);
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

void wave_sync(int op, int site) {
    Lane* me = cur;
    me->state = ST_WAIT_WAVE;
    me->op = op;
    me->site = site;
    to_scheduler();
}

int block_sync(int pred) {
    Lane* me = cur;
    me->state = ST_WAIT_BLOCK;
    me->x[me->parity][0] = pred ? 1ull : 0ull;
    to_scheduler();
    return cur->blk->barrier_or;
}

static bool g_warn_divergent = getenv("HIP_EMU_WARN_DIVERGENT") != nullptr;
// HIP_EMU_SHUFFLE=seed: workgroups in a scrambled order, the waves of a workgroup and the lanes between two synchronisation points in REVERSE order.  All of
// these are schedules the device may produce (tiles take their number from a ticket, not from blockIdx): an output that changes with the seed is a race.
static const unsigned long long g_shuffle = getenv("HIP_EMU_SHUFFLE") ? strtoull(getenv("HIP_EMU_SHUFFLE"), nullptr, 10) + 1ull : 0ull;

static void run_block(Worker* w, uint3 bid, dim3 grid, dim3 block, size_t shmem) {
    const int nt = (int)(block.x * block.y * block.z);
    if (nt > MAX_THREADS || nt <= 0) {
        fprintf(stderr, "[hip-emu] workgroup of %d threads\n", nt);
        abort();
    }
    if (!w->stacks) {
        w->stacks = (char*)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (w->stacks == MAP_FAILED) abort();
    }
    w->blk.bid = bid;
    w->blk.bdim = block;
    w->blk.gdim = grid;
    w->blk.barrier_or = 0;
    if (w->dyn.size() < shmem) w->dyn.resize(shmem);
    w->blk.dyn_smem = w->dyn.data();
    const int nwaves = (nt + 63) / 64;
    for (int t = 0; t < nt; ++t) {
        Lane& L = w->lanes[t];
        L.state = ST_READY;
        L.op = OP_NONE;
        L.site = 0;
        L.tid.x = (unsigned)t % block.x;
        L.tid.y = ((unsigned)t / block.x) % block.y;
        L.tid.z = (unsigned)t / (block.x * block.y);
        L.flat = (unsigned)t;
        L.lane = (unsigned)t & 63u;
        L.wave = (unsigned)t >> 6;
        L.group = 0;
        L.group_ballot = 0;
        L.parity = L.read_parity = 0;
        L.blk = &w->blk;
        L.wave_lanes = &w->lanes[t & ~63];
        // initial frame: six callee-saved registers, the entry point as return address, a null return address above it
        uintptr_t top = (uintptr_t)(w->stacks + STACK_BYTES * (size_t)(t + 1));
        top &= ~(uintptr_t)15;
        void** sp = (void**)(top - 64);
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        sp[6] = (void*)&fiber_main;
        sp[7] = nullptr;
        L.sp = sp;  // after the six pops and the ret: rsp = top - 8, i.e. 8 modulo 16 as at any function entry
#ifdef EMU_ASAN
        __asan_unpoison_memory_region(w->stacks + STACK_BYTES * (size_t)t, STACK_BYTES);  // (the frames the previous fiber on this stack never returned from)
#endif
    }
    tl_running = w;
    int live = nt;
    while (live > 0) {
        int at_barrier = 0;
        live = 0;
        for (int wv0 = 0; wv0 < nwaves; ++wv0) {
            const int wv = g_shuffle ? nwaves - 1 - wv0 : wv0;
            Lane* wl = &w->lanes[wv * 64];
            const int nl = std::min(64, nt - wv * 64);
            while (true) {
                for (int l0 = 0; l0 < nl; ++l0) {
                    const int l = g_shuffle ? nl - 1 - l0 : l0;
                    if (wl[l].state != ST_READY) continue;
                    resume_lane(w, &wl[l]);
                }
                // nobody of this wave can run: complete a wave-level operation if one is pending
                int site = 0x7fffffff;
                bool mixed = false;
                for (int l = 0; l < nl; ++l)
                    if (wl[l].state == ST_WAIT_WAVE) {
                        if (site != 0x7fffffff && wl[l].site != site) mixed = true;
                        site = std::min(site, wl[l].site);
                    }
                if (site == 0x7fffffff) break;
                if (mixed && g_warn_divergent) fprintf(stderr, "[hip-emu] lanes of one wave wait at different lines; completing line %d first\n", site);
                unsigned long long group = 0, ballot = 0;
                for (int l = 0; l < nl; ++l)
                    if (wl[l].state == ST_WAIT_WAVE && wl[l].site == site) {
                        group |= 1ull << l;
                        if (wl[l].x[wl[l].parity][0]) ballot |= 1ull << l;
                    }
                for (int l = 0; l < nl; ++l)
                    if ((group >> l) & 1ull) {
                        Lane& L = wl[l];
                        L.group = group;
                        L.group_ballot = ballot;
                        L.read_parity = L.parity;
                        L.parity ^= 1;
                        L.state = ST_READY;
                    }
            }
            for (int l = 0; l < nl; ++l) {
                if (wl[l].state == ST_WAIT_BLOCK) ++at_barrier;
                if (wl[l].state != ST_DONE) ++live;
            }
        }
        if (live == 0) break;
        if (at_barrier == 0) {
            fprintf(stderr, "[hip-emu] deadlock: %d live threads, none runnable, none at a barrier\n", live);
            abort();
        }
        // (threads that have exited do not take part in a barrier, like waves that have ended on the device)
        int any = 0;
        for (int t = 0; t < nt; ++t)
            if (w->lanes[t].state == ST_WAIT_BLOCK && w->lanes[t].x[w->lanes[t].parity][0]) any = 1;
        w->blk.barrier_or = any;
        for (int t = 0; t < nt; ++t)
            if (w->lanes[t].state == ST_WAIT_BLOCK) w->lanes[t].state = ST_READY;
    }
    tl_running = nullptr;
    cur = nullptr;
}

// ---- pool of OS threads -----------------------------------------------------------------------------------------------------------
struct Job {
    dim3 grid, block;
    size_t shmem = 0;
    void (*thunk)(void*) = nullptr;
    void* ctx = nullptr;
    std::atomic<unsigned long long> next{0};
    unsigned long long total = 0;
};
struct Pool {
    std::mutex launch_mu;  // one launch at a time (one device)
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    Job* job = nullptr;
    unsigned long long generation = 0;
    int busy = 0;
    bool stop = false;
    int nthreads = 1;
    Pool() {
        const char* e = getenv("HIP_EMU_THREADS");
        nthreads = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        if (nthreads < 1) nthreads = 1;
        if (nthreads > 64) nthreads = 64;
    }
    void start() {
        for (int i = 1; i < nthreads; ++i) threads.emplace_back([this] { loop(); });
    }
    static void work(Job* j) {
        if (!tl_worker) tl_worker = new Worker();
        while (true) {
            unsigned long long b = j->next.fetch_add(1, std::memory_order_relaxed);
            if (b >= j->total) break;
            if (g_shuffle && j->total > 1) {  // a bijection of [0, total): multiplication by a stride coprime to it, plus an offset
                static const unsigned long long primes[] = {7919ull, 104729ull, 1299709ull, 15485863ull, 179424673ull};
                unsigned long long stride = 1;
                for (unsigned long long pr : primes)
                    if (j->total % pr != 0) {
                        stride = pr % j->total;
                        break;
                    }
                if (stride == 0) stride = 1;
                b = (unsigned long long)(((unsigned __int128)b * stride + g_shuffle * 2654435761ull) % j->total);
            }
            uint3 bid;
            bid.x = (unsigned)(b % j->grid.x);
            bid.y = (unsigned)((b / j->grid.x) % j->grid.y);
            bid.z = (unsigned)(b / ((unsigned long long)j->grid.x * j->grid.y));
            tl_worker->thunk = j->thunk;
            tl_worker->ctx = j->ctx;
            run_block(tl_worker, bid, j->grid, j->block, j->shmem);
        }
    }
    void loop() {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        while (true) {
            cv_work.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            Job* j = job;
            lk.unlock();
            work(j);
            lk.lock();
            if (--busy == 0) cv_done.notify_all();
        }
    }
    void run(Job* j) {
        std::lock_guard<std::mutex> one(launch_mu);
        if (threads.empty() && nthreads > 1) start();
        const bool parallel = nthreads > 1 && j->total >= 4;
        if (parallel) {
            std::unique_lock<std::mutex> lk(mu);
            job = j;
            busy = (int)threads.size();
            ++generation;
            lk.unlock();
            cv_work.notify_all();
        }
        work(j);
        if (parallel) {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return busy == 0; });
            job = nullptr;
        }
    }
};
static Pool* pool() {
    static Pool* p = new Pool();  // (never destroyed: its threads may outlive static destruction at exit)
    return p;
}

static std::atomic<unsigned long long> g_launches{0};

static const bool g_trace = getenv("HIP_EMU_TRACE") != nullptr;  // one line per launch / memset / copy on stderr (what a call dispatches, in order)

void run_grid(const char* name, dim3 grid, dim3 block, size_t shmem, void (*thunk)(void*), void* ctx) {
    if (g_trace) fprintf(stderr, "[hip-emu] launch %s grid %u x %u x %u block %u\n", name, grid.x, grid.y, grid.z, block.x * block.y * block.z);
    if (cur) {
        fprintf(stderr, "[hip-emu] kernel launch from device code\n");
        abort();
    }
    Job j;
    j.grid = grid;
    j.block = block;
    j.shmem = shmem;
    j.thunk = thunk;
    j.ctx = ctx;
    j.total = (unsigned long long)grid.x * grid.y * grid.z;
    g_launches.fetch_add(1, std::memory_order_relaxed);
    if (j.total == 0) return;
    pool()->run(&j);
}

}  // namespace emu

// ---- runtime calls ------------------------------------------------------------------------------------------------------------------
struct emu_stream { int id; };
struct emu_event { double t_ms; };
namespace {
std::mutex g_mem_mu;
std::map<uintptr_t, size_t> g_dev_allocs;  // base -> size of what hipMalloc handed out ("device memory" for hipPointerGetAttributes)
const bool g_poison = getenv("HIP_EMU_NO_POISON") == nullptr;
// fault injection (tools/emu_fault_injection.py): the n-th hipMalloc / hipHostMalloc from now on fails with hipErrorOutOfMemory (0: none)
std::atomic<long long> g_fail_malloc_in{0}, g_fail_host_malloc_in{0};
std::atomic<unsigned long long> g_mallocs{0}, g_host_mallocs{0};
bool injected(std::atomic<long long>& countdown) {
    long long v = countdown.load();
    while (v > 0 && !countdown.compare_exchange_weak(v, v - 1)) {}
    return v == 1;
}
double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
}  // namespace

extern "C" {
hipError_t hipMalloc(void** p, size_t bytes) {
    if (!p) return hipErrorInvalidValue;
    g_mallocs.fetch_add(1);
    if (injected(g_fail_malloc_in)) return hipErrorOutOfMemory;
    void* q = nullptr;
    const size_t sz = bytes ? bytes : 1;
    if (posix_memalign(&q, 256, sz) != 0) return hipErrorOutOfMemory;
    if (g_poison) memset(q, 0xCD, sz);  // device memory is not zero-initialised: a kernel that relies on it reads garbage here
    std::lock_guard<std::mutex> lk(g_mem_mu);
    g_dev_allocs[(uintptr_t)q] = sz;
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mem_mu);
        auto it = g_dev_allocs.find((uintptr_t)p);
        if (it == g_dev_allocs.end()) return hipErrorInvalidValue;
        g_dev_allocs.erase(it);
    }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
    g_host_mallocs.fetch_add(1);
    if (injected(g_fail_host_malloc_in)) return hipErrorOutOfMemory;
    void* q = nullptr;
    if (posix_memalign(&q, 256, bytes ? bytes : 1) != 0) return hipErrorOutOfMemory;
    *p = q;
    return hipSuccess;
}
hipError_t hipHostFree(void* p) {
    free(p);
    return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) {
    *dev = host;
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) {
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) {
    if (emu::g_trace) fprintf(stderr, "[hip-emu] memcpyAsync %zu bytes\n", n);
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t n, hipStream_t) {
    if (emu::g_trace) fprintf(stderr, "[hip-emu] memsetAsync %zu bytes\n", n);
    if (n) memset(dst, value, n);
    return hipSuccess;
}
hipError_t hipMemset(void* dst, int value, size_t n) {
    if (n) memset(dst, value, n);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) {
    *st = new emu_stream{1};
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t st) {
    delete st;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* ev) {
    *ev = new emu_event{0.0};
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned) { return hipEventCreate(ev); }
hipError_t hipEventDestroy(hipEvent_t ev) {
    delete ev;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t) {
    ev->t_ms = now_ms();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = (float)(b->t_ms - a->t_ms);
    return hipSuccess;
}
// HIP_EMU_DEVICES=n: the process sees n "devices" (all of them this emulator: one process per rank selects its LOCAL_RANK, bench.py --gpus N dry runs)
static const int g_devices = getenv("HIP_EMU_DEVICES") ? (atoi(getenv("HIP_EMU_DEVICES")) > 0 ? atoi(getenv("HIP_EMU_DEVICES")) : 1) : 1;
hipError_t hipSetDevice(int dev) { return dev >= 0 && dev < g_devices ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceCount(int* n) {
    *n = g_devices;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) {
    switch (e) {
        case hipSuccess: return "no error";
        case hipErrorInvalidValue: return "invalid argument";
        case hipErrorOutOfMemory: return "out of memory";
        case hipErrorNotReady: return "not ready";
        default: return "unknown error";
    }
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* p) {
    std::lock_guard<std::mutex> lk(g_mem_mu);
    auto it = g_dev_allocs.upper_bound((uintptr_t)p);
    if (it != g_dev_allocs.begin()) {
        --it;
        if ((uintptr_t)p < it->first + it->second) {
            attr->type = hipMemoryTypeDevice;
            attr->device = 0;
            attr->devicePointer = (void*)p;
            attr->hostPointer = nullptr;
            return hipSuccess;
        }
    }
    return hipErrorInvalidValue;  // (what the real call answers for pageable host memory)
}
// how many kernel launches the emulator has executed (tests)
unsigned long long hip_emu_launch_count(void) { return emu::g_launches.load(); }
unsigned long long hip_emu_malloc_count(int host) { return host ? g_host_mallocs.load() : g_mallocs.load(); }
void hip_emu_fail_malloc_in(int host, long long n) { (host ? g_fail_host_malloc_in : g_fail_malloc_in).store(n); }
}
