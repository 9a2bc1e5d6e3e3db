// tests/emu/include/hip/hip_runtime.h -- TEST INFRASTRUCTURE, not part of the product.
//
// A CPU execution model of the HIP kernel language as splashsurf_amd/csrc uses it (wave64, gfx950), so that the library's kernels --
// the SAME source files, compiled with clang++ for x86-64 against this header instead of ROCm's -- can be run and checked against the
// oracle in a container that has no GPU:   python tests/emu/build_emu.py  ->  tests/emu/_build/libsplashsurf_emu.so.
// Nothing in splashsurf_amd/, bench.py or __graft_entry__ loads that library; only tests do (tests/test_emu_*.py), and they say so.
//
// Execution model (emu_runtime.cpp): a workgroup's threads are FIBERS on one OS thread, its waves are groups of 64 consecutive
// threads.  A lane runs until it reaches a wave-level operation (ballot, shuffle, DPP, readlane, MFMA, permlane swap, wave barrier)
// or a workgroup barrier; a wave-level operation completes when every lane of the wave that has not exited waits in one (lanes that
// wait at the SAME source line form the group: EXEC of the instruction), a barrier when every thread of the workgroup waits in it.
// Workgroups are independent and are handed to a pool of OS threads in ascending block index (a tile that looks back over its
// predecessors' status words finds them written or being written by a running thread -- the forward-progress guarantee of the
// decoupled look-back).  Arithmetic is the host's IEEE f32 / f64 with -ffp-contract=off and hardware fma; the two approximate
// instructions of the device (v_sqrt_f32, v_rcp_f32) are correctly rounded here; the MFMA sums its exact f16 products in double.
// What this does NOT model: timing, occupancy, the memory hierarchy, LDS capacity (checked against 64 KB + 96 KB only by a counter).
#pragma once
#define SS_HIP_EMU 1

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <tuple>
#include <type_traits>
#include <utility>

// ---- qualifiers -------------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIP_SYMBOL(x) (&(x))
#define amdgpu_waves_per_eu(...) unused /* __attribute__((amdgpu_waves_per_eu(a, b))) of the kernels: occupancy is not modelled */

// ---- vector types -----------------------------------------------------------------------------------------------------------------
#define EMU_VEC(T, name, al2, al4)                                                                         \
    struct alignas(al2) name##2 { T x, y; };                                                               \
    struct name##3 { T x, y, z; };                                                                         \
    struct alignas(al4) name##4 { T x, y, z, w; };                                                         \
    static inline name##2 make_##name##2(T x, T y) { return name##2{x, y}; }                               \
    static inline name##3 make_##name##3(T x, T y, T z) { return name##3{x, y, z}; }                       \
    static inline name##4 make_##name##4(T x, T y, T z, T w) { return name##4{x, y, z, w}; }
EMU_VEC(float, float, 8, 16)
EMU_VEC(double, double, 16, 32)
EMU_VEC(int, int, 8, 16)
EMU_VEC(unsigned, uint, 8, 16)
EMU_VEC(unsigned short, ushort, 4, 8)
EMU_VEC(unsigned char, uchar, 2, 4)
EMU_VEC(long long, longlong, 16, 32)
EMU_VEC(unsigned long long, ulonglong, 16, 32)
#undef EMU_VEC
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- runtime API (synchronous: a launch has finished when hipLaunchKernelGGL returns) ---------------------------------------------
typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 } hipError_t;
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
enum { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3, hipMemoryTypeUnregistered = 0 };
struct hipPointerAttribute_t { int type; int device; void* devicePointer; void* hostPointer; };
#define hipStreamNonBlocking 1u
#define hipHostMallocDefault 0u
#define hipHostMallocMapped 2u
#define hipEventDisableTiming 2u

extern "C" {
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t st);
hipError_t hipMemsetAsync(void* dst, int value, size_t n, hipStream_t st);
hipError_t hipMemset(void* dst, int value, size_t n);
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipStreamQuery(hipStream_t st);
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t ev, unsigned flags);
hipError_t hipDeviceSynchronize(void);
hipError_t hipEventCreate(hipEvent_t* ev);
hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t ev);
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t ev);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipSetDevice(int dev);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* p);
}
template <class T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc((void**)p, bytes); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags = 0) { return hipHostMalloc((void**)p, bytes, flags); }
template <class T> static inline hipError_t hipHostGetDevicePointer(T** dev, void* host, unsigned flags) { return hipHostGetDevicePointer((void**)dev, host, flags); }
template <class T> static inline hipError_t hipMemcpyToSymbol(T* sym, const void* src, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) {
    memcpy((char*)sym + off, src, n);
    return hipSuccess;
}
template <class T> static inline hipError_t hipMemcpyFromSymbol(void* dst, const T* sym, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) {
    memcpy(dst, (const char*)sym + off, n);
    return hipSuccess;
}

// ---- the lane a fiber runs as -----------------------------------------------------------------------------------------------------
namespace emu {
enum : int { OP_NONE = 0, OP_BALLOT = 1, OP_XCHG = 2, OP_BARRIER = 3 };
struct Block;
struct Lane {
    void* sp;                 // saved stack pointer of the fiber
    int state, op, site;      // scheduler state; kind and source line of the wave-level operation the lane waits in
    uint3 tid;
    unsigned flat, lane, wave;
    unsigned long long group;         // lanes taking part in the operation the lane was released from (EXEC)
    unsigned long long group_ballot;  // OP_BALLOT: the predicate bits of the group
    unsigned char parity, read_parity;
    alignas(16) unsigned long long x[2][4];  // what the lane shows the others in an operation: two slots, so that a lane that runs ahead into its
                                             // next operation does not overwrite what the lanes behind it still read
    Block* blk;
    Lane* wave_lanes;  // lane 0 of this lane's wave
};
struct Block {
    uint3 bid;
    dim3 bdim, gdim;
    char* dyn_smem;
    int barrier_or;  // result of the last barrier's predicate (written by the scheduler when it releases the barrier)
};
extern thread_local Lane* cur;
void wave_sync(int op, int site);  // the lane has filled cur->x[cur->parity]; returns when its group is released
int block_sync(int pred);          // returns the OR of the predicates of all threads of the workgroup
void run_grid(const char* name, dim3 grid, dim3 block, size_t shmem, void (*thunk)(void*), void* ctx);

template <class... P, class... A>
static inline void launch(const char* name, void (*kern)(P...), dim3 grid, dim3 block, size_t shmem, A&&... a) {
    std::tuple<std::decay_t<P>...> params(static_cast<std::decay_t<P>>(std::forward<A>(a))...);
    struct Ctx { void (*kern)(P...); std::tuple<std::decay_t<P>...>* params; } ctx{kern, &params};
    run_grid(name, grid, block, shmem, [](void* c) { Ctx* x = (Ctx*)c; std::apply(x->kern, *x->params); }, &ctx);
}
// the value lane `src` showed in the operation this lane was released from; ok = false: src was not part of it
template <class T>
static inline T peek(int src, bool* ok) {
    static_assert(sizeof(T) <= 32, "exchange slot");
    Lane* me = cur;
    if (src < 0 || src > 63 || !((me->group >> src) & 1ull)) {
        *ok = false;
        return T();
    }
    const Lane* s = me->wave_lanes + src;
    T v;
    memcpy(&v, s->x[s->read_parity], sizeof(T));
    *ok = true;
    return v;
}
template <class T>
static inline void show(const T& v) {
    static_assert(sizeof(T) <= 32, "exchange slot");
    memcpy(cur->x[cur->parity], &v, sizeof(T));
}
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->blk->bid)
#define blockDim (emu::cur->blk->bdim)
#define gridDim (emu::cur->blk->gdim)
#define warpSize 64
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) emu::launch(#kern, (kern), dim3(grid), dim3(block), (size_t)(shmem), ##__VA_ARGS__)

// ---- integer / bit helpers --------------------------------------------------------------------------------------------------------
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline long long __double_as_longlong(double d) { long long u; memcpy(&u, &d, 8); return u; }
static inline double __longlong_as_double(long long u) { double d; memcpy(&d, &u, 8); return d; }
#define EMU_MINMAX(T)                                             \
    static inline T min(T a, T b) { return b < a ? b : a; }       \
    static inline T max(T a, T b) { return a < b ? b : a; }
EMU_MINMAX(int)
EMU_MINMAX(unsigned)
EMU_MINMAX(long)
EMU_MINMAX(unsigned long)
EMU_MINMAX(long long)
EMU_MINMAX(unsigned long long)
#undef EMU_MINMAX
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

// ---- atomics (workgroups run on several OS threads) -------------------------------------------------------------------------------
template <class T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicMax(T* p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T> static inline T atomicMin(T* p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T> static inline T atomicCAS(T* p, T cmp, T v) {
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return cmp;
}
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), (order))
#define __hip_atomic_exchange(p, v, order, scope) __atomic_exchange_n((p), (v), (order))
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_s_sleep(n) __builtin_ia32_pause()
static inline unsigned long long emu_memtime() { return __builtin_ia32_rdtsc(); }
#define __builtin_amdgcn_s_memtime() emu_memtime()

// ---- workgroup and wave level operations ------------------------------------------------------------------------------------------
static inline void __syncthreads() { (void)emu::block_sync(0); }
static inline int __syncthreads_or(int pred) { return emu::block_sync(pred); }
static inline unsigned __lane_id() { return emu::cur->lane; }
static inline unsigned long long __ballot(int pred, int site = __builtin_LINE()) {
    emu::cur->x[emu::cur->parity][0] = pred ? 1ull : 0ull;
    emu::wave_sync(emu::OP_BALLOT, site);
    return emu::cur->group_ballot;
}
static inline unsigned long long __activemask(int site = __builtin_LINE()) { return __ballot(1, site); }
static inline int __any(int pred, int site = __builtin_LINE()) { return __ballot(pred, site) != 0ull; }
static inline int __all(int pred, int site = __builtin_LINE()) { return __ballot(!pred, site) == 0ull; }
#define __builtin_amdgcn_wave_barrier() emu::wave_sync(emu::OP_BARRIER, __LINE__)
#define __builtin_amdgcn_mbcnt_lo(mask, base) ((unsigned)(base) + (unsigned)__builtin_popcountll((unsigned long long)(unsigned)(mask) & ((1ull << emu::cur->lane) - 1ull) & 0xFFFFFFFFull))
#define __builtin_amdgcn_mbcnt_hi(mask, base) ((unsigned)(base) + (unsigned)__builtin_popcountll(((unsigned long long)(unsigned)(mask) << 32) & ((1ull << emu::cur->lane) - 1ull)))

template <class T>
static inline T emu_shfl_from(T v, int src, int site) {  // value of lane src (own value when src takes no part)
    emu::show(v);
    emu::wave_sync(emu::OP_XCHG, site);
    bool ok;
    const T r = emu::peek<T>(src, &ok);
    return ok ? r : v;
}
template <class T> static inline T __shfl(T v, int src, int width = 64, int site = __builtin_LINE()) {
    const int lane = (int)emu::cur->lane;
    return emu_shfl_from(v, (lane & ~(width - 1)) + (src & (width - 1)), site);
}
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64, int site = __builtin_LINE()) {
    const int lane = (int)emu::cur->lane, src = lane - (int)delta;
    return emu_shfl_from(v, src < (lane & ~(width - 1)) ? lane : src, site);
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64, int site = __builtin_LINE()) {
    const int lane = (int)emu::cur->lane, src = lane + (int)delta;
    return emu_shfl_from(v, src > (lane | (width - 1)) ? lane : src, site);
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64, int site = __builtin_LINE()) {
    const int lane = (int)emu::cur->lane, src = lane ^ mask;
    return emu_shfl_from(v, src > (lane | (width - 1)) ? lane : src, site);
}
static inline int emu_readlane(int v, int src, int site) {
    emu::show(v);
    emu::wave_sync(emu::OP_XCHG, site);
    bool ok;
    const int r = emu::peek<int>(src, &ok);
    return ok ? r : 0;
}
static inline int emu_readfirstlane(int v, int site) {
    emu::show(v);
    emu::wave_sync(emu::OP_XCHG, site);
    bool ok;
    return emu::peek<int>(__builtin_ctzll(emu::cur->group), &ok);
}
#define __builtin_amdgcn_readlane(v, l) emu_readlane((int)(v), (int)(l), __LINE__)
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane((int)(v), __LINE__)

// v_mov_b32_dpp: the controls the library uses (row_shr:n, row_shl:n, row_ror:n, wave shifts, mirrors, row broadcasts, quad_perm)
static inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int site) {
    emu::show(src);
    emu::wave_sync(emu::OP_XCHG, site);
    const int lane = (int)emu::cur->lane, row = lane >> 4, in_row = lane & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x000 && ctrl <= 0x0FF) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);            // quad_perm
    else if (ctrl >= 0x101 && ctrl <= 0x10F) from = in_row + (ctrl & 15) <= 15 ? lane + (ctrl & 15) : -1;  // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11F) from = in_row - (ctrl & 15) >= 0 ? lane - (ctrl & 15) : -1;   // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12F) from = (lane & ~15) | ((in_row - (ctrl & 15)) & 15);          // row_ror
    else if (ctrl == 0x130) from = lane + 1 <= 63 ? lane + 1 : -1;                                         // wave_shl:1
    else if (ctrl == 0x134) from = (lane + 1) & 63;                                                        // wave_rol:1
    else if (ctrl == 0x138) from = lane - 1;                                                               // wave_shr:1
    else if (ctrl == 0x13C) from = (lane - 1) & 63;                                                        // wave_ror:1
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - in_row);                                           // row_mirror
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));                                         // row_half_mirror
    else if (ctrl == 0x142) from = row > 0 ? ((row - 1) << 4) | 15 : -1;                                   // row_bcast:15
    else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;                                                     // row_bcast:31
    else abort();
    bool ok = false;
    const int v = from >= 0 ? emu::peek<int>(from, &ok) : 0;
    return ok ? v : (bound_ctrl ? 0 : old);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_update_dpp((int)(old), (int)(src), (ctrl), (rm), (bm), (bc), __LINE__)

// v_permlane32_swap_b32 vdst, vsrc: lanes 32-63 of vdst change places with lanes 0-31 of vsrc; returns {vdst, vsrc}
typedef unsigned emu_u32x2 __attribute__((ext_vector_type(2)));
static inline emu_u32x2 emu_permlane32_swap(unsigned vdst, unsigned vsrc, int site) {
    const emu_u32x2 mine = {vdst, vsrc};
    emu::show(mine);
    emu::wave_sync(emu::OP_XCHG, site);
    const int lane = (int)emu::cur->lane;
    bool ok;
    const emu_u32x2 other = emu::peek<emu_u32x2>(lane ^ 32, &ok);
    if (!ok) return mine;
    if (lane < 32) return emu_u32x2{vdst, other[0]};
    return emu_u32x2{other[1], vsrc};
}
#define __builtin_amdgcn_permlane32_swap(vdst, vsrc, fi, bc) emu_permlane32_swap((vdst), (vsrc), __LINE__)

// v_mfma_f32_32x32x8_f16: A row i = lane i (k 0-3) and lane i + 32 (k 4-7), B column n likewise, D[i][n] in lane n + 32 ((i >> 2) & 1),
// register (i & 3) + 4 (i >> 3).  f16 products are exact in f32; summed here in double and rounded once.
typedef _Float16 emu_half4 __attribute__((ext_vector_type(4)));
typedef float emu_float16 __attribute__((ext_vector_type(16)));
static inline emu_float16 emu_mfma_32x32x8f16(emu_half4 a, emu_half4 b, emu_float16 c, int site) {
    struct AB { emu_half4 a, b; } mine{a, b};
    emu::show(mine);
    emu::wave_sync(emu::OP_XCHG, site);
    const int lane = (int)emu::cur->lane, n = lane & 31, hi = lane >> 5;
    bool ok0, ok1;
    const AB b_lo = emu::peek<AB>(n, &ok0), b_hi = emu::peek<AB>(n + 32, &ok1);
    emu_float16 d;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const AB a_lo = emu::peek<AB>(i, &ok0), a_hi = emu::peek<AB>(i + 32, &ok1);
        double s = (double)c[r];
        for (int k = 0; k < 4; ++k) s += (double)(float)a_lo.a[k] * (double)(float)b_lo.b[k];
        for (int k = 0; k < 4; ++k) s += (double)(float)a_hi.a[k] * (double)(float)b_hi.b[k];
        d[r] = (float)s;
    }
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, x, y, z) emu_mfma_32x32x8f16((a), (b), (c), __LINE__)

// ---- arithmetic builtins of the device --------------------------------------------------------------------------------------------
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)   /* v_sqrt_f32 (<= 1 ulp on the device): correctly rounded here */
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
static inline float emu_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
#define __builtin_amdgcn_fmed3f(a, b, c) emu_fmed3f((a), (b), (c))
// v_sub_f32 with the clamp output modifier (ss_device.h ss_sub_clamp)
static inline float emu_clamp01(float x) { return !(x > 0.0f) ? 0.0f : (x > 1.0f ? 1.0f : x); }  // (a NaN clamps to 0: DX10_CLAMP)
