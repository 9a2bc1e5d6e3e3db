// ss_prims.h -- the two data-parallel primitives of the reconstruction path, hand-written for gfx950 (wave64): a single-pass chained
// prefix sum whose input and output are functors (so that flag computation, compaction and the follow-up kernel fuse into the scan's
// one dispatch) and a stable least-significant-digit radix sort of (u32 key, u32 value) pairs, one dispatch per 8-bit digit
// ("onesweep": per-tile digit counts are chained by decoupled look-back inside the scatter kernel).  They replace rocPRIM's
// exclusive_scan / inclusive_scan / reduce / radix_sort_pairs on the path -- fewer dispatches per call (a scan was
// two dispatches plus a memset, a compaction a third, its count a copy kernel) and totals delivered straight into host-visible memory.
//
// Reference counterparts: the parallel prefix sums / sorts the reference takes from rayon and std (dense_subdomains.rs:476-488 sort_unstable of
// the per-subdomain particle lists; neighborhood_search.rs:679-710 cell map; dense_subdomains.rs:1693-1733 stitching offsets).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

// A count the host waits for, written by a kernel into pinned host memory that is mapped into the device's address space: the host polls
// `seq` instead of enqueueing a device-to-host copy (a dispatch of its own) and synchronising the stream.
struct SSMailSlot {
    unsigned long long* p = nullptr;  // device-visible address of {value, seq}
    unsigned long long seq = 0;       // what the kernel stores into p[1] after p[0]
};

__device__ __forceinline__ void ss_mail_post(const SSMailSlot& m, unsigned long long value) {
    if (!m.p) return;
    __hip_atomic_store(m.p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(m.p + 1, m.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- chained scan ----------------------------------------------------------------------------------------------------------------
// state: 2 + 2 * (number of tiles) 32-bit words, zeroed before the launch (word 0: tile counter; from word 2: one 64-bit status per tile).
// Tiles of 4096 elements (16 rows of 256 threads), of 1024 (4 rows) for small inputs, where the rows' fixed cost is the kernel's time, and of
// 8192 (16 rows of 512 threads) for large ones: a scan waits for its chain of tiles (one ticket atomic per tile on one address, the
// look-back over tiles that start together; profiles/r04_pmc_wait_counters_prims.csv), so fewer tiles are a shorter scan.
#define SS_SCAN_TILE 4096
#define SS_SCAN_TILE_SMALL 1024
#ifndef SS_SCAN_TILE_LARGE
#define SS_SCAN_TILE_LARGE 8192
#endif
#define SS_SCAN_SMALL_N 65536
#define SS_SCAN_LARGE_N (1u << 20)
inline size_t ss_scan_tile_of(size_t n) { return n <= SS_SCAN_SMALL_N ? SS_SCAN_TILE_SMALL : (n >= SS_SCAN_LARGE_N ? SS_SCAN_TILE_LARGE : SS_SCAN_TILE); }
inline size_t ss_scan_state_words(size_t n) { return 2 + 2 * ((n + ss_scan_tile_of(n) - 1) / ss_scan_tile_of(n)) + 2; }
// ss_scan_state_words is not monotone in n (the tile size changes at SS_SCAN_SMALL_N and SS_SCAN_LARGE_N: 132 words at n = 65536, 38 at 65537).
// A caller that reserves the state before it knows the count uses this bound: max of ss_scan_state_words(m) over all m <= n.
inline size_t ss_scan_state_words_bound(size_t n) {
    size_t w = ss_scan_state_words(n);
    if (n > SS_SCAN_SMALL_N) w = w > ss_scan_state_words(SS_SCAN_SMALL_N) ? w : ss_scan_state_words(SS_SCAN_SMALL_N);
    if (n >= SS_SCAN_LARGE_N) w = w > ss_scan_state_words(SS_SCAN_LARGE_N - 1) ? w : ss_scan_state_words(SS_SCAN_LARGE_N - 1);
    return w;
}
// The kernel indexes elements with 32 bits: the last tile's `base + r * NT + tid` must not wrap.
#define SS_SCAN_MAX_N (0xFFFFFFFFull - (unsigned long long)SS_SCAN_TILE_LARGE)

__device__ __forceinline__ uint32_t ss_prim_wave_incl_u32(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}
__device__ __forceinline__ unsigned long long ss_prim_wave_incl_u64(unsigned long long v) {
    const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}
struct SSOpPlus {
    template <class T> __device__ static T apply(T a, T b) { return a + b; }
    template <class T> __device__ static T identity() { return T(0); }
};
struct SSOpMax {
    template <class T> __device__ static T apply(T a, T b) { return a > b ? a : b; }
    template <class T> __device__ static T identity() { return T(0); }
};
template <class T, class Op>
__device__ __forceinline__ T ss_prim_wave_incl(T v) {
    if constexpr (sizeof(T) == 4 && std::is_same<Op, SSOpPlus>::value) {
        return (T)ss_prim_wave_incl_u32((uint32_t)v);
    } else {
        const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const T t = __shfl_up(v, off);
            if (lane >= off) v = Op::template apply<T>(t, v);
        }
        return v;
    }
}
template <class T, class Op>
__device__ __forceinline__ T ss_prim_wave_reduce(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = Op::template apply<T>(v, (T)__shfl_xor(v, off));
    return v;
}

// out(i, x, exclusive prefix of x) for i in [0, n), x = in(i); the total goes to *total_dev (if not null) and to the mail slot.
// One dispatch: tiles take their number from a counter in the order they start, publish their sum and look back over their
// predecessors' sums 64 tiles at a time.  A status word carries flag and value together, so the look-back needs no ordering
// against other memory: relaxed agent-scope atomics (an acquire / release at agent scope writes back and invalidates the L2 of
// the XCD on every access -- measured 346 us instead of ~40 for 10 M elements).  In / Out are callable from the device; T is
// uint32_t or unsigned long long (values below 2^62); Op: SSOpPlus or SSOpMax.
template <class T, class Op, int TILE, int NT, class In, class Out>
__global__ __launch_bounds__(NT) void k_chained_scan(In in, Out out, uint32_t n, uint32_t* __restrict__ state, T* __restrict__ total_dev, SSMailSlot mail) {
    constexpr int ROWS = TILE / NT, NW = NT / 64;
    static_assert(ROWS * NW <= 128 && TILE % NT == 0 && NT % 64 == 0, "the pieces of a tile are summed by one wave, at most two per lane");
    __shared__ T s_w[ROWS * NW + 64];  // piece (r, w) at r * NW + w (+ 64: the second piece of a lane past the end reads the identity)
    __shared__ T s_excl;
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(state, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * (uint32_t)TILE;
    unsigned long long* status = reinterpret_cast<unsigned long long*>(state + 2);
    // pass A: the inputs (kept in registers: In may be expensive or have side effects) and the sum of every (row, wave) piece
    T x[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint32_t i = base + (uint32_t)(r * NT + tid);
        x[r] = (i < n) ? in(i) : Op::template identity<T>();
        const T piece = ss_prim_wave_reduce<T, Op>(x[r]);
        if (lane == 0) s_w[r * NW + wave] = piece;
    }
    __syncthreads();
    T tile_total = Op::template identity<T>();
    if (wave == 0) {  // ROWS * NW pieces, at most two per lane
        T mine = lane < ROWS * NW ? s_w[lane] : Op::template identity<T>();
        if constexpr (ROWS * NW > 64) mine = Op::template apply<T>(mine, lane + 64 < ROWS * NW ? s_w[lane + 64] : Op::template identity<T>());
        tile_total = ss_prim_wave_reduce<T, Op>(mine);
    }
    if (wave == 0) {
        // publish the tile's sum, then look back: lane l reads the status of tile (p - l); flags 0 not there yet, 1 sum of that tile, 2 prefix up to and
        // including that tile
        const unsigned long long VALUE = (1ull << 62) - 1ull;
        if (lane == 0 && tile > 0) __hip_atomic_store(&status[tile], (1ull << 62) | (unsigned long long)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        T excl = Op::template identity<T>();
        long long p = (long long)tile - 1;
        while (true) {
            const long long idx = p - lane;
            const unsigned long long s = (idx >= 0) ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
            const unsigned flag = (unsigned)(s >> 62);
            const unsigned long long m_pref = __ballot(flag == 2u), m_zero = __ballot(flag == 0u);
            if (m_pref) {
                const int k = __ffsll((unsigned long long)m_pref) - 1;
                const unsigned long long upto = (k == 63) ? ~0ull : ((1ull << (k + 1)) - 1ull);
                if (m_zero & upto) {
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                const T v = ss_prim_wave_reduce<T, Op>((lane <= k) ? (T)(s & VALUE) : Op::template identity<T>());
                excl = Op::template apply<T>(v, excl);
                break;
            }
            if (m_zero) {
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            const T v = ss_prim_wave_reduce<T, Op>((T)(s & VALUE));
            excl = Op::template apply<T>(v, excl);
            p -= 64;
        }
        if (lane == 0) {
            const T incl_total = Op::template apply<T>(excl, tile_total);
            __hip_atomic_store(&status[tile], (2ull << 62) | (unsigned long long)incl_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
            if ((unsigned long long)base + (unsigned long long)TILE >= (unsigned long long)n) {  // the last tile knows the total (64 bits: base + TILE may pass 2^32)
                if (total_dev) *total_dev = incl_total;
                ss_mail_post(mail, (unsigned long long)incl_total);
            }
        }
    }
    __syncthreads();
    // pass B: everything before this thread = the tiles before (s_excl), the rows before, the waves before in this row, the lanes before
    T before_row = s_excl;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint32_t i = base + (uint32_t)(r * NT + tid);
        T before_wave = before_row;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const T piece = s_w[r * NW + w];
            if (w < wave) before_wave = Op::template apply<T>(before_wave, piece);
            before_row = Op::template apply<T>(before_row, piece);
        }
        const T incl = ss_prim_wave_incl<T, Op>(x[r]);
        T before_lane = (T)__shfl_up(incl, 1);
        if (lane == 0) before_lane = Op::template identity<T>();
        if (i < n) out(i, x[r], Op::template apply<T>(before_wave, before_lane));
    }
}

// Launch; n == 0 posts a total of 0 (one thread).  The state words must be zero.
template <class T, class Op = SSOpPlus, class In, class Out>
void ss_chained_scan(In in, Out out, uint32_t n, uint32_t* state, T* total_dev, SSMailSlot mail, hipStream_t st) {
    const uint32_t tile = (uint32_t)ss_scan_tile_of(n);
    const uint32_t tiles = n == 0 ? 1u : (n + tile - 1) / tile;
    if (tile == SS_SCAN_TILE_LARGE)
        hipLaunchKernelGGL((k_chained_scan<T, Op, SS_SCAN_TILE_LARGE, 512, In, Out>), dim3(tiles), dim3(512), 0, st, in, out, n, state, total_dev, mail);
    else if (tile == SS_SCAN_TILE)
        hipLaunchKernelGGL((k_chained_scan<T, Op, SS_SCAN_TILE, 256, In, Out>), dim3(tiles), dim3(256), 0, st, in, out, n, state, total_dev, mail);
    else
        hipLaunchKernelGGL((k_chained_scan<T, Op, SS_SCAN_TILE_SMALL, 256, In, Out>), dim3(tiles), dim3(256), 0, st, in, out, n, state, total_dev, mail);
}

// ---- radix sort ------------------------------------------------------------------------------------------------------------------
#define SS_RS_TILE 4096
// work buffer (32-bit words): zeroed by ss_radix_sort_pairs itself with one memset unless the caller says it is zero already
size_t ss_radix_sort_work_words(uint32_t n, unsigned bits);
// Stable sort of n (key, value) pairs by the low `bits` bits of the keys, 8 bits per pass, ping-pong between buffers 0 and 1.  iota: the
// values are the positions 0 .. n-1 (vals[0] is not read in the first pass, but is written by an even pass).  Returns the index (0 / 1)
// of the buffers that hold the result.  n < 2^30.
int ss_radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], uint32_t n, unsigned bits, bool iota, uint32_t* work, bool work_is_zero, hipStream_t st);
