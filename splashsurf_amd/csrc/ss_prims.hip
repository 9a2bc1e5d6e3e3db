// ss_prims.hip -- stable LSD radix sort of (u32 key, u32 value) pairs for gfx950, one dispatch per 8-bit digit (see ss_prims.h).
//
// Per digit ("pass") one kernel, k_rs_pass: a 256-thread workgroup takes a tile of 4096 pairs (wave w the w-th quarter, 16 rounds of 64),
// ranks every key among the keys of the same digit that precede it in the tile (wave-level match by eight ballots, per-wave digit
// counters in LDS), publishes the tile's 256 digit counts and obtains, per digit, the number of such keys in all preceding tiles by
// decoupled look-back (thread d walks back over the published counts of digit d), stages the tile in LDS in digit order and writes
// every digit's run to its place -- consecutive lanes write consecutive addresses.  The global digit histograms of all passes come
// from one kernel up front (k_rs_hist).  Stable: equal keys keep their input order, which the callers rely on (ascending particle
// index inside a cell = the reference's order, neighborhood_search.rs:692-707).
#include "ss_prims.h"

#define RS_ROUNDS 16  // pairs per thread

// Digit histograms of all passes.  The keys of a wave often agree in their high digits (spatially coherent input): LDS atomics of all 64
// lanes on one counter serialise, so every group of lanes with the same (lane & 7) has its own copy of the counters (8-way instead of
// 64-way conflicts; matching equal digits with ballots first costs as much VALU time as it saves).
template <int COPIES>
__global__ __launch_bounds__(256) void k_rs_hist(const uint32_t* __restrict__ keys, uint32_t n, int npass, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[4][COPIES][256];
    const int tid = threadIdx.x, c = tid & (COPIES - 1);
    for (int p = 0; p < npass; ++p)
        for (int j = 0; j < COPIES; ++j) h[p][j][tid] = 0u;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 256u;
    size_t i = (size_t)blockIdx.x * 256u + (size_t)tid;
    // four keys in flight per thread
    for (; i + 3 * stride < (size_t)n; i += 4 * stride) {
        const uint32_t k0 = keys[i], k1 = keys[i + stride], k2 = keys[i + 2 * stride], k3 = keys[i + 3 * stride];
        for (int p = 0; p < npass; ++p) {
            atomicAdd(&h[p][c][(k0 >> (8 * p)) & 255u], 1u);
            atomicAdd(&h[p][c][(k1 >> (8 * p)) & 255u], 1u);
            atomicAdd(&h[p][c][(k2 >> (8 * p)) & 255u], 1u);
            atomicAdd(&h[p][c][(k3 >> (8 * p)) & 255u], 1u);
        }
    }
    for (; i < (size_t)n; i += stride) {
        const uint32_t k = keys[i];
        for (int p = 0; p < npass; ++p) atomicAdd(&h[p][c][(k >> (8 * p)) & 255u], 1u);
    }
    __syncthreads();
    for (int p = 0; p < npass; ++p) {
        uint32_t t = 0;
        for (int j = 0; j < COPIES; ++j) t += h[p][j][tid];
        if (t) atomicAdd(&hist[p * 256 + tid], t);
    }
}

// exclusive prefix over the NT threads of the workgroup (s_tmp: NT / 64 words)
template <int NT>
__device__ __forceinline__ uint32_t rs_block_excl(uint32_t v, uint32_t* s_tmp, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t incl = ss_prim_wave_incl_u32(v);
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += s_tmp[w];
    __syncthreads();
    return base + incl - v;
}

// NT threads, a tile of 16 NT pairs; the tile is staged through LDS twice (keys, then values: one 4-byte array of the tile's size)
template <int NT>
__global__ __launch_bounds__(NT) void k_rs_pass(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                uint32_t n, int shift, const uint32_t* __restrict__ hist, uint32_t* __restrict__ tile_counter, uint32_t* __restrict__ status) {
    constexpr int NW = NT / 64, TILE = NT * RS_ROUNDS, PER_WAVE = TILE / NW;
    __shared__ uint32_t s_cnt[NW][256];  // per wave: keys of digit d seen so far; afterwards: the wave's offset inside the tile's run of digit d
    __shared__ uint32_t s_lbase[256];    // start of digit d's run in the staged tile
    __shared__ uint32_t s_gbase[256];    // start of the tile's run of digit d in the output
    __shared__ uint32_t s_stage[TILE];
    __shared__ uint32_t s_tmp[NW];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(tile_counter, 1u);
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < NW; ++w) s_cnt[w][tid] = 0u;
    }
    __syncthreads();
    const uint32_t tile = s_tile;
    const size_t base = (size_t)tile * TILE + (size_t)wave * PER_WAVE;
    uint32_t k[RS_ROUNDS], v[RS_ROUNDS];
    uint16_t rk[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t i = base + (size_t)(r * 64 + lane);
        k[r] = (i < (size_t)n) ? kin[i] : 0xFFFFFFFFu;
        v[r] = (i < (size_t)n) ? (vin ? vin[i] : (uint32_t)i) : 0u;
    }
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t i = base + (size_t)(r * 64 + lane);
        const bool valid = i < (size_t)n;
        const uint32_t d = (k[r] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t old = 0;
        const int leader = valid ? (__ffsll((unsigned long long)peers) - 1) : lane;
        if (valid && lane == leader) {
            old = s_cnt[wave][d];
            s_cnt[wave][d] = old + (uint32_t)__popcll(peers);
        }
        old = __shfl(old, leader);
        rk[r] = (uint16_t)(old + (uint32_t)__popcll(peers & lt));
        // (LDS operations of one wave complete in order; the fence only keeps the compiler from reordering the rounds)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    uint32_t total = 0, prev = 0;
    if (tid < 256) {
        // thread d: the tile's count of digit d, the waves' offsets inside that run
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint32_t c = s_cnt[w][tid];
            s_cnt[w][tid] = total;
            total += c;
        }
        // look-back over the preceding tiles' counts of digit d: flags (bits 31:30) 0 not there yet, 1 count of that tile, 2 count of all tiles up to it
        const uint32_t VALUE = (1u << 30) - 1u;
        if (tile > 0) {
            __hip_atomic_store(&status[(size_t)tile * 256u + (size_t)tid], (1u << 30) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long p = (long long)tile - 1;
            while (true) {
                const uint32_t s = __hip_atomic_load(&status[(size_t)p * 256u + (size_t)tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t flag = s >> 30;
                if (flag == 0u) {
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                prev += s & VALUE;
                if (flag == 2u) break;
                --p;
            }
        }
        __hip_atomic_store(&status[(size_t)tile * 256u + (size_t)tid], (2u << 30) | (prev + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (the block-wide prefix sums run over the first 256 threads' values; the other threads contribute zeros)
    const uint32_t digit_base = rs_block_excl<NT>(tid < 256 ? hist[tid] : 0u, s_tmp, tid);  // keys with a smaller digit, anywhere
    const uint32_t lbase = rs_block_excl<NT>(total, s_tmp, tid);
    if (tid < 256) {
        s_gbase[tid] = digit_base + prev;
        s_lbase[tid] = lbase;
    }
    __syncthreads();
    const size_t tile_begin = (size_t)tile * TILE;
    const uint32_t tile_n = (uint32_t)(((size_t)n - tile_begin) < (size_t)TILE ? ((size_t)n - tile_begin) : (size_t)TILE);
    // keys: stage in digit order, write every digit's run to its place (consecutive lanes, consecutive addresses); then the values the same way
    uint32_t pos[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t i = base + (size_t)(r * 64 + lane);
        const uint32_t d = (k[r] >> shift) & 255u;
        pos[r] = s_lbase[d] + s_cnt[wave][d] + (uint32_t)rk[r];
        if (i < (size_t)n) s_stage[pos[r]] = k[r];
    }
    __syncthreads();
    size_t g[RS_ROUNDS];
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; ++j) {
        const uint32_t s = (uint32_t)(j * NT + tid);
        g[j] = 0;
        if (s < tile_n) {
            const uint32_t key = s_stage[s];
            const uint32_t d = (key >> shift) & 255u;
            g[j] = (size_t)s_gbase[d] + (size_t)(s - s_lbase[d]);
            kout[g[j]] = key;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t i = base + (size_t)(r * 64 + lane);
        if (i < (size_t)n) s_stage[pos[r]] = v[r];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; ++j) {
        const uint32_t s = (uint32_t)(j * NT + tid);
        if (s < tile_n) vout[g[j]] = s_stage[s];
    }
}

static inline unsigned rs_passes(unsigned bits) {
    unsigned p = (bits + 7u) / 8u;
    return p < 1u ? 1u : (p > 4u ? 4u : p);
}

// tiles of 8192 pairs (512 threads) for large inputs: a digit's run in a tile is then 128 bytes on average (whole cache lines) and half as many
// look-backs; 4096 (256 threads) below 2^20 pairs
static inline uint32_t rs_tile(uint32_t n) { return n >= (1u << 20) ? 8192u : 4096u; }

size_t ss_radix_sort_work_words(uint32_t n, unsigned bits) {
    const size_t tiles = ((size_t)n + 4096 - 1) / 4096;  // (sized for the small tile)
    const size_t np = rs_passes(bits);
    return np * 256 + 8 + np * tiles * 256 + 64;  // histograms, tile counters, status
}

int ss_radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], uint32_t n, unsigned bits, bool iota, uint32_t* work, bool work_is_zero, hipStream_t st) {
    if (n == 0) return 0;
    const unsigned np = rs_passes(bits);
    const uint32_t tile = rs_tile(n);
    const uint32_t tiles = (uint32_t)(((size_t)n + tile - 1) / tile);
    if (!work_is_zero) (void)hipMemsetAsync(work, 0, ss_radix_sort_work_words(n, bits) * 4, st);
    uint32_t* hist = work;
    uint32_t* counters = work + np * 256;
    uint32_t* status = counters + 8;
    uint32_t hgrid = (n + 256u * 32u - 1u) / (256u * 32u);
    if (hgrid > 2048u) hgrid = 2048u;
    if (n <= (1u << 16))  // (small inputs: one copy of the counters, nothing to zero and add up that is not needed)
        hipLaunchKernelGGL(k_rs_hist<1>, dim3(hgrid), dim3(256), 0, st, keys[0], n, (int)np, hist);
    else
        hipLaunchKernelGGL(k_rs_hist<8>, dim3(hgrid), dim3(256), 0, st, keys[0], n, (int)np, hist);
    int cur = 0;
    for (unsigned p = 0; p < np; ++p) {
        const uint32_t* vin = (p == 0 && iota) ? nullptr : vals[cur];
        if (tile == 8192u)
            hipLaunchKernelGGL(k_rs_pass<512>, dim3(tiles), dim3(512), 0, st, keys[cur], vin, keys[cur ^ 1], vals[cur ^ 1], n, (int)(8 * p), hist + p * 256, counters + p,
                               status + (size_t)p * tiles * 256u);
        else
            hipLaunchKernelGGL(k_rs_pass<256>, dim3(tiles), dim3(256), 0, st, keys[cur], vin, keys[cur ^ 1], vals[cur ^ 1], n, (int)(8 * p), hist + p * 256, counters + p,
                               status + (size_t)p * tiles * 256u);
        cur ^= 1;
    }
    return cur;
}

// ---- debug / test entry points (tests/test_gpu_prims.py): device pointers in, results in place -------------------------------------
struct SSDebugIdent {
    const uint32_t* p;
    __device__ uint32_t operator()(uint32_t i) const { return p[i]; }
};
struct SSDebugStore {
    uint32_t* p;
    __device__ void operator()(uint32_t i, uint32_t, uint32_t excl) const { p[i] = excl; }
};
extern "C" int ss_debug_exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total_dev, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    uint32_t* state = nullptr;
    const size_t words = ss_scan_state_words(n);
    if (hipMalloc(&state, words * 4) != hipSuccess) return 1;
    (void)hipMemsetAsync(state, 0, words * 4, st);
    ss_chained_scan<uint32_t, SSOpPlus>(SSDebugIdent{in}, SSDebugStore{out}, n, state, total_dev, SSMailSlot{}, st);
    const hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(state);
    return e == hipSuccess ? 0 : 2;
}
// keys0 / vals0 in; sorted pairs end in keys_out / vals_out whichever parity; iota != 0: values are the positions
extern "C" int ss_debug_radix_sort_pairs(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, uint32_t n, unsigned bits, int iota, int* result_buffer, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    uint32_t* work = nullptr;
    const size_t words = ss_radix_sort_work_words(n, bits);
    if (hipMalloc(&work, words * 4) != hipSuccess) return 1;
    uint32_t* keys[2] = {keys0, keys1};
    uint32_t* vals[2] = {vals0, vals1};
    const int r = ss_radix_sort_pairs(keys, vals, n, bits, iota != 0, work, false, st);
    const hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(work);
    if (result_buffer) *result_buffer = r;
    return e == hipSuccess ? 0 : 2;
}
