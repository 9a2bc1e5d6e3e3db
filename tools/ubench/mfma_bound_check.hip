// mfma_bound_check.hip -- device check of the building block of the splat's lower-bound pass (ss_kernels.hip, splat_bound_list_mfma):
// squared distances of 64 grid points (lane = point) to 64 tile entries (lane = entry) on the matrix pipe,
//     d2[e][p] = fma(s_e, 1, fma(z_e, -2 z_p, fma(y_e, -2 y_p, fma(x_e, -2 x_p, |p|^2)))),   s_e = |e|^2 (+ slack),
// four entries per chain of four v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 x 4, K = 1): lane l supplies B = its own point, every
// block takes A from block `abid` (cbsz = 4: lanes 4 abid .. 4 abid + 3 hold the four entries), and lane l receives in register r
// the pair (entry 4 abid + r, point l).  Checks (1) that layout and the broadcast modifiers against a scalar fma chain, bit for
// bit, and (2) the issue cost of the chain beside VALU work.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/mfma_bound_check.hip -o tools/ubench/mfma_bound_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float float4v __attribute__((ext_vector_type(4)));

template <int G>
__device__ __forceinline__ float4v chain(const float4 e, float bx, float by, float bz, float p2) {
    float4v c = {p2, p2, p2, p2};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(e.x, bx, c, 4, G, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(e.y, by, c, 4, G, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(e.z, bz, c, 4, G, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(e.w, 1.0f, c, 4, G, 0);
    return c;
}

__global__ void k_layout(const float4* __restrict__ entries, const float4* __restrict__ points, float* __restrict__ d2) {
    const int l = threadIdx.x;
    const float4 e = entries[l];  // (x, y, z, |e|^2)
    const float4 p = points[l];   // (x, y, z, |p|^2)
    const float bx = -2.0f * p.x, by = -2.0f * p.y, bz = -2.0f * p.z;
#define GROUP(G)                                                    \
    {                                                               \
        const float4v c = chain<G>(e, bx, by, bz, p.w);             \
        for (int r = 0; r < 4; ++r) d2[(4 * G + r) * 64 + l] = c[r]; \
    }
    GROUP(0) GROUP(1) GROUP(2) GROUP(3) GROUP(4) GROUP(5) GROUP(6) GROUP(7) GROUP(8) GROUP(9) GROUP(10) GROUP(11) GROUP(12) GROUP(13) GROUP(14) GROUP(15)
#undef GROUP
}

// the per-pair arithmetic that follows the distances in the lower-bound pass: sqrt, two clamped ops, two mul, fma
__device__ __forceinline__ float bound_term(float d2, float vol, float acc) {
    const float q = __builtin_amdgcn_sqrtf(d2);
    float v, t;
    asm("v_sub_f32_e64 %0, 1.0, %1 clamp" : "=v"(v) : "v"(q));
    asm("v_add_f32_e64 %0, %1, %1 clamp" : "=v"(t) : "v"(v));
    return __builtin_fmaf((v * v) * t, vol, acc);
}

// MODE 0: distances by VALU (3 sub, mul, 2 fma) + term; MODE 1: distances by the MFMA chain + term; MODE 2: the MFMA chains alone
template <int MODE>
__global__ __launch_bounds__(64) void k_rate(const float4* __restrict__ entries, const float4* __restrict__ points, int iters, float* __restrict__ out) {
    const int l = threadIdx.x;
    float4 e = entries[l];
    const float4 p = points[l];
    const float bx = -2.0f * p.x, by = -2.0f * p.y, bz = -2.0f * p.z;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float ex = __shfl(e.x, g), ey = __shfl(e.y, g), ez = __shfl(e.z, g), vol = __shfl(e.w, g);  // (stand-in for the LDS broadcast)
                const float dx = ex - p.x, dy = ey - p.y, dz = ez - p.z;
                acc = bound_term(__builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy)), vol, acc);
            }
        } else {
#define GROUP(G)                                                                     \
    {                                                                                \
        const float4v c = chain<G>(e, bx, by, bz, p.w);                              \
        if (MODE == 1) {                                                             \
            for (int r = 0; r < 4; ++r) acc = bound_term(c[r], 0.125f, acc);         \
        } else                                                                       \
            acc += c[0] + c[3];                                                      \
    }
            GROUP(0) GROUP(1) GROUP(2) GROUP(3)
#undef GROUP
        }
        e.x += 1.0e-7f;  // keeps the loop body from being hoisted
    }
    out[blockIdx.x * 64 + l] = acc;
}

static float frand() { return (float)rand() / (float)RAND_MAX; }

int main() {
    float4 he[64], hp[64];
    srand(7);
    for (int i = 0; i < 64; ++i) {
        he[i].x = 1.8f * frand() - 0.9f;
        he[i].y = 1.8f * frand() - 0.9f;
        he[i].z = 1.8f * frand() - 0.9f;
        he[i].w = (he[i].x * he[i].x + he[i].y * he[i].y) + he[i].z * he[i].z;
        hp[i].x = 0.125f * (float)((i >> 4) & 3) - 0.1875f;
        hp[i].y = 0.125f * (float)((i >> 2) & 3) - 0.1875f;
        hp[i].z = 0.125f * (float)(i & 3) - 0.1875f;
        hp[i].w = (hp[i].x * hp[i].x + hp[i].y * hp[i].y) + hp[i].z * hp[i].z;
    }
    float4 *de, *dp;
    float *dd, *dout;
    hipMalloc(&de, sizeof(he));
    hipMalloc(&dp, sizeof(hp));
    hipMalloc(&dd, 64 * 64 * 4);
    hipMalloc(&dout, 4096 * 64 * 4 * 8);
    hipMemcpy(de, he, sizeof(he), hipMemcpyHostToDevice);
    hipMemcpy(dp, hp, sizeof(hp), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, de, dp, dd);
    static float hd[64 * 64];
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    double worst = 0.0;
    for (int e = 0; e < 64; ++e)
        for (int p = 0; p < 64; ++p) {
            float c = hp[p].w;
            c = fmaf(he[e].x, -2.0f * hp[p].x, c);
            c = fmaf(he[e].y, -2.0f * hp[p].y, c);
            c = fmaf(he[e].z, -2.0f * hp[p].z, c);
            c = fmaf(he[e].w, 1.0f, c);
            const float got = hd[e * 64 + p];
            if (memcmp(&c, &got, 4) != 0) {
                if (bad < 5) printf("MISMATCH entry %d point %d: expected %.9g got %.9g\n", e, p, c, got);
                ++bad;
            }
            const double dx = (double)he[e].x - hp[p].x, dy = (double)he[e].y - hp[p].y, dz = (double)he[e].z - hp[p].z;
            const double err = fabs((double)got - (dx * dx + dy * dy + dz * dz));
            if (err > worst) worst = err;
        }
    printf("layout: %d mismatches of 4096 against the scalar fma chain; largest |d2 - exact| = %.3g (|e| <= 1.56, |p| <= 0.33)\n", bad, worst);

    // rates: grid fills the chip with 6 waves per SIMD
    const int blocks = 256 * 4 * 6, iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a, 0);
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(64), 0, 0, de, dp, iters, dout);
            if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64), 0, 0, de, dp, iters, dout);
            if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(64), 0, 0, de, dp, iters, dout);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        const double pairs = (double)blocks * 64.0 * iters * 16.0;
        printf("mode %d (%s): %.3f ms, %.1f Gpairs/s\n", mode, mode == 0 ? "VALU distances + term" : mode == 1 ? "MFMA distances + term" : "MFMA chains alone", ms,
               pairs / ms * 1e-6);
    }
    return bad ? 1 : 0;
}
