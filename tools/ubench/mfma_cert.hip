// mfma_cert.hip -- building block of the splat's certificate on the matrix pipe (round 6):
//     D[j][n] = s_j (1 - |x_n - p_j|^2)   for 32 list entries j (rows) and 32 grid points n (columns)
// as ONE v_mfma_f32_32x32x8_f16 over the K = 8 slots
//     A (entry j):  (s a)_hi, (s a)_lo, 2 s px, 2 s py | 2 s pz, -s, -s, 0          a = 1 - |p|^2
//     B (point n):  1,        1,        x,      y      | z,      X2_hi, X2_lo, 0     X2 = |x|^2
// followed by three VALU instructions per output: m = max(D, 0); m2 = m m; acc += m2 m2   (the bound 0.76 u^4 <= W / sigma, V folded into s).
// (1) checks the fragment layout against the host, (2) measures the issue cost of the tile beside its VALU part at 1 .. 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/ubench/mfma_cert.hip -o tools/ubench/mfma_cert
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ void k_layout(const _Float16* __restrict__ A /*32 x 8*/, const _Float16* __restrict__ B /*8 x 32*/, float* __restrict__ D /*32 x 32*/) {
    const int l = threadIdx.x;
    half4v a, b;
    for (int k = 0; k < 4; ++k) {
        a[k] = A[(l & 31) * 8 + 4 * (l >> 5) + k];
        b[k] = B[(4 * (l >> 5) + k) * 32 + (l & 31)];
    }
    float16v c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

__device__ __forceinline__ float term(float d, float acc) {
    float m, m2;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(m) : "v"(d));
    asm("v_mul_f32_e32 %0, %1, %1" : "=v"(m2) : "v"(m));
    asm("v_fmac_f32_e32 %0, %1, %1" : "+v"(acc) : "v"(m2));
    return acc;
}

// MODE 0: VALU part alone (16 outputs x {max, mul, fmac}); 1: MFMA 32x32x8 f16 alone; 2: both, the VALU part consuming the MFMA's outputs;
// 3: as 2 with two tiles in flight (the second MFMA issued before the first tile's outputs are consumed); 4: 32x32x16 f16 + VALU;
// 5: VALU part with v_max as "v_max_f32_e32 d, 0, s" ; 6: VALU part = {v_mul (D*D), v_fmac} only (what an unclamped bound would cost)
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_rate(const uint32_t* __restrict__ in, int iters, float* __restrict__ out) {
    const int l = threadIdx.x;
    half4v a, b;
    half8v a8, b8;
    for (int k = 0; k < 4; ++k) {
        a[k] = (_Float16)(0.01f * (float)((in[l] >> (4 * k)) & 15u));
        b[k] = (_Float16)(0.02f * (float)((in[64 + l] >> (4 * k)) & 15u));
    }
    for (int k = 0; k < 8; ++k) {
        a8[k] = a[k & 3];
        b8[k] = b[k & 3];
    }
    float acc = 0.0f, acc2 = 0.0f;
    float16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float16v c = z, c2 = z;
    for (int r = 0; r < 16; ++r) c[r] = 0.001f * (float)(l + r) - 0.03f;
    c2 = c;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 5 || MODE == 6) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float m, m2;
                if (MODE == 0) {
                    asm volatile("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(c[r]), "v"(0.0f));
                    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(m2) : "v"(m));
                } else if (MODE == 5) {
                    asm volatile("v_max_f32_e32 %0, 0, %1" : "=v"(m) : "v"(c[r]));
                    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(m2) : "v"(m));
                } else {
                    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(m2) : "v"(c[r]));
                }
                asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(acc) : "v"(m2));
            }
            asm volatile("" : "+v"(c));
        } else if (MODE == 1) {
            c = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, z, 0, 0, 0);
            asm volatile("" : "+v"(c));
            acc += c[0];
        } else if (MODE == 2) {
            c = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = term(c[r], acc);
        } else if (MODE == 3) {
            c2 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = term(c[r], acc);
            c = __builtin_amdgcn_mfma_f32_32x32x8f16(b, a, z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2 = term(c2[r], acc2);
        } else if (MODE == 4) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = term(c[r], acc);
        }
        a[0] += (_Float16)0.0001f;  // keeps the loop body from being hoisted
        a8[0] += (_Float16)0.0001f;
    }
    out[blockIdx.x * 64 + l] = acc + acc2 + c[3] + c2[5];
}

static float h2f(_Float16 h) { return (float)h; }

int main() {
    // ---- (1) layout: entries, points, the K = 8 slots, against the host in f64 on the SAME f16 operands
    _Float16 hA[32 * 8], hB[8 * 32];
    float hD[32 * 32];
    srand(7);
    auto rnd = []() { return (float)rand() / (float)RAND_MAX; };
    double worst_model = 0.0;
    float px[32], py[32], pz[32], sj[32], X[32], Y[32], Z[32];
    for (int j = 0; j < 32; ++j) {
        px[j] = 2.0f * rnd() - 1.0f, py[j] = 2.0f * rnd() - 1.0f, pz[j] = 2.0f * rnd() - 1.0f;
        sj[j] = 0.6f + 0.2f * rnd();
        const float a = 1.0f - (px[j] * px[j] + py[j] * py[j] + pz[j] * pz[j]);
        const float sa = sj[j] * a;
        const _Float16 hi = (_Float16)sa;
        hA[j * 8 + 0] = hi;
        hA[j * 8 + 1] = (_Float16)(sa - (float)hi);
        hA[j * 8 + 2] = (_Float16)(2.0f * sj[j] * px[j]);
        hA[j * 8 + 3] = (_Float16)(2.0f * sj[j] * py[j]);
        hA[j * 8 + 4] = (_Float16)(2.0f * sj[j] * pz[j]);
        hA[j * 8 + 5] = (_Float16)(-sj[j]);
        hA[j * 8 + 6] = (_Float16)(-sj[j]);
        hA[j * 8 + 7] = (_Float16)0.0f;
    }
    for (int n = 0; n < 32; ++n) {
        X[n] = 0.875f * rnd() - 0.4375f, Y[n] = 0.875f * rnd() - 0.4375f, Z[n] = 0.875f * rnd() - 0.4375f;
        const float x2 = X[n] * X[n] + Y[n] * Y[n] + Z[n] * Z[n];
        const _Float16 hi = (_Float16)x2;
        hB[0 * 32 + n] = (_Float16)1.0f;
        hB[1 * 32 + n] = (_Float16)1.0f;
        hB[2 * 32 + n] = (_Float16)X[n];
        hB[3 * 32 + n] = (_Float16)Y[n];
        hB[4 * 32 + n] = (_Float16)Z[n];
        hB[5 * 32 + n] = hi;
        hB[6 * 32 + n] = (_Float16)(x2 - (float)hi);
        hB[7 * 32 + n] = (_Float16)0.0f;
    }
    _Float16 *dA, *dB;
    float* dD;
    hipMalloc(&dA, sizeof(hA));
    hipMalloc(&dB, sizeof(hB));
    hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int j = 0; j < 32; ++j)
        for (int n = 0; n < 32; ++n) {
            double ref = 0.0;
            for (int k = 0; k < 8; ++k) ref += (double)h2f(hA[j * 8 + k]) * (double)h2f(hB[k * 32 + n]);
            worst = fmax(worst, fabs(ref - (double)hD[j * 32 + n]));
            const double dx = X[n] - px[j], dy = Y[n] - py[j], dz = Z[n] - pz[j];
            const double truth = (double)sj[j] * (1.0 - (dx * dx + dy * dy + dz * dz));
            worst_model = fmax(worst_model, fabs(truth - (double)hD[j * 32 + n]));
        }
    printf("layout: max |D - sum_k A B| (same f16 operands, f64 host) = %.3g ; max |D - s (1 - |x - p|^2)| = %.3g\n", worst, worst_model);

    // ---- (2) rates
    uint32_t h[128];
    for (int i = 0; i < 128; ++i) h[i] = (uint32_t)rand();
    uint32_t* din;
    float* dout;
    hipMalloc(&din, sizeof(h));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    const char* names[] = {"VALU 16 x {max, mul, fmac}", "MFMA 32x32x8 f16 alone", "MFMA + VALU (dependent)", "MFMA + VALU, two tiles in flight (per tile)", "MFMA 32x32x16 f16 + VALU",
                           "VALU 16 x {max_e32 0, mul, fmac}", "VALU 16 x {mul, fmac}"};
    for (int wps : {1, 2, 4, 6, 8}) {
        const int blocks = 256 * 4 * wps;
        hipMalloc(&dout, (size_t)blocks * 64 * 4);
        for (int mode = 0; mode < 7; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(64), 0, 0, din, iters, dout); break;
                    case 1: hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64), 0, 0, din, iters, dout); break;
                    case 2: hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(64), 0, 0, din, iters, dout); break;
                    case 3: hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(64), 0, 0, din, iters, dout); break;
                    case 4: hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(64), 0, 0, din, iters, dout); break;
                    case 5: hipLaunchKernelGGL(k_rate<5>, dim3(blocks), dim3(64), 0, 0, din, iters, dout); break;
                    case 6: hipLaunchKernelGGL(k_rate<6>, dim3(blocks), dim3(64), 0, 0, din, iters, dout); break;
                }
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            if (hipGetLastError() != hipSuccess) printf("launch error in mode %d\n", mode);
            const double tiles_per_simd = (double)wps * iters * (mode == 3 ? 2.0 : 1.0);
            printf("%d waves/SIMD  %-45s %.3f ms  %.1f cycles per 32 x 32 tile per SIMD at 2.4 GHz\n", wps, names[mode], ms, ms * 1e-3 * 2.4e9 / tiles_per_simd);
        }
        hipFree(dout);
    }
    return 0;
}
