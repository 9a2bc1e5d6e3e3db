// mix_rate.hip -- issue rate of v_fma_mix_f32 (f16 operand converted on the fly) against v_fma_f32 on gfx950, 6 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mix_rate.hip -o tools/ubench/mix_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(64) void k(const uint32_t* __restrict__ in, float* __restrict__ out, int iters) {
    const int l = threadIdx.x;
    const uint32_t w = in[l];
    const float p = __uint_as_float(in[64 + l]);
    float a0 = 0.f, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {  // plain fma, 8 independent chains
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a4) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a5) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a6) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a7) : "v"(p), "v"(p));
            } else if (MODE == 1) {  // fma_mix with one f16 source
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a0) : "v"(w), "v"(p));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a1) : "v"(w), "v"(p));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a2) : "v"(w), "v"(p));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a3) : "v"(w), "v"(p));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a4) : "v"(w), "v"(p));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a5) : "v"(w), "v"(p));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a6) : "v"(w), "v"(p));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a7) : "v"(w), "v"(p));
            } else if (MODE == 2) {  // one dependent chain of plain fma (latency of back-to-back dependent issue)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(p), "v"(p));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(p), "v"(p));
            } else if (MODE == 3) {  // v_mul_f32 e32 (VOP2), 8 independent
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a0) : "v"(p));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a1) : "v"(p));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a2) : "v"(p));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a3) : "v"(p));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a4) : "v"(p));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a5) : "v"(p));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a6) : "v"(p));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a7) : "v"(p));
            } else if (MODE == 4) {  // one scalar add between vector ops: s_add_u32 interleaved 1:1
                asm volatile("v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1" : "+v"(a0) : "v"(p), "v"(p) : "s20");
                asm volatile("v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1" : "+v"(a1) : "v"(p), "v"(p) : "s20");
                asm volatile("v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1" : "+v"(a2) : "v"(p), "v"(p) : "s20");
                asm volatile("v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1" : "+v"(a3) : "v"(p), "v"(p) : "s20");
                asm volatile("v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1" : "+v"(a4) : "v"(p), "v"(p) : "s20");
                asm volatile("v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1" : "+v"(a5) : "v"(p), "v"(p) : "s20");
                asm volatile("v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1" : "+v"(a6) : "v"(p), "v"(p) : "s20");
                asm volatile("v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1" : "+v"(a7) : "v"(p), "v"(p) : "s20");
            } else {  // scalar only: 8 s_add
                asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" ::: "s20", "s21", "s22", "s23");
            }
        }
    }
    out[blockIdx.x * 64 + l] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    uint32_t h[128];
    for (int i = 0; i < 128; ++i) h[i] = 0x3c003c00u;  // f16 1.0 pairs / some f32
    uint32_t* din;
    float* dout;
    hipMalloc(&din, sizeof(h));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 4000;
    const char* names[6] = {"v_fma_f32 x8 independent", "v_fma_mix_f32 x8 independent", "v_fma_f32 x8 dependent", "v_mul_f32 (VOP2) x8 independent", "v_fma_f32 + s_add_u32 1:1", "s_add_u32 x8"};
    for (int wps = 1; wps <= 6; wps += 5) {
        const int blocks = 256 * 4 * wps;
        hipMalloc(&dout, (size_t)blocks * 64 * 4);
        for (int mode = 0; mode < 6; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a, 0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    default: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                }
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            const double insts_per_simd = (double)wps * iters * 64.0;  // instructions of the measured kind per SIMD
            printf("%d waves/SIMD  %-34s %.3f ms  -> %.2f cycles per instruction per SIMD at 2.4 GHz\n", wps, names[mode], ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
        }
        hipFree(dout);
    }
    return 0;
}
