// Micro-benchmark: issue cost (cycles per wave64 instruction, many waves resident) of the VALU ops used in
// the splat inner loop.  Each kernel runs a long unrolled chain of ONE independent-ish op per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP 256
#define ITER 200

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int n) {
    float x0 = a + threadIdx.x, x1 = b + threadIdx.x, x2 = a * 0.5f, x3 = b * 0.25f;
    float y0 = 1.f, y1 = 2.f, y2 = 3.f, y3 = 4.f;
    int i0 = threadIdx.x, i1 = threadIdx.x * 3;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 4; ++r) {
            if (OP == 0) { asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0)); }
            if (OP == 1) { asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0)); }
            if (OP == 2) { asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1)); }
            if (OP == 3) { asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            if (OP == 4) { asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_gt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1) : "vcc"); }
            if (OP == 5) { asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(i1)); }
            if (OP == 6) { asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "s20", "s21", "s22", "s23"); }
            if (OP == 7) { asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(*(double*)&x0), "+v"(*(double*)&x2) : "v"(*(double*)&y0)); }
            if (OP == 8) { asm volatile("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(a)); }
            if (OP == 9) { asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            if (OP == 10) { asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cmp_gt_f32 vcc, %1, %4\n v_cmp_gt_f32 vcc, %2, %4\n v_cmp_gt_f32 vcc, %3, %4" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y0) : "vcc"); }
            if (OP == 11) { asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0) : "vcc"); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3 + i0 + i1;
}

template <int OP>
double run(const char* name, float* d_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, 2.0f, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, 2.0f, ITER);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions issued per SIMD: 8 waves * ITER * REP ; SIMD count = 256 CUs * 4
    const double insts_per_simd = 8.0 * ITER * REP;
    const double ns_per_inst = ms * 1e6 / insts_per_simd;
    printf("%-28s %8.3f ms   %6.3f ns / wave-instr / SIMD  (= %5.2f cycles @2.4GHz, %5.2f @2.1GHz)\n", name, ms, ns_per_inst, ns_per_inst * 2.4, ns_per_inst * 2.1);
    return ns_per_inst;
}

int main() {
    float* d_out; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_add_f32", d_out); run<1>("v_mul_f32", d_out); run<2>("v_fma_f32", d_out); run<8>("v_sub_f32 (sgpr src)", d_out);
    run<3>("v_sqrt_f32", d_out); run<9>("v_rcp_f32", d_out);
    run<10>("v_cmp_gt_f32", d_out); run<11>("v_cndmask_b32", d_out); run<4>("v_cmp + v_cndmask (pairs)", d_out);
    run<5>("v_add_u32", d_out); run<6>("v_readlane_b32", d_out); run<7>("v_pk_mul_f32", d_out);
    return 0;
}
