// inst_rates.hip -- cycles per wave64 instruction per SIMD for the instruction forms the splat's inner loops are built from
// (gfx950, 1 and 6 waves per SIMD, 8 independent accumulators per wave).  Generated list: tools/ubench/inst_rates.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/inst_rates.hip -o tools/ubench/inst_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define OP8(T) OP(T, a0, b0) OP(T, a1, b1) OP(T, a2, b2) OP(T, a3, b3) OP(T, a4, b0) OP(T, a5, b1) OP(T, a6, b2) OP(T, a7, b3)
#define OP(T, A, B) asm volatile(T : "+v"(A), "+v"(B) : "v"(p), "v"(q), "v"(w), "v"(pp) : "vcc", "s8", "s9", "s10");
template <int MODE>
__global__ __launch_bounds__(64) void k(const uint32_t* __restrict__ in, float* __restrict__ out, int iters) {
    const int l = threadIdx.x;
    const uint32_t w = in[l];
    const float p = __uint_as_float(in[64 + l]), q = __uint_as_float(in[128 + l]);
    f2 pp = {p, q};
    float a0 = 0.f, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    f2 b0 = {1.f, 2.f}, b1 = {2.f, 3.f}, b2 = {3.f, 4.f}, b3 = {4.f, 5.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) { OP8("v_mul_f32 %0, %2, %0") }
            else if (MODE == 1) { OP8("v_sub_f32 %0, %2, %0") }
            else if (MODE == 2) { OP8("v_fmac_f32 %0, %2, %3") }
            else if (MODE == 3) { OP8("v_add_f32 %0, %2, %0") }
            else if (MODE == 4) { OP8("v_fma_f32 %0, %2, %3, %0") }
            else if (MODE == 5) { OP8("v_fma_f32 %0, %2, %2, %0") }
            else if (MODE == 6) { OP8("v_fma_f32 %0, -%2, %2, %0 clamp") }
            else if (MODE == 7) { OP8("v_fmamk_f32 %0, %0, 0x3f4906cd, %2") }
            else if (MODE == 8) { OP8("v_fmaak_f32 %0, %0, %2, 0x3f4906cd") }
            else if (MODE == 9) { OP8("v_sub_f32_e64 %0, 1.0, %0 clamp") }
            else if (MODE == 10) { OP8("v_fma_mix_f32 %0, %4, %2, %0 op_sel_hi:[1,0,0]") }
            else if (MODE == 11) { OP8("v_fma_mix_f32 %0, %4, 1.0, %0 op_sel_hi:[1,0,0]") }
            else if (MODE == 12) { OP8("v_dot2_f32_f16 %0, %4, %4, %0") }
            else if (MODE == 13) { OP8("v_dot2c_f32_f16 %0, %4, %4") }
            else if (MODE == 14) { OP8("v_pk_fma_f16 %0, %4, %4, %0") }
            else if (MODE == 15) { OP8("v_pk_mul_f32 %1, %5, %1") }
            else if (MODE == 16) { OP8("v_mul_f32 %0, s4, %0") }
            else if (MODE == 17) { OP8("v_fma_f32 %0, %2, %3, s4") }
            else if (MODE == 18) { OP8("v_cndmask_b32 %0, %0, %2, vcc") }
            else if (MODE == 19) { OP8("v_exp_f32 %0, %0") }
            else if (MODE == 20) { OP8("v_rcp_f32 %0, %0") }
            else if (MODE == 21) { OP8("v_sqrt_f32 %0, %0") }
            else if (MODE == 22) { OP8("v_cndmask_b32_e64 %0, %0, %2, s[6:7]") }
            else if (MODE == 23) { OP8("v_cmp_lt_f32 vcc, %0, %2") }
            else if (MODE == 24) { OP8("v_cmp_lt_f32_e64 s[8:9], %0, %2") }
            else if (MODE == 25) { OP8("v_cmp_lt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %3, vcc") }
            else if (MODE == 26) { OP8("v_and_b32 %0, %2, %0") }
            else if (MODE == 27) { OP8("v_or_b32 %0, %2, %0") }
            else if (MODE == 28) { OP8("v_lshlrev_b32 %0, 1, %0") }
            else if (MODE == 29) { OP8("v_add_u32 %0, %2, %0") }
            else if (MODE == 30) { OP8("v_bfe_u32 %0, %0, 3, 5") }
            else if (MODE == 31) { OP8("v_lshl_add_u32 %0, %0, 1, %2") }
            else if (MODE == 32) { OP8("v_max_f32 %0, %2, %0") }
            else if (MODE == 33) { OP8("v_mbcnt_lo_u32_b32 %0, s6, %0") }
            else if (MODE == 34) { OP8("v_cvt_f32_f16 %0, %0") }
            else if (MODE == 35) { OP8("v_cvt_f16_f32 %0, %0") }
            else if (MODE == 36) { OP8("v_mov_b32 %0, %2") }
            else if (MODE == 37) { OP8("v_mul_f32 %0, %2, %2") }
            else if (MODE == 38) { OP8("v_readlane_b32 s10, %0, 3") }
            else if (MODE == 39) { OP8("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") }
        }
    }
    out[blockIdx.x * 64 + l] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0.x + b1.x + b2.y + b3.y;
}
int main() {
    uint32_t h[192];
    for (int i = 0; i < 64; ++i) h[i] = 0x38003800u;                 // f16 0.5 pairs
    for (int i = 64; i < 128; ++i) h[i] = 0x3f7fff00u;               // f32 just below 1
    for (int i = 128; i < 192; ++i) h[i] = 0x3f000000u;              // f32 0.5
    uint32_t* din;
    float* dout;
    hipMalloc(&din, sizeof(h));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 3000;
    const char* names[] = {"v_mul_f32 (VOP2)", "v_sub_f32 (VOP2)", "v_fmac_f32 (VOP2)", "v_add_f32 (VOP2)", "v_fma_f32 d,a,b,d", "v_fma_f32 d,a,a,d", "v_fma_f32 d,-a,a,d clamp", "v_fmamk_f32 d,d,K,b", "v_fmaak_f32 d,d,a,K", "v_sub_f32_e64 clamp", "v_fma_mix_f32 (1 f16 src)", "v_fma_mix_f32 d,h,1.0,f", "v_dot2_f32_f16", "v_dot2c_f32_f16 (VOP2)", "v_pk_fma_f16", "v_pk_mul_f32", "v_mul_f32 sgpr operand", "v_fma_f32 sgpr addend", "v_cndmask_b32", "v_exp_f32 (trans)", "v_rcp_f32 (trans)", "v_sqrt_f32 (trans)", "v_cndmask_b32_e64 sgpr mask", "v_cmp_lt_f32 vcc", "v_cmp_lt_f32_e64 s[8:9]", "v_cmp + v_cndmask (per pair)", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_add_u32", "v_bfe_u32", "v_lshl_add_u32", "v_max_f32", "v_mbcnt_lo_u32_b32", "v_cvt_f32_f16", "v_cvt_f16_f32", "v_mov_b32", "v_mul_f32 d,a,a (same reg)", "v_readlane_b32 s,v,3", "v_mov_b32_dpp row_shr:1"};
    for (int wps = 1; wps <= 6; wps += 5) {
        const int blocks = 256 * 4 * wps;
        hipMalloc(&dout, (size_t)blocks * 64 * 4);
        for (int mode = 0; mode < 40; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a, 0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 9: hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 10: hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 11: hipLaunchKernelGGL(k<11>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 12: hipLaunchKernelGGL(k<12>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 13: hipLaunchKernelGGL(k<13>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 14: hipLaunchKernelGGL(k<14>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 15: hipLaunchKernelGGL(k<15>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 16: hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 17: hipLaunchKernelGGL(k<17>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 18: hipLaunchKernelGGL(k<18>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 19: hipLaunchKernelGGL(k<19>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 20: hipLaunchKernelGGL(k<20>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 21: hipLaunchKernelGGL(k<21>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 22: hipLaunchKernelGGL(k<22>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 23: hipLaunchKernelGGL(k<23>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 24: hipLaunchKernelGGL(k<24>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 25: hipLaunchKernelGGL(k<25>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 26: hipLaunchKernelGGL(k<26>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 27: hipLaunchKernelGGL(k<27>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 28: hipLaunchKernelGGL(k<28>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 29: hipLaunchKernelGGL(k<29>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 30: hipLaunchKernelGGL(k<30>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 31: hipLaunchKernelGGL(k<31>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 32: hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 33: hipLaunchKernelGGL(k<33>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 34: hipLaunchKernelGGL(k<34>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 35: hipLaunchKernelGGL(k<35>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 36: hipLaunchKernelGGL(k<36>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 37: hipLaunchKernelGGL(k<37>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 38: hipLaunchKernelGGL(k<38>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                    case 39: hipLaunchKernelGGL(k<39>, dim3(blocks), dim3(64), 0, 0, din, dout, iters); break;
                }
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            if (hipGetLastError() != hipSuccess) printf("launch error in mode %d\n", mode);
            const double insts_per_simd = (double)wps * iters * 64.0;
            printf("%d waves/SIMD  %-30s %.3f ms  %.2f cycles per instruction per SIMD at 2.4 GHz\n", wps, names[mode], ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
        }
        hipFree(dout);
    }
    return 0;
}
