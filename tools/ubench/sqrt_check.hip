// Exhaustive check of v_sqrt_f32 on gfx950 against the correctly rounded square root (residual test against both
// neighbours), over every positive normal float.  Prints how often the raw instruction is exact / one ulp high / one ulp low.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float sqrt_rn(float x, float s) {
    const float sm = __int_as_float(__float_as_int(s) - 1);
    const float sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x);
    const float rp = __builtin_fmaf(-sp, s, x);
    float r = (rm <= 0.0f) ? sm : s;
    r = (rp > 0.0f) ? sp : r;
    return r;
}
// a branch-free variant: +1 / -1 ulp from the sign bits of the two residuals with integer adds and shifts instead of
// v_cmp / v_cndmask; s is kept >= 2^-100 so that s - 1 ulp stays a normal float.  Exact (see below) but SLOWER in the splat
// (k_splat_accumulate 25.1 instead of 24.5 ms), so the kernels keep the compare/select form
__device__ __forceinline__ float sqrt_rn_int(float x) {
    const float s = fmaxf(__builtin_amdgcn_sqrtf(x), 7.888609052210118e-31f);
    const int sb = __float_as_int(s);
    const float sm = __int_as_float(sb - 1), sp = __int_as_float(sb + 1);
    const int um = __float_as_int(__builtin_fmaf(-sm, s, x));   // x - sm*s  (<= 0: one ulp down)
    const int up = __float_as_int(__builtin_fmaf(sp, s, -x));   // sp*s - x  (<  0: one ulp up)
    return __int_as_float(sb + (int)((unsigned)up >> 31) + ((um - 1) >> 31));
}
__global__ void k2(unsigned long long* out) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    // x >= 2^-97 (biased exponent 30): the residuals x - s'*s are normal floats there.  Below, they underflow and only
    // "tiny and finite" is required (the callers' W does not depend on r for d^2 < (2^-14 h)^2 >= 2^-88, h > 1e-9).
    unsigned long long bad = 0, bad_low = 0;
    for (uint32_t e = 1; e <= 254; ++e) {
        const float x = __uint_as_float((e << 23) | m);
        const float s = __builtin_amdgcn_sqrtf(x);
        if (__float_as_int(sqrt_rn(x, s)) != __float_as_int(sqrt_rn_int(x))) {
            if (e >= 30) ++bad; else ++bad_low;
        }
    }
    atomicAdd(&out[5], bad);
    atomicAdd(&out[7], bad_low);
    if (m < 64) {  // zero, denormals and the smallest normals: the result only has to be tiny and finite (callers: W == W(0) there)
        const float xs[4] = {0.0f, __uint_as_float(m + 1u), __uint_as_float(0x00800000u + m), __uint_as_float((1u << 22) + m)};
        for (int i = 0; i < 4; ++i) {
            const float r = sqrt_rn_int(xs[i]);
            if (!(r >= 0.0f && r < 1.0e-15f)) atomicAdd(&out[6], 1ull);
        }
    }
}
__global__ void k(unsigned long long* out, uint32_t lo_exp, uint32_t hi_exp) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;  // 23-bit significand
    unsigned long long exact = 0, high = 0, low = 0, other = 0, vs_lib = 0;
    for (uint32_t e = lo_exp; e <= hi_exp; ++e) {
        const float x = __uint_as_float((e << 23) | m);
        const float s = __builtin_amdgcn_sqrtf(x);
        const float r = sqrt_rn(x, s);
        const int d = __float_as_int(s) - __float_as_int(r);
        if (d == 0) ++exact; else if (d == 1) ++high; else if (d == -1) ++low; else ++other;
        if (__float_as_int(r) != __float_as_int(__fsqrt_rn(x))) ++vs_lib;
    }
    atomicAdd(&out[0], exact); atomicAdd(&out[1], high); atomicAdd(&out[2], low); atomicAdd(&out[3], other); atomicAdd(&out[4], vs_lib);
}
// the AMDGPU backend's other correctly rounded form: Goldschmidt step on v_rsq_f32 + residual correction (ss_sqrt_rn_rsq)
__device__ __forceinline__ float sqrt_rn_rsq(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    float h = 0.5f * y;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    h = __builtin_fmaf(h, e, h);
    s = __builtin_fmaf(s, e, s);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__global__ void k3(unsigned long long* out) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad_hi = 0, bad_mid = 0, bad_lo = 0;
    for (uint32_t e = 1; e <= 254; ++e) {
        const float x = __uint_as_float((e << 23) | m);
        const float ref = sqrt_rn(x, __builtin_amdgcn_sqrtf(x));
        if (__float_as_int(ref) != __float_as_int(sqrt_rn_rsq(x))) {
            if (e >= 230) ++bad_hi; else if (e >= 30) ++bad_mid; else ++bad_lo;
        }
    }
    atomicAdd(&out[0], bad_hi);
    atomicAdd(&out[1], bad_mid);
    atomicAdd(&out[2], bad_lo);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    k<<<(1u << 23) / 256, 256>>>(d, 1, 254);
    k2<<<(1u << 23) / 256, 256>>>(d);
    unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("v_sqrt_f32 over all positive normals: exact %llu, one ulp high %llu, one ulp low %llu, other %llu; fix-up vs __fsqrt_rn mismatches %llu\n", h[0], h[1], h[2], h[3], h[4]);
    printf("branch-free fix-up vs compare/select fix-up: %llu mismatches for x >= 2^-97, %llu below (residual underflow); zero/denormal inputs outside [0, 1e-15): %llu\n", h[5], h[7], h[6]);
    hipMemset(d, 0, 64);
    k3<<<(1u << 23) / 256, 256>>>(d);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("rsq/Goldschmidt form vs v_sqrt fix-up form: mismatches %llu for x >= 2^103, %llu for 2^-97 <= x < 2^103, %llu below 2^-97\n", h[0], h[1], h[2]);
    return 0;
}
