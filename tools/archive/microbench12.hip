// Round-2 calibration, part 9: are there 3-instruction forms of the two divisions of quantize_row_q8_0 / q8_1 (d = amax / 127,
// id = 1 / d; biogpt.cpp -> ggml quantize_row_q8_*_reference) that give the IEEE quotient bit for bit ?  Exhaustive over every
// non-negative float amax (2^31 patterns):
//    d'  = fma(fma(-127, q, a), C, q)  with q = a * C, C = RN(1 / 127)
//    id' = fma(fma(-d, r, 1), r, r)    with r = v_rcp_f32(d)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__device__ __forceinline__ float fast_div127(float a) {
    const float C = 1.0f / 127.0f;
    const float q = __fmul_rn(a, C);
    return __fmaf_rn(__fmaf_rn(-127.0f, q, a), C, q);
}
__device__ __forceinline__ float fast_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return __fmaf_rn(__fmaf_rn(-d, r, 1.0f), r, r);
}

__global__ void check(unsigned long long *stats, uint32_t *first_bad) {
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long bad_d = 0, bad_id = 0, lo_d = ~0ull, hi_d = 0, lo_i = ~0ull, hi_i = 0;
    for (uint64_t b = blockIdx.x * blockDim.x + threadIdx.x; b < 0x7F800000ull; b += stride) {   // every finite non-negative float
        const float a = __uint_as_float((uint32_t)b);
        const float d = a / 127.0f;
        const float df = fast_div127(a);
        if (__float_as_uint(d) != __float_as_uint(df)) { bad_d++; if (b < lo_d) lo_d = b; if (b > hi_d) hi_d = b; }
        if (d != 0.0f) {
            const float id = 1.0f / d;
            const float idf = fast_rcp(d);
            if (__float_as_uint(id) != __float_as_uint(idf)) { bad_id++; if (b < lo_i) lo_i = b; if (b > hi_i) hi_i = b; }
        }
    }
    atomicAdd(&stats[0], bad_d); atomicAdd(&stats[1], bad_id);
    atomicMin(&stats[2], lo_d); atomicMax(&stats[3], hi_d); atomicMin(&stats[4], lo_i); atomicMax(&stats[5], hi_i);
}

// the same, restricted to [lo, hi): counts only
__global__ void check_range(unsigned long long *stats, uint32_t lo, uint32_t hi) {
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long bad_d = 0, bad_id = 0;
    for (uint64_t b = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += stride) {
        const float a = __uint_as_float((uint32_t)b);
        const float d = a / 127.0f;
        if (__float_as_uint(d) != __float_as_uint(fast_div127(a))) bad_d++;
        if (d != 0.0f && __float_as_uint(1.0f / d) != __float_as_uint(fast_rcp(d))) bad_id++;
    }
    atomicAdd(&stats[0], bad_d); atomicAdd(&stats[1], bad_id);
}

int main() {
    unsigned long long *st, h[6];
    CK(hipMalloc(&st, 64));
    const unsigned long long init[6] = {0, 0, ~0ull, 0, ~0ull, 0};
    CK(hipMemcpy(st, init, 48, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, st, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, st, 48, hipMemcpyDeviceToHost));
    auto f = [](unsigned long long b) { union { uint32_t u; float x; } c; c.u = (uint32_t)b; return c.x; };
    printf("all finite amax >= 0 (2139095040 values): d = amax/127 differs for %llu (amax in [%g, %g]); id = 1/d differs for %llu (amax in [%g, %g])\n", h[0],
           h[0] ? f(h[2]) : 0.0, h[0] ? f(h[3]) : 0.0, h[1], h[1] ? f(h[4]) : 0.0, h[1] ? f(h[5]) : 0.0);
    // the range the engine would use the short forms in: 2^-100 <= amax < 2^100
    const uint32_t lo = (127u - 100u) << 23, hi = (127u + 100u) << 23;
    const unsigned long long z[2] = {0, 0};
    CK(hipMemcpy(st, z, 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check_range, dim3(4096), dim3(256), 0, 0, st, lo, hi);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, st, 16, hipMemcpyDeviceToHost));
    printf("2^-100 <= amax < 2^100 (%u values): d differs for %llu, id differs for %llu\n", hi - lo, h[0], h[1]);
    return 0;
}
