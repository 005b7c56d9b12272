// Round-2 calibration, part 7: the exact-attention MAC (v_pk_mul_f32 -> 2 x v_cvt_f64_f32 -> 2 x v_add_f64 into 16 rotating
// double accumulators) as one instruction stream: SIMD cycles per MAC at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k_mac(unsigned long long *cyc, float *o, float seed, int iters) {
    double acc[16];
    for (int i = 0; i < 16; i++) acc[i] = i;
    v2f k0 = {threadIdx.x * seed, 1.f + seed}, k1 = {2.f + seed, 3.f * seed}, q0 = {1.0001f, 0.9999f}, q1 = {1.0002f, 0.9998f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < 16; a += 2) {
            v2f p = (a & 2) ? k0 * q0 : k1 * q1;       // one v_pk_mul_f32 = two products
            acc[a] += (double)p.x;
            acc[a + 1] += (double)p.y;
            k0.x += 1e-7f * a;                         // keep the products live (one extra VALU per pair)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t0; }
    double s = 0; for (int i = 0; i < 16; i++) s += acc[i];
    if (s == -1.0) o[0] = 1.f;
}
int main() {
    unsigned long long *cyc, h[2]; float *o;
    CK(hipMalloc(&cyc, 64)); CK(hipMalloc(&o, 64));
    const int iters = 512;
    for (int blk : {256, 512, 1024}) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_mac, 256, blk, 0, 0, cyc, o, 0.5f, iters); CK(hipDeviceSynchronize()); }
        CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
        const double macs = 16.0 * iters;   // per lane
        printf("%4d threads (%d wave(s) per SIMD): workgroup done after %.1f cycles per MAC per wave = %.2f SIMD cycles per wave-MAC\n", blk, blk / 256,
               h[1] / macs, h[1] / macs / (blk / 256));
    }
    return 0;
}
