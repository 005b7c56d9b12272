// Round-2 calibration, part 6: VALU issue cost of the instructions of the exact attention contraction (product rounded to
// f32, double accumulation): cycles per wave-instruction with 4 waves per SIMD all issuing the same instruction class.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ __launch_bounds__(1024) void k_op(unsigned long long *cyc, float *o, float seed) {
    float f0 = threadIdx.x * seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, c = 1.0001f;
    double d0 = f0, d1 = f1, d2 = f2, d3 = f3, dc = 1.0001;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {f0, f1}, p1 = {f2, f3}, pc = {c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 16; it++) {
        if (OP == 0) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(c));) }
        if (OP == 1) { REP64(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dc));) }
        if (OP == 2) { REP64(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));) }
        if (OP == 3) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dc));) }
        if (OP == 4) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(p0), "+v"(p1) : "v"(pc));) }
        if (OP == 5) { REP64(asm volatile("v_lshrrev_b32 %0, 3, %0\n v_add_u32 %1, 0x38000000, %1\n v_lshlrev_b32 %2, 29, %2\n v_and_b32 %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(c));) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t0; }
    if (f0 + f1 + f2 + f3 + (float)(d0 + d1 + d2 + d3) + p0.x + p1.y == -1.f) o[0] = 1.f;
}
int main() {
    unsigned long long *cyc, h[2]; float *o;
    CK(hipMalloc(&cyc, 64)); CK(hipMalloc(&o, 64));
    const char *names[6] = {"v_mul_f32", "v_add_f64", "v_cvt_f64_f32", "v_fma_f64", "v_pk_mul_f32", "int ops (shift/add/and)"};
    for (int op = 0; op < 6; op++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (op) {
                case 0: hipLaunchKernelGGL(k_op<0>, 256, 1024, 0, 0, cyc, o, 0.5f); break;
                case 1: hipLaunchKernelGGL(k_op<1>, 256, 1024, 0, 0, cyc, o, 0.5f); break;
                case 2: hipLaunchKernelGGL(k_op<2>, 256, 1024, 0, 0, cyc, o, 0.5f); break;
                case 3: hipLaunchKernelGGL(k_op<3>, 256, 1024, 0, 0, cyc, o, 0.5f); break;
                case 4: hipLaunchKernelGGL(k_op<4>, 256, 1024, 0, 0, cyc, o, 0.5f); break;
                default: hipLaunchKernelGGL(k_op<5>, 256, 1024, 0, 0, cyc, o, 0.5f); break;
            }
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
        const double n = 16.0 * 64 * 4;   // instructions per wave
        printf("%-26s wave 0: %.2f cycles/instr; until all 16 waves of the workgroup are done: %.2f cycles/instr per wave = %.2f SIMD cycles per wave-instruction (4 waves/SIMD)\n",
               names[op], h[0] / n, h[1] / n, h[1] / n / 4);
    }
    return 0;
}
