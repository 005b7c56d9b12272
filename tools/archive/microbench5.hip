// Round-2 calibration, part 2: where do the ~2-3 us per launch go that are neither the 1.58 us boundary nor the
// kernel's own dependent work?  (a) straight-line code: first pass (cold instruction cache) vs second pass (warm);
// (b) end of one kernel -> first instruction of the next, on the constant 100 MHz clock (wall_clock64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define I8 "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
#define I64 I8 I8 I8 I8 I8 I8 I8 I8
#define I512 I64 I64 I64 I64 I64 I64 I64 I64
// 2048 independent-enough VALU instructions (8 rotating accumulators), 4 bytes each = 8 KB of straight-line code
#define BODY2048 asm volatile(I512 I512 I512 I512 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
__global__ void k_straight(int passes, unsigned long long *cyc, float *o) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, c = 1.0f;
    unsigned long long t[5];
    t[0] = __builtin_readcyclecounter();
    for (int p = 0; p < passes && p < 4; p++) {
        BODY2048
        t[p + 1] = __builtin_readcyclecounter();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int p = 0; p <= passes && p < 5; p++) cyc[p] = t[p];
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == -1.f) o[0] = a0;
}
// (b) gap: each kernel records min entry / max exit wall clock over all its waves
__global__ void k_gap(unsigned long long *slots, int k, int work) {
    const unsigned long long t0 = wall_clock64();
    float a = threadIdx.x;
    for (int i = 0; i < work; i++) a = a * 1.0001f + 0.5f;
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&slots[2 * k], t0);
        atomicMax(&slots[2 * k + 1], wall_clock64() + (a == -1.f ? 1 : 0));
    }
}

template <typename F> float time_graph(hipStream_t st, int reps, int per_graph, F enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < per_graph; i++) enqueue(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / (reps * per_graph);
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *cyc; CK(hipMalloc(&cyc, 4096)); float *o; CK(hipMalloc(&o, 64));
    unsigned long long h[8];
    printf("== (a) straight-line 2048 VALU instructions (8 KB of code): passes over the SAME code inside one launch ==\n");
    for (int blk : {64, 256, 1024}) {
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k_straight, 256, blk, 0, st, 3, cyc, o);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost));
            printf("  block %4d run %d: pass 1 %6llu cyc, pass 2 %6llu, pass 3 %6llu\n", blk, rep, h[1] - h[0], h[2] - h[1], h[3] - h[2]);
        }
    }
    printf("  the same kernel in a graph of 100 (1 pass / 3 passes), 256 x 256: %.2f / %.2f us per kernel\n",
           time_graph(st, 20, 100, [&](int) { hipLaunchKernelGGL(k_straight, 256, 256, 0, st, 1, cyc, o); }),
           time_graph(st, 20, 100, [&](int) { hipLaunchKernelGGL(k_straight, 256, 256, 0, st, 3, cyc, o); }));
    printf("== (b) last exit of kernel k -> first entry of kernel k+1 (100 MHz wall clock, 10 ns ticks), graph of 8 kernels ==\n");
    unsigned long long *slots; CK(hipMalloc(&slots, 16 * 16));
    for (int grid : {16, 128, 256}) for (int blk : {256, 1024}) for (int work : {0, 2000}) {
        std::vector<unsigned long long> init(16);
        for (int k = 0; k < 8; k++) { init[2 * k] = ~0ull; init[2 * k + 1] = 0; }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < 8; k++) hipLaunchKernelGGL(k_gap, grid, blk, 0, st, slots, k, work);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemcpy(slots, init.data(), 128, hipMemcpyHostToDevice));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        }
        std::vector<unsigned long long> r(16);
        CK(hipMemcpy(r.data(), slots, 128, hipMemcpyDeviceToHost));
        double gap = 0, dur = 0, spread = 0;
        for (int k = 1; k < 8; k++) { gap += (double)(long long)(r[2 * k] - r[2 * k - 1]); dur += (double)(r[2 * k + 1] - r[2 * k]); }
        printf("  %3d WGs x %4d thr, work %4d: exit->entry %.2f us, kernel span (first entry -> last exit) %.2f us, entry->entry %.2f us\n", grid, blk, work,
               gap / 7 * 0.01, dur / 7 * 0.01, (double)(r[14] - r[0]) / 7 * 0.01);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
