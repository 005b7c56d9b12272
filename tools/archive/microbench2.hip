// Instruction-issue calibration: straight-line (cold I-cache) vs looped (hot) VALU code, independent vs dependent.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int N, bool DEP>
__global__ void k_straight(float *o, float a, float b) {
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
#pragma unroll
    for (int i = 0; i < N / 8; i++) {
        if (DEP) { v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; }
        else { v0 = v0 * a + b; v1 = v1 * a + b; v2 = v2 * a + b; v3 = v3 * a + b; v4 = v4 * a + b; v5 = v5 * a + b; v6 = v6 * a + b; v7 = v7 * a + b; }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
template <bool DEP>
__global__ void k_loop(float *o, float a, float b, int iters) {
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    for (int i = 0; i < iters; i++) {
        if (DEP) { v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; v0 = v0 * a + b; }
        else { v0 = v0 * a + b; v1 = v1 * a + b; v2 = v2 * a + b; v3 = v3 * a + b; v4 = v4 * a + b; v5 = v5 * a + b; v6 = v6 * a + b; v7 = v7 * a + b; }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
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
    return ms * 1e3f / (reps * per_graph);
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *o; CK(hipMalloc(&o, 1 << 22));
#define RUN(KERNEL, GRID, BLOCK, LABEL, ...) printf("%-58s %.2f us/kernel\n", LABEL, time_graph(st, 20, 100, [&](int) { KERNEL<<<GRID, BLOCK, 0, st>>>(__VA_ARGS__); }))
    RUN((k_straight<512, false>), 256, 256, "straight-line  512 independent fma (256 WG x 256)", o, 1.0001f, 0.5f);
    RUN((k_straight<2048, false>), 256, 256, "straight-line 2048 independent fma", o, 1.0001f, 0.5f);
    RUN((k_straight<8192, false>), 256, 256, "straight-line 8192 independent fma", o, 1.0001f, 0.5f);
    RUN((k_loop<false>), 256, 256, "loop           512 independent fma", o, 1.0001f, 0.5f, 64);
    RUN((k_loop<false>), 256, 256, "loop          2048 independent fma", o, 1.0001f, 0.5f, 256);
    RUN((k_loop<false>), 256, 256, "loop          8192 independent fma", o, 1.0001f, 0.5f, 1024);
    RUN((k_straight<2048, true>), 256, 256, "straight-line 2048 dependent fma", o, 1.0001f, 0.5f);
    RUN((k_loop<true>), 256, 256, "loop          2048 dependent fma", o, 1.0001f, 0.5f, 256);
    RUN((k_straight<2048, false>), 16, 1024, "straight-line 2048 independent fma (16 WG x 1024)", o, 1.0001f, 0.5f);
    RUN((k_loop<false>), 16, 1024, "loop          2048 independent fma (16 WG x 1024)", o, 1.0001f, 0.5f, 256);
    // alternate two different kernels (does the other kernel evict / invalidate the I-cache?)
    printf("%-58s %.2f us/kernel\n", "alternating straight-2048 / straight-512",
           time_graph(st, 20, 100, [&](int i) { if (i & 1) k_straight<2048, false><<<256, 256, 0, st>>>(o, 1.0001f, 0.5f); else k_straight<512, false><<<256, 256, 0, st>>>(o, 1.0001f, 0.5f); }));
    return 0;
}
