// Round-2 calibration, part 5: kernel-argument latency.  A dependent chain of tiny kernels (each reads the vector the
// previous one wrote and writes the next): arguments as plain pointers vs one by-value struct; built twice, without and
// with  -mllvm -amdgpu-kernarg-preload-count=N  (gfx950 firmware preloads the first kernarg dwords into SGPRs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
struct Args { const float *in; float *out; float add; int pad[20]; };
__global__ void k_ptr(const float *in, float *out, float add) { out[blockIdx.x * blockDim.x + threadIdx.x] = in[blockIdx.x * blockDim.x + threadIdx.x] + add; }
__global__ void k_struct(const Args a) { a.out[blockIdx.x * blockDim.x + threadIdx.x] = a.in[blockIdx.x * blockDim.x + threadIdx.x] + a.add; }
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
int main(int argc, char **argv) {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *a, *b; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
    for (int grid : {16, 64}) for (int blk : {256, 1024}) {
        const float tp = time_graph(st, 20, 200, [&](int i) { hipLaunchKernelGGL(k_ptr, grid, blk, 0, st, (i & 1) ? b : a, (i & 1) ? a : b, 1.0f); });
        const float ts = time_graph(st, 20, 200, [&](int i) { Args g{}; g.in = (i & 1) ? b : a; g.out = (i & 1) ? a : b; g.add = 1.0f; hipLaunchKernelGGL(k_struct, grid, blk, 0, st, g); });
        printf("%s: %3d x %4d: pointer args %.3f us/kernel, 104-byte struct arg %.3f us/kernel\n", argv[0], grid, blk, tp, ts);
    }
    return 0;
}
