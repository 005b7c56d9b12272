// Round-2 calibration, part 3: what does the END of a kernel cost as a function of how its outputs were stored?
// producer (G workgroups, each storing B bytes) -> consumer (256 workgroups each reading ALL G*B bytes) pairs in a graph;
// store flavours: plain, sc1 (write-through, agent scope), nt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void k_prod(float *out, int per_wg, float v) {
    for (int i = threadIdx.x; i < per_wg; i += blockDim.x) {
        float *p = out + (size_t)blockIdx.x * per_wg + i;
        const float val = v + i;
        if (MODE == 0) *p = val;
        else if (MODE == 1) __hip_atomic_store(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __builtin_nontemporal_store(val, p);
    }
}
__global__ void k_cons(const float *in, int n, float *o) {
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += in[i];
    if (a == -1.f) o[blockIdx.x] = a;
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
    float *buf; CK(hipMalloc(&buf, 16 << 20)); float *o; CK(hipMalloc(&o, 4096));
    const char *names[3] = {"plain", "sc1", "nt"};
    for (int G : {16, 64, 256}) for (int per_wg : {16, 64, 1024, 8192}) {
        const int n = G * per_wg;
        printf("G=%3d x %5d floats (%4d KB total): ", G, per_wg, n * 4 / 1024);
        for (int mode = 0; mode < 3; mode++) {
            auto prod = [&](float v) {
                if (mode == 0) hipLaunchKernelGGL(k_prod<0>, G, 256, 0, st, buf, per_wg, v);
                else if (mode == 1) hipLaunchKernelGGL(k_prod<1>, G, 256, 0, st, buf, per_wg, v);
                else hipLaunchKernelGGL(k_prod<2>, G, 256, 0, st, buf, per_wg, v);
            };
            const float only = time_graph(st, 20, 100, [&](int i) { prod((float)i); });
            const float pair = 2 * time_graph(st, 20, 100, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_cons, 256, 256, 0, st, buf, n < 4096 ? n : 4096, o); else prod((float)i); });
            printf(" %s: producer alone %.2f us, producer+consumer pair %.2f us |", names[mode], only, pair);
        }
        printf("\n");
    }
    return 0;
}
