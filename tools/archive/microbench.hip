// Latency calibration for the launch-bound decode path (MI355X): what does a kernel boundary, a
// dependent global load, a block barrier and a DPP chain step cost at the clocks this workload sees?
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_empty() {}
__global__ void k_store(float *o) { if (threadIdx.x == 0) o[blockIdx.x] = 1.0f; }
// chain of dependent loads: idx = buf[idx]
__global__ void k_chase(const int *buf, int steps, int *out) {
    int idx = threadIdx.x + blockIdx.x * blockDim.x;
    for (int s = 0; s < steps; s++) idx = buf[idx];
    if (idx == -1) out[0] = idx;
}
__global__ void k_chase_timed(const int *buf, int steps, long long *cyc) {
    int idx = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; s++) idx = buf[idx];
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = idx; }
}
__global__ void k_syncs(int n, float *o) {
    __shared__ float s[256];
    float v = threadIdx.x;
    for (int i = 0; i < n; i++) { s[threadIdx.x] = v; __syncthreads(); v = s[(threadIdx.x + 64) & 255] + 1.0f; __syncthreads(); }
    if (v == -5.f) o[0] = v;
}
__global__ void k_dpp(int n, float *o) {
    float acc = threadIdx.x, c = 1.0f;
    for (int i = 0; i < n; i++) {
        float t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x138, 0xf, 0xf, true));
        acc = ((threadIdx.x & 31) == 0) ? acc : t + c;
    }
    if (acc == -5.f) o[0] = acc;
}
__global__ void k_clock(long long *out, int iters) {
    long long t0 = __builtin_readcyclecounter();
    long long w0 = wall_clock64();
    float a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = (long long)a; }
}
// producer/consumer across a kernel boundary: producer writes vector, consumer reads it (fresh data latency)
__global__ void k_prod(float *v, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = i * 0.5f; }
__global__ void k_cons(const float *v, int n, float *o) {
    __shared__ float s[256];
    float a = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += v[i];
    s[threadIdx.x] = a; __syncthreads();
    if (threadIdx.x == 0) { float t = 0; for (int k = 0; k < 256; k++) t += s[k]; o[blockIdx.x] = t; }
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
    const int N = 64 << 20;  // 256 MB of ints: pointer-chase over HBM
    int *buf; CK(hipMalloc(&buf, (size_t)N * 4));
    std::vector<int> h(N);
    // random-ish permutation with large strides (defeats caches): idx -> (idx * 1664525 + 1013904223) mod N
    for (long long i = 0; i < N; i++) h[i] = (int)((i * 1664525LL + 1013904223LL) % N);
    CK(hipMemcpy(buf, h.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    int *small; CK(hipMalloc(&small, 4096 * 4));
    std::vector<int> hs(4096); for (int i = 0; i < 4096; i++) hs[i] = (i * 17 + 5) % 4096;
    CK(hipMemcpy(small, hs.data(), 4096 * 4, hipMemcpyHostToDevice));
    float *o; CK(hipMalloc(&o, 1 << 20));
    int *oi; CK(hipMalloc(&oi, 64));
    long long *cyc; CK(hipMalloc(&cyc, 64));
    long long hc[4];

    printf("empty kernel, 1 block x64:   %.2f us/kernel (graph of 200)\n", time_graph(st, 20, 200, [&](int) { hipLaunchKernelGGL(k_empty, 1, 64, 0, st); }));
    printf("empty kernel, 256 blocks x256: %.2f us/kernel\n", time_graph(st, 20, 200, [&](int) { hipLaunchKernelGGL(k_empty, 256, 256, 0, st); }));
    printf("store kernel, 256 blocks x256: %.2f us/kernel\n", time_graph(st, 20, 200, [&](int) { hipLaunchKernelGGL(k_store, 256, 256, 0, st, o); }));
    for (int steps : {1, 2, 4, 8, 16}) {
        float a = time_graph(st, 20, 100, [&](int) { hipLaunchKernelGGL(k_chase, 1, 64, 0, st, buf, steps, oi); });
        float b = time_graph(st, 20, 100, [&](int) { hipLaunchKernelGGL(k_chase, 1, 64, 0, st, small, steps, oi); });
        printf("chase %2d dependent loads: HBM-wide %.2f us/kernel, 16KB-hot %.2f us/kernel\n", steps, a, b);
    }
    for (int steps : {4, 64}) {
        hipLaunchKernelGGL(k_chase_timed, 1, 64, 0, st, buf, steps, cyc); CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
        printf("in-kernel: %d HBM-wide dependent loads = %lld cycles (%.0f /load)\n", steps, hc[0], (double)hc[0] / steps);
        hipLaunchKernelGGL(k_chase_timed, 1, 64, 0, st, small, steps, cyc); CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
        printf("in-kernel: %d hot dependent loads = %lld cycles (%.0f /load)\n", steps, hc[0], (double)hc[0] / steps);
    }
    for (int n : {0, 4, 16, 64})
        printf("kernel with %2d x2 __syncthreads (256 blocks): %.2f us/kernel\n", n, time_graph(st, 20, 100, [&](int) { hipLaunchKernelGGL(k_syncs, 256, 256, 0, st, n, o); }));
    for (int n : {0, 32, 128, 512})
        printf("kernel with %3d DPP chain steps: %.2f us/kernel\n", n, time_graph(st, 20, 100, [&](int) { hipLaunchKernelGGL(k_dpp, 256, 256, 0, st, n, o); }));
    hipLaunchKernelGGL(k_clock, 1, 64, 0, st, cyc, 1 << 20); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hc, cyc, 24, hipMemcpyDeviceToHost));
    printf("clock: %lld shader cycles in %lld wall ticks (100 MHz?) -> %.0f MHz if wall=100MHz\n", hc[0], hc[1], (double)hc[0] / hc[1] * 100.0);
    printf("prod->cons pair (4 KB vector, 256 consumers): %.2f us/pair\n",
           2 * time_graph(st, 20, 100, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_cons, 256, 256, 0, st, o, 1024, o + 4096); else hipLaunchKernelGGL(k_prod, 4, 256, 0, st, o, 1024); }));
    printf("prod->cons pair (16 KB vector): %.2f us/pair\n",
           2 * time_graph(st, 20, 100, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_cons, 256, 256, 0, st, o, 4096, o + 8192); else hipLaunchKernelGGL(k_prod, 16, 256, 0, st, o, 4096); }));
    // eager (non-graph) launch rate
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k_empty, 256, 256, 0, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eager empty launches: %.2f us/kernel\n", ms * 1e3f / 2000);
    }
    return 0;
}
