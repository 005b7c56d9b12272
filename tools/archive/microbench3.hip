// What does a minimal dependent "phase" cost?  Each kernel reads the 4 KB vector the previous kernel wrote
// (every workgroup reads all of it), and writes its slice of the next vector.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_phase(const float *in, float *out, int n) {   // n = 1024
    const float4 v = reinterpret_cast<const float4 *>(in)[threadIdx.x];          // 256 threads x 16 B = whole vector
    float s = v.x + v.y + v.z + v.w;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const int rows = n / gridDim.x;                                              // rows written by this workgroup
    if ((int)threadIdx.x < rows) out[blockIdx.x * rows + threadIdx.x] = s * 1e-3f + threadIdx.x;
}
__global__ void k_phase_w(const float *in, float *out, int n, const uint4 *w, int wstride) {   // + 16 B of weights per lane
    const uint4 q = w[(size_t)blockIdx.x * wstride + threadIdx.x];
    const float4 v = reinterpret_cast<const float4 *>(in)[threadIdx.x];
    float s = v.x + v.y + v.z + v.w + (float)(q.x & 15) + (float)(q.y & 15) + (float)(q.z & 15) + (float)(q.w & 15);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const int rows = n / gridDim.x;
    if ((int)threadIdx.x < rows) out[blockIdx.x * rows + threadIdx.x] = s * 1e-3f + threadIdx.x;
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
    float *a, *b; CK(hipMalloc(&a, 1 << 16)); CK(hipMalloc(&b, 1 << 16));
    CK(hipMemset(a, 0, 1 << 16)); CK(hipMemset(b, 0, 1 << 16));
    uint4 *w; CK(hipMalloc(&w, (size_t)200 << 20)); CK(hipMemset(w, 1, (size_t)200 << 20));
    for (int grid : {16, 64, 128, 256, 512, 1024}) {
        float t = time_graph(st, 20, 120, [&](int i) { k_phase<<<grid, 256, 0, st>>>((i & 1) ? a : b, (i & 1) ? b : a, 1024); });
        printf("dependent phase, %4d WGs x 256: %.2f us/kernel\n", grid, t);
    }
    for (int grid : {128, 256, 512}) {
        float t = time_graph(st, 20, 120, [&](int i) { k_phase_w<<<grid, 256, 0, st>>>((i & 1) ? a : b, (i & 1) ? b : a, 1024, w + (size_t)(i % 24) * 500000, 256); });
        printf("dependent phase + 4 KB weights/WG, %4d WGs: %.2f us/kernel\n", grid, t);
    }
    return 0;
}
