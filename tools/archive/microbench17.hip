// What ONE launch can do on the stand-alone lm_head's 24.6 MB (42384 rows x 576 bytes of Q4_0): a pure read with every 16-byte load of a thread issued up front
// (no LayerNorm, no dots), back-to-back launches timed with events -- the floor under matvec_fast_kernel<EPI_LOGITS> (7.0 us, 43 % of 8 TB/s).  Grid shapes: 256 x 512
// threads (96 KB per workgroup), 512 x 256, 663 x 256 (the kernel's own), 1024 x 256; and a 302 MB read for the asymptote.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
template <int PER>
__global__ void rd(const u4v *src, size_t n16, uint32_t *sink) {
    const size_t base = (size_t)blockIdx.x * blockDim.x * PER + threadIdx.x;
    u4v v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) { const size_t k = base + (size_t)i * blockDim.x; v[i] = k < n16 ? __builtin_nontemporal_load(src + k) : u4v{0, 0, 0, 0}; }
    uint32_t a = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) a ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (a == 0x9e3779b9u) sink[0] = a;
}
template <int PER>
static void run(const char *name, const u4v *src, size_t bytes, int threads, uint32_t *sink) {
    const size_t n16 = bytes / 16;
    const int grid = (int)((n16 + (size_t)threads * PER - 1) / ((size_t)threads * PER));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(rd<PER>, dim3(grid), dim3(threads), 0, 0, src, n16, sink);
    hipEventRecord(e0);
    const int reps = 50;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(rd<PER>, dim3(grid), dim3(threads), 0, 0, src, n16, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s grid %5d x %4d, %2d loads per thread: %6.2f us per launch = %5.2f TB/s\n", name, grid, threads, PER, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
}
int main() {
    const size_t lm = (size_t)42384 * 576, big = (size_t)302 << 20;
    u4v *src; uint32_t *sink;
    hipMalloc((void **)&src, big); hipMemset(src, 1, big); hipMalloc((void **)&sink, 64);
    run<12>("24.6 MB, 256 workgroups", src, lm, 512, sink);
    run<12>("24.6 MB, 512 workgroups", src, lm, 256, sink);
    run<9>("24.6 MB, 663-ish workgroups", src, lm, 256, sink);
    run<6>("24.6 MB, 1024 workgroups", src, lm, 256, sink);
    run<3>("24.6 MB, 2048 workgroups", src, lm, 256, sink);
    run<12>("302 MB", src, big, 256, sink);
    return 0;
}
