// Round-2 calibration for the decode redesign (fewer, fatter, head-local kernels).  Questions:
//   A  what does a dependent kernel boundary cost as a function of grid x block shape (16..1024 WGs, 256/1024 threads),
//      dynamic LDS size and kernarg size?
//   B  how fast can ONE workgroup (1024 threads) pull B KB that were issued all at once (per-CU fill rate), when only
//      G workgroups run (G = 16: one per head)?  Source: a 256 MB buffer, distinct slice per launch (no L2 reuse).
//   C  G workgroups each re-reading the SAME 128 KB (block terms written by the previous kernel) -- the cost of the
//      "sum the out_proj block terms in the consumer's prologue" idea.
//   D  last-arriver ticket hand-off inside a launch (sc1 stores -> drain -> agent atomic -> sc1 loads), for reference.
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench4 microbench4.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct BigArgs { float v[60]; float *o; };   // 248-byte kernarg
__global__ void k_empty() {}
__global__ void k_empty_lds(float *o) { extern __shared__ float s[]; if (threadIdx.x == 2000) { s[0] = 1.f; o[0] = s[1]; } }
__global__ void k_bigargs(BigArgs a) { if (threadIdx.x == 0 && a.v[59] == -1.f) a.o[0] = a.v[3]; }

// B: every lane issues NL 16-byte loads at once, sums, writes one value per WG
template <int NL>
__global__ __launch_bounds__(1024) void k_pull(const uint4 *src, size_t wg_stride16, float *o) {
    const uint4 *p = src + (size_t)blockIdx.x * wg_stride16 + threadIdx.x;
    uint4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) v[i] = p[(size_t)i * 1024];
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345u) o[blockIdx.x] = 1.f;
}
// C: thread r reads 32 consecutive floats (8 x 16 B) of a shared [1024][32] array and adds them in order
__global__ __launch_bounds__(1024) void k_terms(const float4 *terms, float *o) {
    const float4 *p = terms + (size_t)threadIdx.x * 8;
    float4 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = p[i];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) { s += t[i].x; s += t[i].y; s += t[i].z; s += t[i].w; }
    if (s == -123.f) o[blockIdx.x] = s;
}
__global__ void k_terms_write(float *terms) { terms[blockIdx.x * blockDim.x + threadIdx.x] = 1.0f; }

// D: ticket hand-off.  G workgroups; each writes 64 floats write-through, drains, takes a ticket; the last arriver reads
// all G x 64 floats with sc1 loads and checks them.  Stamps: [0] first entry, [1] last arriver after ticket, [2] after read.
__global__ __launch_bounds__(256) void k_ticket(float *slab, unsigned *cnt, unsigned long long *stamps, int G, int epoch, int *bad) {
    __shared__ unsigned s_ticket;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (threadIdx.x < 64) __hip_atomic_store(slab + blockIdx.x * 64 + threadIdx.x, (float)(epoch * 1000 + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) atomicMin(stamps, t0);
    if (s_ticket != (unsigned)(G - 1)) return;
    const unsigned long long t1 = __builtin_readcyclecounter();
    int nb = 0;
    for (int i = threadIdx.x; i < G * 64; i += 256) {
        const float v = __hip_atomic_load(slab + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != (float)(epoch * 1000 + i / 64)) nb++;
    }
    if (nb) atomicAdd(bad, nb);
    __syncthreads();
    if (threadIdx.x == 0) { stamps[1] = t1; stamps[2] = __builtin_readcyclecounter(); *cnt = 0; }
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
    float *o; CK(hipMalloc(&o, 1 << 20));
    printf("== A: dependent kernel boundary by launch shape (graph of 200 empty kernels) ==\n");
    for (int blk : {256, 1024})
        for (int grid : {16, 64, 128, 256, 512, 1024})
            printf("  empty %4d WGs x %4d thr: %.2f us/kernel\n", grid, blk, time_graph(st, 20, 200, [&](int) { hipLaunchKernelGGL(k_empty, grid, blk, 0, st); }));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int lds : {0, 16 * 1024, 64 * 1024, 96 * 1024})
        printf("  empty 128 WGs x 1024 thr, %3d KB dynamic LDS: %.2f us/kernel\n", lds / 1024, time_graph(st, 20, 200, [&](int) { hipLaunchKernelGGL(k_empty_lds, 128, 1024, lds, st, o); }));
    { BigArgs a{}; a.o = o; printf("  248-byte kernarg, 128 WGs x 256: %.2f us/kernel\n", time_graph(st, 20, 200, [&](int) { hipLaunchKernelGGL(k_bigargs, 128, 256, 0, st, a); })); }

    printf("== B: one workgroup (1024 thr) pulling B KB issued at once, G workgroups, fresh 256 MB source ==\n");
    const size_t SRC = 256ull << 20;
    uint4 *src; CK(hipMalloc(&src, SRC)); CK(hipMemset(src, 1, SRC));
    auto pull = [&](auto kern, int nl, int G) {
        const size_t per_wg16 = (size_t)nl * 1024;               // uint4 per WG per launch
        const size_t per_launch16 = per_wg16 * G;
        const int nslots = (int)((SRC / 16) / per_launch16);
        const float us = time_graph(st, 10, std::min(200, nslots), [&](int i) { hipLaunchKernelGGL(kern, G, 1024, 0, st, src + (size_t)(i % nslots) * per_launch16, per_wg16, o); });
        printf("  G=%3d  %3d KB/WG: %.2f us/kernel  (%.0f GB/s per WG, %.2f TB/s total)\n", G, nl * 16, us, nl * 16384.0 / us * 1e-3, nl * 16384.0 * G / us * 1e-6);
    };
    for (int G : {16, 64, 128, 256}) {
        pull(k_pull<2>, 2, G); pull(k_pull<4>, 4, G); pull(k_pull<8>, 8, G); pull(k_pull<16>, 16, G);
    }
    printf("== C: G workgroups each reading the same 128 KB [1024][32] term array written by the previous kernel ==\n");
    float *terms; CK(hipMalloc(&terms, 128 * 1024));
    for (int G : {16, 64, 128, 256})
        printf("  G=%3d: write+read pair %.2f us (read kernel alone %.2f us)\n", G,
               2 * time_graph(st, 20, 100, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_terms, G, 1024, 0, st, reinterpret_cast<const float4 *>(terms), o); else hipLaunchKernelGGL(k_terms_write, 32, 1024, 0, st, terms); }),
               time_graph(st, 20, 100, [&](int) { hipLaunchKernelGGL(k_terms, G, 1024, 0, st, reinterpret_cast<const float4 *>(terms), o); }));

    printf("== D: last-arriver ticket (sc1 stores -> drain -> agent fetch_add -> sc1 loads) ==\n");
    float *slab; unsigned *cnt; unsigned long long *stamps; int *bad;
    CK(hipMalloc(&slab, 1024 * 64 * 4)); CK(hipMalloc(&cnt, 64)); CK(hipMalloc(&stamps, 64)); CK(hipMalloc(&bad, 64));
    CK(hipMemset(cnt, 0, 64)); CK(hipMemset(bad, 0, 64));
    for (int G : {16, 24, 64, 256}) {
        double a = 0, b = 0; int n = 0;
        for (int ep = 1; ep <= 40; ep++) {
            unsigned long long init[3] = {~0ull, 0, 0};
            CK(hipMemcpyAsync(stamps, init, 24, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_ticket, G, 256, 0, st, slab, cnt, stamps, G, ep, bad);
            unsigned long long h[3]; CK(hipMemcpyAsync(h, stamps, 24, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            if (ep > 8) { a += (double)(h[1] - h[0]); b += (double)(h[2] - h[1]); n++; }
        }
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("  G=%3d: first entry -> last ticket %.0f cyc, slab read by the last arriver %.0f cyc, stale values %d\n", G, a / n, b / n, hb);
        printf("        kernel incl. boundary: %.2f us\n", time_graph(st, 10, 100, [&](int i) { hipLaunchKernelGGL(k_ticket, G, 256, 0, st, slab, cnt, stamps, G, 1, bad); }));
    }
    return 0;
}
