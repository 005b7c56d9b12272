// What ONE XCD can pull from the fabric (round 4, kernels_xcols.hip.h: every XCD streams all 216 MB of weights per pass; measured there: 0.42 TB/s per XCD).
// 256 workgroups x 512 threads; the 32 workgroups of every ACTIVE XCD (HW_REG_XCC_ID < n_active) read the SAME `bytes` of a buffer -- the others leave -- as 16-byte loads,
// 1 KB contiguous per wave-load (the units of a weight row pair), DEPTH loads per lane in flight, plain or streaming (nt) policy; wall time of the launch by events.
// Arms: 1 / 2 / 4 / 8 active XCDs on one buffer (the Infinity Cache serves all but the first reader), 8 XCDs on 8 different buffers (every byte from HBM), a buffer of
// 7 MB read 31 times (a layer's weights: L2 4 MB per XCD, so it streams) and of 216 MB read once.
//   hipcc --offload-arch=gfx950 -O3 -o microbench20 microbench20.hip && ./microbench20
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
template <int DEPTH, bool NT>
__global__ __launch_bounds__(512) void rd(const u4v *src, size_t n16, size_t stride16, int n_active, int passes, uint32_t *sink) {
    const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);
    if (xcc >= n_active) return;
    __shared__ int s_slot;
    if (threadIdx.x == 0) s_slot = (int)atomicAdd(sink + 16 + xcc, 1u) & 31;
    __syncthreads();
    const int slot = s_slot;                                   // rank inside the XCD (32 workgroups per XCD)
    const u4v *p = src + (size_t)xcc * stride16;
    const size_t per_wg = n16 / 32, base = (size_t)slot * per_wg;
    uint32_t a = 0;
    for (int ps = 0; ps < passes; ps++)
        for (size_t i = threadIdx.x; i + (size_t)(DEPTH - 1) * 512 < per_wg; i += (size_t)DEPTH * 512) {
            u4v v[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; k++) v[k] = NT ? __builtin_nontemporal_load(p + base + i + (size_t)k * 512) : p[base + i + (size_t)k * 512];
#pragma unroll
            for (int k = 0; k < DEPTH; k++) a ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    if (a == 0x9e3779b9u) sink[0] = a;
}
template <int DEPTH, bool NT>
static void run(const char *name, const u4v *src, size_t bytes, size_t stride_bytes, int n_active, int passes, uint32_t *sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 6;
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipMemset(sink, 0, 256);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((rd<DEPTH, NT>), dim3(256), dim3(512), 0, 0, src, bytes / 16, stride_bytes / 16, n_active, passes, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    const double per_xcd = (double)bytes * passes / (best * 1e-3) / 1e12;
    printf("%-58s %d XCD(s), depth %2d%s: %8.1f us = %5.2f TB/s per XCD, %5.2f TB/s in all\n", name, n_active, DEPTH, NT ? " nt" : "   ", best * 1e3, per_xcd, per_xcd * n_active);
}
int main() {
    const size_t model = (size_t)216 << 20, layer = (size_t)7 << 20;
    u4v *src; uint32_t *sink;
    hipMalloc((void **)&src, model * 8); hipMemset(src, 1, model * 8); hipMalloc((void **)&sink, 256);
    for (int n : {1, 2, 4, 8}) run<8, false>("216 MB once, every XCD the SAME buffer", src, model, 0, n, 1, sink);
    run<8, false>("216 MB once, every XCD its OWN buffer (all from HBM)", src, model, model, 8, 1, sink);
    run<8, true>("216 MB once, SAME buffer", src, model, 0, 8, 1, sink);
    run<16, false>("216 MB once, SAME buffer", src, model, 0, 8, 1, sink);
    run<4, false>("216 MB once, SAME buffer", src, model, 0, 8, 1, sink);
    run<2, false>("216 MB once, SAME buffer", src, model, 0, 8, 1, sink);
    run<8, false>("7 MB x 31 (one layer again and again: 4 MB L2)", src, layer, 0, 8, 31, sink);
    run<8, false>("7 MB x 31", src, layer, 0, 1, 31, sink);
    run<8, false>("3 MB x 72 (fits the XCD's L2)", src, (size_t)3 << 20, 0, 8, 72, sink);
    return 0;
}
