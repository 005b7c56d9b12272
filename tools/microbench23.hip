// attn_tile_kernel<16, MB_DMA> (csrc/kernels_fast.hip.h) alone: one layer's attention of a 512-column prompt pass (-b 8 visibility), with per-workgroup stamps at the phase borders.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DATTN_STAMPS -Ibiogpt.cpp_amd/csrc -Iinclude -o tools/microbench23 tools/microbench23.hip && tools/microbench23
#include "kernels_fast.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
using namespace bgk;
#ifndef MB_DMA
#define MB_DMA true
#endif
typedef unsigned long long u64;
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 512, H = 16, D = 1024, P = 1024, DK = 64;
    float *q, *kc, *vc, *out; uint16_t *et; DevState *st; int8_t *oq; float *od; uint32_t *os; u64 *ts;
    hipMalloc((void **)&q, (size_t)N * D * 4); hipMalloc((void **)&kc, (size_t)P * D * 4); hipMalloc((void **)&vc, (size_t)P * D * 4); hipMalloc((void **)&out, (size_t)N * D * 4);
    hipMalloc((void **)&et, 65536 * 2); hipMalloc((void **)&st, 64 + 8 * P); hipMalloc((void **)&oq, (size_t)N * D); hipMalloc((void **)&od, (size_t)N * 32 * 4); hipMalloc((void **)&os, (size_t)N * 32 * 4);
    const int ny = (N + 15) / 16;
    hipMalloc((void **)&ts, (size_t)H * ny * 8 * 8);
    std::vector<float> h((size_t)P * D);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((int)((i * 2654435761u >> 9) % 2001) - 1000) * 1e-3f;
    hipMemcpy(kc, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(vc, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(q, h.data(), (size_t)N * D * 4, hipMemcpyHostToDevice);
    std::vector<uint16_t> e(65536);
    for (int i = 0; i < 65536; i++) { __half hv = __ushort_as_half((uint16_t)i); float f = __half2float(hv); e[i] = __half_as_ushort(__float2half(std::isfinite(f) ? expf(f) : 0.f)); }
    hipMemcpy(et, e.data(), 65536 * 2, hipMemcpyHostToDevice);
    DevState hs{}; hs.n_past = 0; hs.causal = 0; hs.chunk = 8; hipMemset(st, 0, 64 + 8 * P); hipMemcpy(st, &hs, sizeof(hs), hipMemcpyHostToDevice);
    AttnParams a{}; a.q = q; a.kcache = kc; a.vcache = vc; a.out = out; a.st = st; a.exp_tab = et; a.N = N; a.D = D; a.dk = DK; a.P = P; a.t_cap = std::min(P, N);
    a.oq_q = oq; a.oq_d = od; a.oq_s = os; a.tstamp = ts;
    const size_t smb = attn_tile_smem_bytes<16>(a.t_cap);
    hipFuncSetAttribute(reinterpret_cast<const void *>(attn_tile_kernel<16, MB_DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_tile_smem_bytes<16>(P));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((attn_tile_kernel<16, MB_DMA>), dim3(H, ny), dim3(512), smb, 0, a);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((attn_tile_kernel<16, MB_DMA>), dim3(H, ny), dim3(512), smb, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const size_t nwg = (size_t)H * ny;
    std::vector<u64> t(nwg * 8); hipMemcpy(t.data(), ts, nwg * 64, hipMemcpyDeviceToHost);
    u64 w0 = ~0ull, w1 = 0; for (size_t w = 0; w < nwg; w++) { w0 = std::min(w0, t[w * 8 + 5]); w1 = std::max(w1, t[w * 8 + 6]); }
    printf("N %d: %zu workgroups, LDS %zu B, %.2f us per launch (back to back), first entry -> last end %.2f us\n", N, nwg, smb, ms * 1000 / reps, (double)(w1 - w0) * 0.01);
    printf("tile (keys)  entry(us)  end(us) | cycles: scores softmax    PV   tail | clock GHz | cu\n");
    const int nhalf = (ny + 1) / 2;
    for (int yb = 0; yb < ny; yb++) {
        const int i0 = ((yb < nhalf) ? ny - 1 - yb : yb - nhalf) * 16;
        const u64 *x = &t[((size_t)yb * H + 0) * 8];      // head 0 of this tile row
        double sc = 0, sm = 0, pv = 0, tl = 0, en = 0, ex = 0, ghz = 0;
        for (int hh = 0; hh < H; hh++) { const u64 *y = &t[((size_t)yb * H + hh) * 8]; sc += y[1] - y[0]; sm += y[2] - y[1]; pv += y[3] - y[2]; tl += y[4] - y[3]; en += (double)(y[5] - w0); ex += (double)(y[6] - w0); ghz += (double)(y[4] - y[0]) / ((double)(y[6] - y[5]) * 10.0); }
        printf("y %2d %4d keys  %8.2f %8.2f | %7.0f %7.0f %7.0f %6.0f | %.2f | %x.%03x\n", yb, i0 + 16 < N ? ((i0 + 16 + 7) / 8) * 8 : N, en / H * 0.01, ex / H * 0.01, sc / H, sm / H, pv / H, tl / H, ghz / H, (unsigned)(x[7] >> 32), (unsigned)((x[7] >> 8) & 0xff));
    }
    return 0;
}
