// matmul_mfma_kernel (csrc/kernels_mfma.hip.h) alone, on synthetic operands, with per-wave shader-clock stamps: where does a workgroup's time go?
//   stamps of a computing wave: 0 entry, 1 its phase-0 DMAs issued, 2 loop done, (3 unused);  staging wave: 0 entry, 1 DMAs of phase 0 issued, 2 landed, 3 exit
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMFMA_STAMPS -Ibiogpt.cpp_amd/csrc -Iinclude -o tools/microbench22 tools/microbench22.hip && tools/microbench22
#include "kernels_mfma.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace bgk;
typedef unsigned long long u64;

template <int EPI, int K, int J = 1>
void run(const char *name, int M, int N, size_t sm_override = 0) {
    const int BPR = K / 32;
    const size_t nblk = (size_t)M * BPR;
    uint8_t *iq, *is; int8_t *aq; float *ad, *bias, *resid, *out, *kc, *vc, *qo, *od; uint32_t *as, *os; int8_t *oq; u64 *ts; uint16_t *gelu; DevState *st;
    hipMalloc((void **)&iq, nblk * 32); hipMalloc((void **)&is, nblk * 4); hipMalloc((void **)&aq, (size_t)N * K); hipMalloc((void **)&ad, (size_t)N * BPR * 4); hipMalloc((void **)&as, (size_t)N * BPR * 4);
    hipMalloc((void **)&bias, M * 4); hipMalloc((void **)&resid, (size_t)N * M * 4); hipMalloc((void **)&out, (size_t)N * M * 4);
    hipMalloc((void **)&kc, (size_t)16 * 1024 * 64 * 4); hipMalloc((void **)&vc, (size_t)16 * 1024 * 64 * 4); hipMalloc((void **)&qo, (size_t)N * 1024 * 4);
    hipMalloc((void **)&oq, (size_t)N * M); hipMalloc((void **)&od, (size_t)N * (M / 32) * 4); hipMalloc((void **)&os, (size_t)N * (M / 32) * 4);
    hipMalloc((void **)&gelu, 65536 * 2); hipMalloc((void **)&st, 4096 * 8 + 64);
    const dim3 grid((M + 64 * J - 1) / (64 * J), (N + 15) / 16);
    const size_t nst = (size_t)grid.x * grid.y * 5 * 8;
    hipMalloc((void **)&ts, nst * 8);
    std::vector<uint8_t> h(nblk * 32); for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)((i * 2654435761u >> 13) % 15 - 7);
    hipMemcpy(iq, h.data(), h.size(), hipMemcpyHostToDevice);
    std::vector<float> f(nblk); for (size_t i = 0; i < nblk; i++) f[i] = 0.001f + (float)(i % 97) * 1e-5f;
    hipMemcpy(is, f.data(), nblk * 4, hipMemcpyHostToDevice);
    std::vector<int8_t> a((size_t)N * K); for (size_t i = 0; i < a.size(); i++) a[i] = (int8_t)((i * 40503u >> 7) % 255 - 127);
    hipMemcpy(aq, a.data(), a.size(), hipMemcpyHostToDevice);
    std::vector<float> d((size_t)N * BPR, 0.01f); hipMemcpy(ad, d.data(), d.size() * 4, hipMemcpyHostToDevice);
    hipMemset(as, 0, (size_t)N * BPR * 4); hipMemset(bias, 0, M * 4); hipMemset(resid, 0, (size_t)N * M * 4); hipMemset(gelu, 0, 65536 * 2); hipMemset(st, 0, 64);
    MatvecParams p{};
    p.W.M = M; p.W.K = K; p.N = N; p.aq_q = aq; p.aq_d = ad; p.aq_s = as; p.bias = bias; p.resid = resid; p.ldr = M; p.out = out; p.ldo = M;
    p.q_out = qo; p.kcache = kc; p.vcache = vc; p.dk = 64; p.dk_log2 = 6; p.P = 1024; p.D = 1024; p.q_scale = 0.125f; p.st = st; p.gelu_tab = gelu;
    p.oq_q = oq; p.oq_d = od; p.oq_s = os; p.tstamp = ts;
    DevMatrix img{}; img.qs = iq; img.sc = is; img.M = M; img.K = K;
    const size_t sm = sm_override ? sm_override : matmul_mfma_smem_bytes(K, false, EPI == EPI_GELU_Q8, J);
    if (sm > 65536) hipFuncSetAttribute(reinterpret_cast<const void *>(matmul_mfma_kernel<W_Q4_0, EPI, K, J>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((matmul_mfma_kernel<W_Q4_0, EPI, K, J>), grid, dim3(mfma_threads(K)), sm, 0, p, img);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((matmul_mfma_kernel<W_Q4_0, EPI, K, J>), grid, dim3(mfma_threads(K)), sm, 0, p, img);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<u64> t(nst); hipMemcpy(t.data(), ts, nst * 8, hipMemcpyDeviceToHost);
    u64 tmin = ~0ull, tmax = 0;
    double c01 = 0, c12 = 0, c23 = 0, s01 = 0, s12 = 0, s03 = 0; std::vector<double> fin; size_t nw = 0, nwg = grid.x * grid.y;
    std::vector<double> start, wend;
    for (size_t w = 0; w < nwg; w++) {
        for (int v = 0; v < 4; v++) { const u64 *q = &t[(w * 5 + v) * 8]; wend.push_back((double)q[5]); c01 += (double)(q[1] - q[0]); c12 += (double)(q[2] - q[1]); c23 += (double)(q[3] - q[2]); fin.push_back((double)q[7]); s01 += (double)(q[1] - q[0]); nw++; }
        if (K > 1024) { const u64 *q = &t[(w * 5 + 4) * 8]; s12 += (double)(q[2] - q[1]); s03 += (double)(q[3] - q[0]); }
        start.push_back((double)t[(w * 5) * 8 + 4]);
    }
    // workgroups resident on one compute unit at a time: (XCC, SE, SH, CU) of wave 0, interval = [entry, loop end of wave 0]
    int maxconc = 0; double avgconc = 0; int ncu = 0;
    {
        std::vector<std::vector<std::pair<u64, int>>> ev(8 * 4096);
        for (size_t w = 0; w < nwg; w++) {
            const u64 *q = &t[(w * 5) * 8];
            const unsigned hw = (unsigned)q[6], xcc = (unsigned)(q[6] >> 32);
            const unsigned cu = (hw >> 8) & 0xff;      // cu_id[11:8] sh_id[12] se_id[15:13]
            ev[xcc * 4096 + cu].push_back({q[4], +1}); ev[xcc * 4096 + cu].push_back({q[5], -1});
        }
        for (auto &e : ev) if (!e.empty()) {
            std::sort(e.begin(), e.end());
            int c = 0, m = 0; for (auto &x : e) { c += x.second; m = std::max(m, c); }
            if (getenv("MB_GAPS") && ncu < 6) { printf("   cu: "); for (auto &x : e) printf("%s%llu ", x.second > 0 ? "+" : "-", (unsigned long long)(x.first - (u64)0)); printf("\n"); }
            maxconc = std::max(maxconc, m); avgconc += m; ncu++;
        }
        avgconc /= std::max(1, ncu);
    }
    std::sort(start.begin(), start.end()); std::sort(wend.begin(), wend.end()); std::sort(fin.begin(), fin.end());
    printf("%-10s M %5d K %5d N %4d  %4zu workgroups  %7.2f us per launch | computing waves: entry -> phase-0 DMAs issued %7.0f cyc, from there to the loop's end %7.0f cyc, epilogue %6.0f cyc (ends, wall: median +%.0f last +%.0f) | (the same, all waves) +%6.0f; staging wave: landed +%6.0f, exit %7.0f | wall (10 ns ticks): entries median +%.0f last +%.0f, loop ends first +%.0f median +%.0f last +%.0f | %d compute units seen, workgroups at a time on one: max %d, mean of the per-unit maxima %.2f\n",
           name, M, K, N, nwg, ms * 1000.0 / reps, c01 / nw, c12 / nw, c23 / nw, fin[fin.size() / 2] - start[0], fin.back() - start[0], s01 / nw, s12 / nwg, s03 / nwg, start[nwg / 2] - start[0], start[nwg - 1] - start[0], wend[0] - start[0], wend[wend.size() / 2] - start[0], wend.back() - start[0], ncu, maxconc, avgconc);
    hipFree(iq); hipFree(is); hipFree(aq); hipFree(ad); hipFree(as); hipFree(bias); hipFree(resid); hipFree(out); hipFree(kc); hipFree(vc); hipFree(qo); hipFree(oq); hipFree(od); hipFree(os); hipFree(gelu); hipFree(st); hipFree(ts);
}
int main() {
    if (getenv("MB_ONLY")) {      // one shape only (the ablation builds: tools/r6_ablate.sh)
        run<EPI_GELU_Q8, 1024>("fc1", 4096, 512); run<EPI_GELU_Q8, 1024>("fc1", 4096, 64); run<EPI_RESID, 4096>("fc2", 1024, 512);
        return 0;
    }
    if (getenv("MB_GAPS")) {      // per compute unit: workgroup entries (+) and loop ends (-) in 10 ns ticks (the first six units)
        if (atoi(getenv("MB_GAPS")) == 2) run<EPI_GELU_Q8, 1024>("fc1", 4096, 512); else if (atoi(getenv("MB_GAPS")) == 3) run<EPI_GELU_Q8, 1024, 2>("fc1 walk 2", 4096, 512); else run<EPI_RESID, 1024>("out_proj", 1024, 512);
        return 0;
    }
    for (int N : {512, 64}) {
        run<EPI_RESID, 1024>("out_proj", 1024, N);
        run<EPI_RESID, 4096>("fc2", 1024, N);
        run<EPI_QKV, 1024>("q/k/v", 3072, N);
        run<EPI_GELU_Q8, 1024>("fc1", 4096, N);
        run<EPI_GELU_Q8, 1024, 2>("fc1 walk 2", 4096, N);
        run<EPI_LOGITS, 1024>("lm_head", 42384, N);
    }
    return 0;
}
