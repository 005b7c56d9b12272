// Translation unit of the many-column chain on the int8 matrix cores (kernels_mfma.hip.h): 5 block formats x {q/k/v, out_proj, fc1, fc2, lm_head} = 25 software-pipelined
// kernels + the image builder.  Its own unit so that the library builds in parallel (engine.hip was 90 s alone with them); same arrangement as xpipe_tu.hip: own namespace
// name for the headers' non-inline kernels, the parameter blocks cross as bytes.
#define bgk bgk_mm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

#include "kernels_mfma.hip.h"

namespace {

std::mutex g_attr_mu;
std::set<std::pair<int, const void *>> g_attr_done;      // (device, kernel): > 64 KB of dynamic LDS needs the opt-in attribute once per device

bool walk_ok() { static const bool on = [] { const char *e = getenv("BIOGPT_HIP_MFMA_WALK"); return !(e && e[0] == '0'); }(); return on; }

template <int WT, int EPI, int K, int J = 1>
hipError_t launch_one(const bgk::MatvecParams &p, const bgk::DevMatrix &img, hipStream_t st) {
    const size_t sm = bgk::matmul_mfma_smem_bytes(K, bgk::TypeInfo<WT>::q81, EPI == bgk::EPI_GELU_Q8, J);
    const void *fn = reinterpret_cast<const void *>(bgk::matmul_mfma_kernel<WT, EPI, K, J>);
    if (sm > 64 * 1024) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lk(g_attr_mu);
        if (!g_attr_done.count({dev, fn})) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            if (e != hipSuccess) return e;
            g_attr_done.insert({dev, fn});
        }
    }
    hipLaunchKernelGGL((bgk::matmul_mfma_kernel<WT, EPI, K, J>), dim3((p.W.M + 64 * J - 1) / (64 * J), (p.N + 15) / 16), dim3(bgk::mfma_threads(K)), sm, st, p, img);
    return hipGetLastError();
}

template <int WT>
hipError_t launch_t(int op, const bgk::MatvecParams &p, const bgk::DevMatrix &img, hipStream_t st) {
    switch (op) {      // the chain's sites (engine.hip ChainOp)
        // fc1, the tallest site (Q4_0 / Q5_0 / Q8_0): a wave walks two row tiles when the rows allow it (32 | M) and the launch still has >= 2 workgroups per compute unit
        // (kernels_mfma.hip.h, J); BIOGPT_HIP_MFMA_WALK=0 (read once per process): the one-tile kernel (A/B)
        case 0:
            if constexpr (!bgk::TypeInfo<WT>::q81) {
                if (walk_ok() && p.W.M % 32 == 0 && (int64_t)(p.W.M / 128) * ((p.N + 15) / 16) >= 512) return launch_one<WT, bgk::EPI_QKV, 1024, 2>(p, img, st);
            }
            return launch_one<WT, bgk::EPI_QKV, 1024>(p, img, st);
        case 1: return launch_one<WT, bgk::EPI_RESID, 1024>(p, img, st);
        case 2:
            if constexpr (!bgk::TypeInfo<WT>::q81) {
                if (walk_ok() && p.W.M % 32 == 0 && (int64_t)(p.W.M / 128) * ((p.N + 15) / 16) >= 512) return launch_one<WT, bgk::EPI_GELU_Q8, 1024, 2>(p, img, st);
            }
            return launch_one<WT, bgk::EPI_GELU_Q8, 1024>(p, img, st);
        case 3: return launch_one<WT, bgk::EPI_RESID, 4096>(p, img, st);
        case 4: return launch_one<WT, bgk::EPI_LOGITS, 1024>(p, img, st);
        default: return hipErrorInvalidValue;
    }
}

template <int WT>
hipError_t retile_t(const bgk::DevMatrix &src, uint8_t *dq, uint8_t *ds, hipStream_t st) {
    const int64_t n = (int64_t)src.M * (src.K / bgk::QK);
    hipLaunchKernelGGL((bgk::retile_kernel<WT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dq, ds);
    return hipGetLastError();
}

}  // namespace

// wt: the kernels' WType value (2, 3, 6, 7, 8); op: 0 q/k/v, 1 out_proj, 2 fc1 (GELU + Q8), 3 fc2, 4 lm_head; params: a bgk::MatvecParams; img: a bgk::DevMatrix (the row-tiled image)
extern "C" int bg_mfma_launch(int wt, int op, const void *params, size_t params_bytes, const void *img, size_t img_bytes, hipStream_t st) {
    if (!params || !img || params_bytes != sizeof(bgk::MatvecParams) || img_bytes != sizeof(bgk::DevMatrix)) return (int)hipErrorInvalidValue;
    const bgk::MatvecParams &p = *static_cast<const bgk::MatvecParams *>(params);
    const bgk::DevMatrix &im = *static_cast<const bgk::DevMatrix *>(img);
    switch (wt) {
        case bgk::W_Q4_0: return (int)launch_t<bgk::W_Q4_0>(op, p, im, st);
        case bgk::W_Q4_1: return (int)launch_t<bgk::W_Q4_1>(op, p, im, st);
        case bgk::W_Q5_0: return (int)launch_t<bgk::W_Q5_0>(op, p, im, st);
        case bgk::W_Q5_1: return (int)launch_t<bgk::W_Q5_1>(op, p, im, st);
        case bgk::W_Q8_0: return (int)launch_t<bgk::W_Q8_0>(op, p, im, st);
        default: return (int)hipErrorInvalidValue;
    }
}

// the row-tiled, expanded image of one matrix (retile_kernel): src = the SoA arena matrix
extern "C" int bg_mfma_retile(int wt, const void *src, size_t src_bytes, uint8_t *dq, uint8_t *ds, hipStream_t st) {
    if (!src || src_bytes != sizeof(bgk::DevMatrix)) return (int)hipErrorInvalidValue;
    const bgk::DevMatrix &m = *static_cast<const bgk::DevMatrix *>(src);
    switch (wt) {
        case bgk::W_Q4_0: return (int)retile_t<bgk::W_Q4_0>(m, dq, ds, st);
        case bgk::W_Q4_1: return (int)retile_t<bgk::W_Q4_1>(m, dq, ds, st);
        case bgk::W_Q5_0: return (int)retile_t<bgk::W_Q5_0>(m, dq, ds, st);
        case bgk::W_Q5_1: return (int)retile_t<bgk::W_Q5_1>(m, dq, ds, st);
        case bgk::W_Q8_0: return (int)retile_t<bgk::W_Q8_0>(m, dq, ds, st);
        default: return (int)hipErrorInvalidValue;
    }
}
