// Translation unit of the column-per-XCD chunk launches (kernels_xcols.hip.h: biogpt_eval with 2 .. 8 tokens as one persistent launch): 5 block formats x 4 context
// variants (<= 64 / 128 / 256 / 512 keys).  Same arrangement as xpipe_tu.hip: own namespace name for the headers' non-inline kernels, the parameter block crosses as bytes.
#define bgk bgk_xc
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "kernels_xcols.hip.h"

namespace {

template <int WT>
hipError_t launch_t(int t_cap, size_t sm, hipStream_t st, const bgk::XcParams &xc) {
    if (t_cap <= 64) hipLaunchKernelGGL((bgk::dec_xcols_kernel<WT, 8, 64>), dim3(256), dim3(512), sm, st, xc);
    else if (t_cap <= 128) hipLaunchKernelGGL((bgk::dec_xcols_kernel<WT, 4, 128>), dim3(256), dim3(512), sm, st, xc);
    else if (t_cap <= 256) hipLaunchKernelGGL((bgk::dec_xcols_kernel<WT, 2, 256>), dim3(256), dim3(512), sm, st, xc);
    else if (t_cap <= 512 && xc.seq == nullptr) hipLaunchKernelGGL((bgk::dec_xcols_kernel<WT, 2, 512>), dim3(256), dim3(512), sm, st, xc);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

template <int WT>
hipError_t set_lds_t(size_t sm) {
    const void *fns[4] = {reinterpret_cast<const void *>(bgk::dec_xcols_kernel<WT, 8, 64>), reinterpret_cast<const void *>(bgk::dec_xcols_kernel<WT, 4, 128>),
                          reinterpret_cast<const void *>(bgk::dec_xcols_kernel<WT, 2, 256>), reinterpret_cast<const void *>(bgk::dec_xcols_kernel<WT, 2, 512>)};
    for (const void *fn : fns) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

// wt: the kernels' WType value (2, 3, 6, 7, 8); params: a bgk::XcParams
extern "C" int bg_xcols_launch(int wt, int t_cap, size_t smem_bytes, hipStream_t st, const void *params, size_t params_bytes) {
    if (!params || params_bytes != sizeof(bgk::XcParams)) return (int)hipErrorInvalidValue;
    const bgk::XcParams &xc = *static_cast<const bgk::XcParams *>(params);
    switch (wt) {
        case bgk::W_Q4_0: return (int)launch_t<bgk::W_Q4_0>(t_cap, smem_bytes, st, xc);
#ifndef BIOGPT_HIP_ONLY_Q4_0
        case bgk::W_Q4_1: return (int)launch_t<bgk::W_Q4_1>(t_cap, smem_bytes, st, xc);
        case bgk::W_Q5_0: return (int)launch_t<bgk::W_Q5_0>(t_cap, smem_bytes, st, xc);
        case bgk::W_Q5_1: return (int)launch_t<bgk::W_Q5_1>(t_cap, smem_bytes, st, xc);
        case bgk::W_Q8_0: return (int)launch_t<bgk::W_Q8_0>(t_cap, smem_bytes, st, xc);
#endif
        default: return (int)hipErrorInvalidValue;
    }
}

// > 64 KB of dynamic LDS needs the opt-in attribute (per device); set outside any stream capture
extern "C" int bg_xcols_set_lds(int wt, size_t smem_bytes) {
    switch (wt) {
        case bgk::W_Q4_0: return (int)set_lds_t<bgk::W_Q4_0>(smem_bytes);
#ifndef BIOGPT_HIP_ONLY_Q4_0
        case bgk::W_Q4_1: return (int)set_lds_t<bgk::W_Q4_1>(smem_bytes);
        case bgk::W_Q5_0: return (int)set_lds_t<bgk::W_Q5_0>(smem_bytes);
        case bgk::W_Q5_1: return (int)set_lds_t<bgk::W_Q5_1>(smem_bytes);
        case bgk::W_Q8_0: return (int)set_lds_t<bgk::W_Q8_0>(smem_bytes);
#endif
        default: return (int)hipErrorInvalidValue;
    }
}
