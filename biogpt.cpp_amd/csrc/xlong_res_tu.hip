// Translation unit of the RESIDENT long-context pipelined decode launches (kernels_xlong.hip.h with RES = true: biogpt_hip_eval's loop beyond 256 keys): 5 block
// formats x 2.  Same arrangement as xlong_tu.hip.
#define bgk bgk_xlr
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "kernels_xlong.hip.h"

namespace {

template <int WT>
hipError_t launch_t(int t_cap, size_t sm, hipStream_t st, const bgk::XpParams &xp) {
    if (xp.gran_l == nullptr || xp.resident == 0 || t_cap <= 256 || t_cap > 1024) return hipErrorInvalidValue;
    if (t_cap <= 512) hipLaunchKernelGGL((bgk::dec_xlong_kernel<WT, 32, true>), dim3(256), dim3(512), sm, st, xp);
    else hipLaunchKernelGGL((bgk::dec_xlong_kernel<WT, 64, true>), dim3(256), dim3(512), sm, st, xp);
    return hipGetLastError();
}

template <int WT>
hipError_t set_lds_t(size_t sm) {
    const void *fns[2] = {reinterpret_cast<const void *>(bgk::dec_xlong_kernel<WT, 32, true>), reinterpret_cast<const void *>(bgk::dec_xlong_kernel<WT, 64, true>)};
    for (const void *fn : fns) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

extern "C" int bg_xpipe_launch_long_resident(int wt, int t_cap, size_t smem_bytes, hipStream_t st, const void *params, size_t params_bytes) {
    if (!params || params_bytes != sizeof(bgk::XpParams)) return (int)hipErrorInvalidValue;
    const bgk::XpParams &xp = *static_cast<const bgk::XpParams *>(params);
    switch (wt) {
        case bgk::W_Q4_0: return (int)launch_t<bgk::W_Q4_0>(t_cap, smem_bytes, st, xp);
        case bgk::W_Q4_1: return (int)launch_t<bgk::W_Q4_1>(t_cap, smem_bytes, st, xp);
        case bgk::W_Q5_0: return (int)launch_t<bgk::W_Q5_0>(t_cap, smem_bytes, st, xp);
        case bgk::W_Q5_1: return (int)launch_t<bgk::W_Q5_1>(t_cap, smem_bytes, st, xp);
        case bgk::W_Q8_0: return (int)launch_t<bgk::W_Q8_0>(t_cap, smem_bytes, st, xp);
        default: return (int)hipErrorInvalidValue;
    }
}

extern "C" int bg_xpipe_set_lds_long_resident(int wt, size_t smem_bytes) {
    switch (wt) {
        case bgk::W_Q4_0: return (int)set_lds_t<bgk::W_Q4_0>(smem_bytes);
        case bgk::W_Q4_1: return (int)set_lds_t<bgk::W_Q4_1>(smem_bytes);
        case bgk::W_Q5_0: return (int)set_lds_t<bgk::W_Q5_0>(smem_bytes);
        case bgk::W_Q5_1: return (int)set_lds_t<bgk::W_Q5_1>(smem_bytes);
        case bgk::W_Q8_0: return (int)set_lds_t<bgk::W_Q8_0>(smem_bytes);
        default: return (int)hipErrorInvalidValue;
    }
}
