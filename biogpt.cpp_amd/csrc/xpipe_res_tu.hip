// Translation unit of the RESIDENT pipelined decode launches (kernels_xpipe.hip.h with RES = true: biogpt_hip_eval's launch that stays on the
// device between calls): 5 block formats x 5 context variants (<= 64 / 128 / 192 / 256 keys, 257 .. 512 with two workgroups per head).  Same arrangement as xpipe_tu.hip (own namespace name, parameter block as bytes).
#define bgk bgk_xr
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "kernels_xpipe.hip.h"

namespace {

template <int WT>
hipError_t launch_t(int t_cap, size_t sm, hipStream_t st, const bgk::XpParams &xp) {
    if (t_cap <= 64) hipLaunchKernelGGL((bgk::dec_xpipe_kernel<WT, 8, 8, 64, true, true>), dim3(256), dim3(512), sm, st, xp);
    else if (t_cap <= 128) hipLaunchKernelGGL((bgk::dec_xpipe_kernel<WT, 4, 8, 128, true, true>), dim3(256), dim3(512), sm, st, xp);
    else if (t_cap <= 192) hipLaunchKernelGGL((bgk::dec_xpipe_kernel<WT, 2, 8, 192, true, true>), dim3(256), dim3(512), sm, st, xp);
    else if (t_cap <= 256) hipLaunchKernelGGL((bgk::dec_xpipe_kernel<WT, 2, 8, 256, true, true>), dim3(256), dim3(512), sm, st, xp);
    else if (t_cap <= 512 && xp.dual != 0 && xp.gran_l != nullptr) hipLaunchKernelGGL((bgk::dec_xpipe_kernel<WT, 2, 8, 512, true, true>), dim3(256), dim3(512), sm, st, xp);   // two workgroups per head, 256 keys each
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

template <int WT>
hipError_t set_lds_t(size_t sm) {
    const void *fns[5] = {reinterpret_cast<const void *>(bgk::dec_xpipe_kernel<WT, 8, 8, 64, true, true>), reinterpret_cast<const void *>(bgk::dec_xpipe_kernel<WT, 4, 8, 128, true, true>),
                          reinterpret_cast<const void *>(bgk::dec_xpipe_kernel<WT, 2, 8, 192, true, true>), reinterpret_cast<const void *>(bgk::dec_xpipe_kernel<WT, 2, 8, 256, true, true>),
                          reinterpret_cast<const void *>(bgk::dec_xpipe_kernel<WT, 2, 8, 512, true, true>)};
    for (const void *fn : fns) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

extern "C" int bg_xpipe_launch_resident(int wt, int t_cap, size_t smem_bytes, hipStream_t st, const void *params, size_t params_bytes) {
    if (!params || params_bytes != sizeof(bgk::XpParams)) return (int)hipErrorInvalidValue;
    const bgk::XpParams &xp = *static_cast<const bgk::XpParams *>(params);
    switch (wt) {
        case bgk::W_Q4_0: return (int)launch_t<bgk::W_Q4_0>(t_cap, smem_bytes, st, xp);
#ifndef BIOGPT_HIP_ONLY_Q4_0      // (experiment builds: only the Q4_0 kernels, a quarter of the compile time)
        case bgk::W_Q4_1: return (int)launch_t<bgk::W_Q4_1>(t_cap, smem_bytes, st, xp);
        case bgk::W_Q5_0: return (int)launch_t<bgk::W_Q5_0>(t_cap, smem_bytes, st, xp);
        case bgk::W_Q5_1: return (int)launch_t<bgk::W_Q5_1>(t_cap, smem_bytes, st, xp);
        case bgk::W_Q8_0: return (int)launch_t<bgk::W_Q8_0>(t_cap, smem_bytes, st, xp);
#endif
        default: return (int)hipErrorInvalidValue;
    }
}

extern "C" int bg_xpipe_set_lds_resident(int wt, size_t smem_bytes) {
    switch (wt) {
        case bgk::W_Q4_0: return (int)set_lds_t<bgk::W_Q4_0>(smem_bytes);
#ifndef BIOGPT_HIP_ONLY_Q4_0      // (experiment builds: only the Q4_0 kernels, a quarter of the compile time)
        case bgk::W_Q4_1: return (int)set_lds_t<bgk::W_Q4_1>(smem_bytes);
        case bgk::W_Q5_0: return (int)set_lds_t<bgk::W_Q5_0>(smem_bytes);
        case bgk::W_Q5_1: return (int)set_lds_t<bgk::W_Q5_1>(smem_bytes);
        case bgk::W_Q8_0: return (int)set_lds_t<bgk::W_Q8_0>(smem_bytes);
#endif
        default: return (int)hipErrorInvalidValue;
    }
}
