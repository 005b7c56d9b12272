// Translation unit of the persistent float-weight decode launch (kernels_fpipe.hip.h: F32 / F16 files, one token, all layers).  Same arrangement as xcols_tu.hip: own
// namespace name for the headers' non-inline kernels, the parameter block crosses as bytes.
#define bgk bgk_fp
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "kernels_fpipe.hip.h"

// wt: 0 F32, 1 F16; params: a bgk::FpParams
extern "C" int bg_fpipe_launch(int wt, hipStream_t st, const void *params, size_t params_bytes) {
    if (!params || params_bytes != sizeof(bgk::FpParams)) return (int)hipErrorInvalidValue;
    const bgk::FpParams &fp = *static_cast<const bgk::FpParams *>(params);
    const size_t sm = bgk::fpipe_smem_bytes();
    if (wt != bgk::W_F32 && wt != bgk::W_F16) return (int)hipErrorInvalidValue;
    if (fp.stamps) {
        if (wt == bgk::W_F32) hipLaunchKernelGGL((bgk::fpipe_kernel<bgk::W_F32, true>), dim3(256), dim3(384), sm, st, fp);
        else hipLaunchKernelGGL((bgk::fpipe_kernel<bgk::W_F16, true>), dim3(256), dim3(384), sm, st, fp);
    } else {
        if (wt == bgk::W_F32) hipLaunchKernelGGL((bgk::fpipe_kernel<bgk::W_F32, false>), dim3(256), dim3(384), sm, st, fp);
        else hipLaunchKernelGGL((bgk::fpipe_kernel<bgk::W_F16, false>), dim3(256), dim3(384), sm, st, fp);
    }
    return (int)hipGetLastError();
}

// 152 KB of dynamic LDS needs the opt-in attribute (per device); set outside any stream capture
extern "C" int bg_fpipe_set_lds(void) {
    const int sm = (int)bgk::fpipe_smem_bytes();
    const void *fs[4] = {reinterpret_cast<const void *>(bgk::fpipe_kernel<bgk::W_F32, false>), reinterpret_cast<const void *>(bgk::fpipe_kernel<bgk::W_F16, false>),
                         reinterpret_cast<const void *>(bgk::fpipe_kernel<bgk::W_F32, true>), reinterpret_cast<const void *>(bgk::fpipe_kernel<bgk::W_F16, true>)};
    for (const void *f : fs) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipSuccess;
}
extern "C" size_t bg_fpipe_params_bytes(void) { return sizeof(bgk::FpParams); }
