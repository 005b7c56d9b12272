// Translation unit of the XCD-pipelined decode launches (kernels_xpipe.hip.h: contexts up to 256 keys): 5 block formats x 5 context variants (<= 64 / 128 / 192 / 256 keys, and 257 .. 512 with two workgroups per head) = 25
// persistent kernels here, the 20 resident ones in xpipe_res_tu.hip, the 20 long-context ones (kernels_xlong.hip.h) in four units of five (xlong_*_tu.hip), compiled apart
// from engine.hip so that all of them build in parallel.  The kernel headers define non-inline __global__ functions, so this unit sees them under its own namespace name; the
// parameter block crosses the boundary as bytes (same header, same layout; the size is checked).
#define bgk bgk_xp
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "kernels_xpipe.hip.h"

extern "C" int bg_xpipe_launch_resident(int wt, int t_cap, size_t smem_bytes, hipStream_t st, const void *params, size_t params_bytes);
extern "C" int bg_xpipe_set_lds_resident(int wt, size_t smem_bytes);
// kernels_xlong.hip.h, one translation unit per (resident form, keys per helper): xlong_32_tu.hip, xlong_64_tu.hip, xlong_res_32_tu.hip, xlong_res_64_tu.hip
#define XLONG_DECL(tag) \
    extern "C" int bg_xlong_launch_##tag(int wt, int t_cap, size_t smem_bytes, hipStream_t st, const void *params, size_t params_bytes); \
    extern "C" int bg_xlong_set_lds_##tag(int wt, size_t smem_bytes);
XLONG_DECL(32) XLONG_DECL(64) XLONG_DECL(res_32) XLONG_DECL(res_64)
static int bg_xpipe_launch_long(int wt, int t_cap, size_t sm, hipStream_t st, const void *params, size_t bytes, bool resident) {
    if (t_cap <= 512) return resident ? bg_xlong_launch_res_32(wt, t_cap, sm, st, params, bytes) : bg_xlong_launch_32(wt, t_cap, sm, st, params, bytes);
    return resident ? bg_xlong_launch_res_64(wt, t_cap, sm, st, params, bytes) : bg_xlong_launch_64(wt, t_cap, sm, st, params, bytes);
}
static int bg_xpipe_set_lds_long(int wt, size_t sm) {
    for (int (*fn)(int, size_t) : {bg_xlong_set_lds_32, bg_xlong_set_lds_64, bg_xlong_set_lds_res_32, bg_xlong_set_lds_res_64}) {
        const int e = fn(wt, sm);
        if (e != (int)hipSuccess) return e;
    }
    return (int)hipSuccess;
}

namespace {

template <int WT>
hipError_t launch_t(int t_cap, size_t sm, hipStream_t st, const bgk::XpParams &xp) {
    // 8 waves per workgroup: 14-16 weight units per lane, unpacked to 9 registers each, + the head's old keys / values (<= 256 keys) or this
    // workgroup's key range of the next layer (beyond) fit the 256-register budget
    if (xp.resident != 0 && t_cap > 256 && !(t_cap <= 512 && xp.dual != 0 && xp.gran_l != nullptr)) return (hipError_t)bg_xpipe_launch_long(WT, t_cap, sm, st, &xp, sizeof(xp), true);
    if (xp.resident != 0) return (hipError_t)bg_xpipe_launch_resident(WT, t_cap, sm, st, &xp, sizeof(xp));      // its own translation unit (xpipe_res_tu.hip)
    // measurement only (BIOGPT_HIP_XPIPE_AS_RES=1, an engine option like every other): ordinary launches through the RES instantiations with resident = 0 -- what the resident form's exits cost the chain itself
    const bool as_res = xp.as_res != 0;
    if (as_res && t_cap <= 256) return (hipError_t)bg_xpipe_launch_resident(WT, t_cap, sm, st, &xp, sizeof(xp));
    if (t_cap <= 64) hipLaunchKernelGGL((bgk::dec_xpipe_kernel<WT, 8, 8, 64, true>), dim3(256), dim3(512), sm, st, xp);
    else if (t_cap <= 128) hipLaunchKernelGGL((bgk::dec_xpipe_kernel<WT, 4, 8, 128, true>), dim3(256), dim3(512), sm, st, xp);
    else if (t_cap <= 192) hipLaunchKernelGGL((bgk::dec_xpipe_kernel<WT, 2, 8, 192, true>), dim3(256), dim3(512), sm, st, xp);   // 24 instead of 32 value registers
    else if (t_cap <= 256) hipLaunchKernelGGL((bgk::dec_xpipe_kernel<WT, 2, 8, 256, true>), dim3(256), dim3(512), sm, st, xp);
    else if (t_cap <= 512 && xp.dual != 0 && xp.gran_l != nullptr) hipLaunchKernelGGL((bgk::dec_xpipe_kernel<WT, 2, 8, 512, true>), dim3(256), dim3(512), sm, st, xp);   // two workgroups per head, 256 keys each
    else return (hipError_t)bg_xpipe_launch_long(WT, t_cap, sm, st, &xp, sizeof(xp), false);      // beyond 256 keys
    return hipGetLastError();
}

template <int WT>
hipError_t set_lds_t(size_t sm) {
    if (bg_xpipe_set_lds_resident(WT, sm) != (int)hipSuccess || bg_xpipe_set_lds_long(WT, sm) != (int)hipSuccess) return hipErrorInvalidValue;
    const void *fns[5] = {reinterpret_cast<const void *>(bgk::dec_xpipe_kernel<WT, 8, 8, 64, true>), reinterpret_cast<const void *>(bgk::dec_xpipe_kernel<WT, 4, 8, 128, true>),
                          reinterpret_cast<const void *>(bgk::dec_xpipe_kernel<WT, 2, 8, 192, true>), reinterpret_cast<const void *>(bgk::dec_xpipe_kernel<WT, 2, 8, 256, true>),
                          reinterpret_cast<const void *>(bgk::dec_xpipe_kernel<WT, 2, 8, 512, true>)};
    for (const void *fn : fns) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

// wt: the kernels' WType value (2, 3, 6, 7, 8); params: a bgk::XpParams
extern "C" int bg_xpipe_launch(int wt, int t_cap, size_t smem_bytes, hipStream_t st, const void *params, size_t params_bytes) {
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

// > 64 KB of dynamic LDS needs the opt-in attribute (per device); set outside any stream capture
extern "C" int bg_xpipe_set_lds(int wt, size_t smem_bytes) {
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
