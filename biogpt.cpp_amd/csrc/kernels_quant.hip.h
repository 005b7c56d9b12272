// Block quantizer on the device (biogpt_model_quantize_internal's inner loops, biogpt.cpp:565-603 ->
// ggml_quantize_q4_0 ... q8_0): f32 rows -> the FILE's block formats, byte for byte what csrc/quant_host.cpp (and the
// oracle's codec) produce.  One thread per 32-value block; the arithmetic is spelled with explicit round-to-nearest
// operations so that no contraction can differ from the host's `-ffp-contract=off` build.  HBM-bound byte work: 128 B in,
// 18-34 B out per block.
#pragma once

#include "kernels.hip.h"

namespace bgk {

// f32 -> f16 bits, sign of zero kept: __float2half_rn returns +0 for -0.0f on this toolchain (measured), and the reference's all-zero
// symmetric block stores d = 0 / -8 = -0.0
__device__ __forceinline__ uint16_t f2h_file(float f) {
    return (f == 0.0f) ? (uint16_t)((__float_as_uint(f) >> 16) & 0x8000u) : f2h(f);
}

// file layouts (ggml): q4_0 {f16 d; u8 qs[16]}  q4_1 {f16 d, m; u8 qs[16]}  q5_0 {f16 d; u32 qh; u8 qs[16]}
//                      q5_1 {f16 d, m; u32 qh; u8 qs[16]}  q8_0 {f16 d; i8 qs[32]}
template <int BITS, bool ASYM>
__device__ __forceinline__ void quant_nibble_block(const float (&x)[32], uint8_t *out) {
    constexpr int LEVELS = 1 << BITS;
    float scale, base = 0.0f;
    if (ASYM) {
        float lo = x[0], hi = x[0];
#pragma unroll
        for (int j = 1; j < 32; j++) { lo = fminf(lo, x[j]); hi = fmaxf(hi, x[j]); }
        scale = __fdiv_rn(__fsub_rn(hi, lo), (float)(LEVELS - 1));
        base = lo;
    } else {
        float extreme = 0.0f, mag = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; j++)
            if (fabsf(x[j]) > mag) { mag = fabsf(x[j]); extreme = x[j]; }      // first of equal magnitudes wins, as on the host
        scale = __fmul_rn(extreme, -1.0f / (float)(LEVELS / 2));      // / -8 or / -16, exactly; an all-zero block gives -0.0 like the reference
    }
    const float inv = scale != 0.0f ? __fdiv_rn(1.0f, scale) : 0.0f;
    uint8_t *p = out;
    const uint16_t hs = f2h_file(scale);
    p[0] = (uint8_t)hs; p[1] = (uint8_t)(hs >> 8); p += 2;
    if (ASYM) { const uint16_t hb = f2h_file(base); p[0] = (uint8_t)hb; p[1] = (uint8_t)(hb >> 8); p += 2; }
    uint8_t *qh = nullptr;
    if (BITS == 5) { qh = p; p += 4; }
    uint32_t high_bits = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        int q[2];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const float v = x[j + half * 16];
            int code;
            if (ASYM) {
                const float t = __fadd_rn(__fmul_rn(__fsub_rn(v, base), inv), 0.5f);
                code = (BITS == 4) ? min(LEVELS - 1, (int)(int8_t)(int)t) : (int)(uint8_t)(int)t;
            } else {
                const float t = __fadd_rn(__fmul_rn(v, inv), (float)(LEVELS / 2) + 0.5f);
                code = min(LEVELS - 1, (int)(int8_t)(int)t);
            }
            q[half] = code;
        }
        p[j] = (uint8_t)((q[0] & 0x0F) | ((q[1] & 0x0F) << 4));
        if (BITS == 5) {
            high_bits |= (uint32_t)((q[0] >> 4) & 1) << j;
            high_bits |= (uint32_t)((q[1] >> 4) & 1) << (j + 16);
        }
    }
    if (BITS == 5) { qh[0] = (uint8_t)high_bits; qh[1] = (uint8_t)(high_bits >> 8); qh[2] = (uint8_t)(high_bits >> 16); qh[3] = (uint8_t)(high_bits >> 24); }
}

__device__ __forceinline__ void quant_q8_0_block(const float (&x)[32], uint8_t *out) {
    float mag = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; j++) mag = fmaxf(mag, fabsf(x[j]));
    const float scale = __fdiv_rn(mag, 127.0f);
    const float inv = scale != 0.0f ? __fdiv_rn(1.0f, scale) : 0.0f;
    const uint16_t hs = f2h_file(scale);
    out[0] = (uint8_t)hs; out[1] = (uint8_t)(hs >> 8);
#pragma unroll
    for (int j = 0; j < 32; j++) out[2 + j] = (uint8_t)(int8_t)(int)roundf(__fmul_rn(x[j], inv));   // half away from zero
}

// type: WType ids of the file formats (W_Q4_0 ... W_Q8_0); block_bytes = 18 / 20 / 22 / 24 / 34
__global__ __launch_bounds__(256) void quantize_blocks_kernel(const float *src, uint8_t *dst, long long nblocks, int type, int block_bytes) {
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    float x[32];
    const float4 *s4 = reinterpret_cast<const float4 *>(src + b * 32);
#pragma unroll
    for (int j = 0; j < 8; j++) { const float4 v = s4[j]; x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w; }
    uint8_t *out = dst + b * block_bytes;
    switch (type) {
        case W_Q4_0: quant_nibble_block<4, false>(x, out); break;
        case W_Q4_1: quant_nibble_block<4, true>(x, out); break;
        case W_Q5_0: quant_nibble_block<5, false>(x, out); break;
        case W_Q5_1: quant_nibble_block<5, true>(x, out); break;
        default: quant_q8_0_block(x, out); break;
    }
}

}  // namespace bgk
