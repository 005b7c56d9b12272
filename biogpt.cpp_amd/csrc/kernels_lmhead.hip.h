// Final LayerNorm + lm_head of a single token (biogpt.cpp:799-811, last row only: F8) as ONE pass with every weight byte requested up front.
//
// matvec_fast_kernel<PRO_LN, EPI_LOGITS> gives a 64-row block to a 256-thread workgroup: 663 workgroups of 36 KB, 7.0 us for the 24.6 MB (43 % of 8 TB/s).  A pure read
// of the same bytes (tools/microbench17.hip, profiles/microbench17_lm_head_read_r3.txt: the Q4_0 model lives in the 256 MB Infinity Cache between steps) takes 4.4 us in
// that launch shape and 2.4 us as 249 workgroups x 512 threads with 12 sixteen-byte loads per thread issued before anything else.  So: NB = 3 blocks of 64 rows per
// workgroup of 8 waves -- the layout of the lm_head stage inside the pipelined launch (kernels_xpipe.hip.h): lane = one 32-weight block of a row, 12 block units per lane,
// all loaded at entry; LayerNorm + Q8 of the column by 4 waves meanwhile (ln4_q8_1024); int8 dots (unit_dot_quant), the 32 block terms of a row through LDS and summed
// in block order by one lane (sum32_in_order) -- bit for bit the stand-alone kernel's and the oracle's arithmetic; the same per-64-row-block arg-max partials, so every
// consumer (the next step's sampler, topk_kernel, argmax_kernel) is unchanged; block 0 moves the device-side position on.
#pragma once

#include "kernels_decode.hip.h"

namespace bgk {

template <int NB>
__host__ __device__ constexpr size_t lm_stream_smem_bytes() { return (size_t)NB * 64 * DEC_PS * 4; }

template <int WT, int NB, int NW>
__global__ __launch_bounds__(NW * 64) void lm_stream_kernel(const MatvecParams p) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant, "block-quantized weights");
    static_assert((NB * 64) % (2 * NW) == 0 && 64 % NW == 0, "rows per wave step");
    constexpr int LMS = NB * 64 / (2 * NW);       // block units per lane: rows 2 NW s + 2 wave + (lane >> 5), s < LMS
    extern __shared__ __attribute__((aligned(16))) unsigned char lm_smem[];
    float *const s_part = reinterpret_cast<float *>(lm_smem);      // [NW][2 LMS rows][DEC_PS]
    __shared__ double s_red[8];
    __shared__ __attribute__((aligned(16))) uint32_t s_xq[256];
    __shared__ float s_xd[32];
    __shared__ uint32_t s_xs[32];
    __shared__ float s_redf[NB * NW];
    __shared__ int s_redi[NB * NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 31, rsub = lane >> 5;
    const bool worker = tid < 256;
    const int M = p.W.M, row0 = blockIdx.x * NB * 64;
    // ---- the column and the LayerNorm vectors FIRST (a wave's loads return in order: behind 100 KB of weights they would arrive last, and the LayerNorm -- 0.9 us of
    //      barriers and double sums -- would start when the stream is over instead of running beside it) ----
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), lnw = xv, lnb = xv;
    if (worker) {
        xv = reinterpret_cast<const float4 *>(p.x)[tid];
        lnw = reinterpret_cast<const float4 *>(p.ln_w)[tid]; lnb = reinterpret_cast<const float4 *>(p.ln_b)[tid];
    }
    asm volatile("" : "+v"(xv.x), "+v"(lnw.x), "+v"(lnb.x));      // keep them in front of the weight loads
    // ---- then every weight byte of this workgroup's rows ----
    Unit<WT> wl[LMS];
#pragma unroll
    for (int s = 0; s < LMS; s++) {
        const int row = row0 + s * 2 * NW + wave * 2 + rsub;
        if (row < M) load_unit<WT>(wl[s], p.W, (int64_t)row * 32 + sub);
        else { wl[s].q0 = make_uint4(0u, 0u, 0u, 0u); wl[s].q1 = wl[s].q0; wl[s].sc = 0u; wl[s].qh = 0u; }
    }
    // ---- final LayerNorm + Q8 (waves 0-3; ends with a workgroup barrier) ----
    ln4_q8_1024<TI::q81>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
    uint32_t ax[8];
    const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
    ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
    const float axd = s_xd[sub];
    const uint32_t axs = s_xs[sub];
    float *const part = s_part + wave * 2 * LMS * DEC_PS;
#pragma unroll
    for (int s = 0; s < LMS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = unit_dot_quant<WT>(wl[s], ax, axd, __uint_as_float(axs), (int)axs);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane < 2 LMS finishes local row (lane >> 1) 2 NW + 2 wave + (lane & 1): lanes 8 j .. 8 j + 7 hold rows of block j
    float best_val = -INFINITY;
    int best_idx = 0x7fffffff;
    if (lane < 2 * LMS) {
        const int row = row0 + (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);
        if (row < M) {
            const float v = sum32_in_order(part + lane * DEC_PS);
            p.out[row] = v;
            best_val = v; best_idx = row;
        }
    }
    constexpr int LPB = 64 / NW;      // finisher lanes per 64-row block in one wave
#pragma unroll
    for (int off = 1; off < LPB; off <<= 1) {
        const float ov = __shfl_xor(best_val, off, 64);
        const int oi = __shfl_xor(best_idx, off, 64);
        if (ov > best_val || (ov == best_val && oi < best_idx)) { best_val = ov; best_idx = oi; }
    }
    if (p.pmax_val == nullptr) return;
    if (lane < 2 * LMS && (lane & (LPB - 1)) == 0) { s_redf[(lane / LPB) * NW + wave] = best_val; s_redi[(lane / LPB) * NW + wave] = best_idx; }
    __syncthreads();
    if (tid < NB) {
        float bv = s_redf[tid * NW];
        int bi = s_redi[tid * NW];
#pragma unroll
        for (int w = 1; w < NW; w++) {
            const float ov = s_redf[tid * NW + w];
            const int oi = s_redi[tid * NW + w];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        const int blk = blockIdx.x * NB + tid;
        if (blk * 64 < M) { p.pmax_val[blk] = bv; p.pmax_idx[blk] = bi; }
        // fused decode step (kernels_decode.hip.h): every kernel of this step has read the position by now
        if (blk == 0 && p.st_adv != nullptr && p.adv != 0) { p.st_adv->n_past += p.adv; p.st_adv->n_gen += p.adv; }
    }
}

}  // namespace bgk
