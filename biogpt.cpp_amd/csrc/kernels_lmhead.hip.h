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

// dynamic LDS: the block terms [NB * 64 rows][DEC_PS] f32, then the workgroup's weight scales [NB * 64 rows][32] (fp16 d, or half2 {d, m})
__host__ __device__ constexpr size_t lm_stream_smem_bytes(int nb, bool q81) { return (size_t)nb * 64 * DEC_PS * 4 + (size_t)nb * 64 * 32 * (q81 ? 4 : 2); }

template <int WT, int NB, int NW>
__global__ __launch_bounds__(NW * 64) void lm_stream_kernel(const MatvecParams p) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant, "block-quantized weights");
    static_assert((NB * 64) % (2 * NW) == 0 && 64 % NW == 0, "rows per wave step");
    constexpr int LMS = NB * 64 / (2 * NW);       // block units per lane: rows 2 NW s + 2 wave + (lane >> 5), s < LMS
    extern __shared__ __attribute__((aligned(16))) unsigned char lm_smem[];
    float *const s_part = reinterpret_cast<float *>(lm_smem);      // [NW][2 LMS rows][DEC_PS]
    unsigned char *const s_sc = lm_smem + (size_t)NB * 64 * DEC_PS * 4;
    __shared__ double s_red[8];
    __shared__ __attribute__((aligned(16))) uint32_t s_xq[256];
    __shared__ float s_xd[32];
    __shared__ uint32_t s_xs[32];
    __shared__ float s_redf[NB * NW];
    __shared__ int s_redi[NB * NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 31, rsub = lane >> 5;
    const bool worker = tid < 256;
    const int M = p.W.M, row0 = blockIdx.x * NB * 64;
    BG_STAMP(0);
    // ---- the column and the LayerNorm vectors FIRST (a wave's loads return in order: behind 100 KB of weights they would arrive last, and the LayerNorm -- 0.9 us of
    //      barriers and double sums -- would start when the stream is over instead of running beside it) ----
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), lnw = xv, lnb = xv;
    if (worker) {
        xv = reinterpret_cast<const float4 *>(p.x)[tid];
        lnw = reinterpret_cast<const float4 *>(p.ln_w)[tid]; lnb = reinterpret_cast<const float4 *>(p.ln_b)[tid];
    }
    asm volatile("" : "+v"(xv.x), "+v"(lnw.x), "+v"(lnb.x));      // keep them in front of the weight loads
    // the rows' scales: one contiguous piece of the scale array (12 / 24 KB), 16 bytes per lane, through LDS -- a 2-byte load per unit costs the compute unit's
    // memory pipeline as much as the unit's 16 weight bytes (measured: 4500 cycles until a wave's 24 loads are issued, 3300 with 12 + 2)
    constexpr int SB = TI::q81 ? 4 : 2, NSL = (NB * 64 * 32 * SB / 16 + NW * 64 - 1) / (NW * 64);
    const int n16 = min(NB * 64, M - row0) * 32 * SB / 16;
    uint4 sc_stage[NSL];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(p.W.sc) + (size_t)row0 * 32 * SB);
#pragma unroll
        for (int j = 0; j < NSL; j++) {
            sc_stage[j] = make_uint4(0u, 0u, 0u, 0u);
            if (tid + j * NW * 64 < n16) sc_stage[j] = src[tid + j * NW * 64];
        }
    }
    asm volatile("" : "+v"(sc_stage[0].x));
    // ---- then every weight byte of this workgroup's rows ----
    Unit<WT> wl[LMS];
#pragma unroll
    for (int s = 0; s < LMS; s++) {
        const int row = row0 + s * 2 * NW + wave * 2 + rsub;
        wl[s].q0 = make_uint4(0u, 0u, 0u, 0u); wl[s].q1 = wl[s].q0; wl[s].sc = 0u; wl[s].qh = 0u;
        if (row < M) {
            const int64_t idx = (int64_t)row * 32 + sub;
            wl[s].q0 = *reinterpret_cast<const uint4 *>(p.W.qs + idx * TI::qbytes);
            if (WT == W_Q5_0 || WT == W_Q5_1) wl[s].qh = p.W.qh[idx];
        }
    }
    // (the scales were requested before the weights -- a wave's loads return in order, so writing them to LDS now waits for them only)
#pragma unroll
    for (int j = 0; j < NSL; j++)
        if (tid + j * NW * 64 < n16) reinterpret_cast<uint4 *>(s_sc)[tid + j * NW * 64] = sc_stage[j];
    BG_STAMP(1);
    // ---- final LayerNorm + Q8 (waves 0-3; ends with a workgroup barrier) ----
    if (p.dbg & 32) { asm volatile("" :: "v"(xv.x)); BG_STAMP(2); }
    ln4_q8_1024<TI::q81>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
    BG_STAMP(3);
    uint32_t ax[8];
    const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
    ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
    const float axd = s_xd[sub];
    const uint32_t axs = s_xs[sub];
    float *const part = s_part + wave * 2 * LMS * DEC_PS;
#pragma unroll
    for (int s = 0; s < LMS; s++) {
        const int lr = s * 2 * NW + wave * 2 + rsub;
        if (TI::q81) wl[s].sc = reinterpret_cast<const uint32_t *>(s_sc)[lr * 32 + sub];
        else wl[s].sc = reinterpret_cast<const uint16_t *>(s_sc)[lr * 32 + sub];
        part[(s * 2 + rsub) * DEC_PS + sub] = unit_dot_quant<WT>(wl[s], ax, axd, __uint_as_float(axs), (int)axs);
    }
    BG_STAMP(4);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane < 2 LMS finishes local row (lane >> 1) 2 NW + 2 wave + (lane & 1): lanes 8 j .. 8 j + 7 hold rows of block j
    float best_val = -INFINITY, my_val = 0.0f;
    int best_idx = 0x7fffffff;
    const int my_row = row0 + (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);
    const bool mine = lane < 2 * LMS && my_row < M;
    if (mine) {
        my_val = sum32_in_order(part + lane * DEC_PS);
        best_val = my_val; best_idx = my_row;
    }
    BG_STAMP(5);
    constexpr int LPB = 64 / NW;      // finisher lanes per 64-row block in one wave
#pragma unroll
    for (int off = 1; off < LPB; off <<= 1) {
        const float ov = __shfl_xor(best_val, off, 64);
        const int oi = __shfl_xor(best_idx, off, 64);
        if (ov > best_val || (ov == best_val && oi < best_idx)) { best_val = ov; best_idx = oi; }
    }
    if (p.pmax_val == nullptr) { if (mine) p.out[my_row] = my_val; return; }
    if (lane < 2 * LMS && (lane & (LPB - 1)) == 0) { s_redf[(lane / LPB) * NW + wave] = best_val; s_redi[(lane / LPB) * NW + wave] = best_idx; }
    __syncthreads();
    if (mine) p.out[my_row] = my_val;      // behind the barrier: in front of it the barrier would wait for the stores to land
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
    BG_STAMP(6);
}

}  // namespace bgk
