// Final LayerNorm + lm_head of a single token (biogpt.cpp:799-811, last row only: F8) as ONE pass.
//
// matvec_fast_kernel<PRO_LN, EPI_LOGITS> gives a 64-row block to a 256-thread workgroup: 663 workgroups of 36 KB, 7.1 us for Q4_0's 24.6 MB (43 % of 8 TB/s).  A pure read
// of the same bytes (tools/microbench17.hip, profiles/microbench17_lm_head_read_r3.txt: the Q4_0 model lives in the 256 MB Infinity Cache between steps) takes 4.4 us in
// that launch shape and 2.4 us as 249 workgroups x 512 threads with every 16-byte load of a thread issued up front.  Here a workgroup of 8 waves takes three 64-row
// blocks in the layout of the lm_head stage inside the pipelined launch (kernels_xpipe.hip.h): lane = one 32-weight block of a row; int8 dots (unit_dot_quant), the 32
// block terms of a row through LDS and summed in block order by one lane (sum32_in_order) -- bit for bit the block kernel's and the oracle's arithmetic; the same
// per-64-row-block arg-max partials, so every consumer (the next step's sampler, topk_kernel, argmax_kernel) is unchanged; block 0 moves the device-side position on.
// What the stage stamps (tools/lm_head_timeline.py) taught, in the order it went in: weight scales as ONE contiguous 16-byte-per-lane piece through LDS instead of a
// 2-byte load per unit (a load instruction costs the memory pipeline the same whatever its width); that piece and the column requested BEFORE the weights (a wave's
// loads return in order); the logits stores behind the last barrier; the four waves that run the LayerNorm ask for their weights after it, the other four at entry.
// Q4_0 7.13 -> 5.77 us (53 % of 8 TB/s), Q5_1 9.06 -> 7.5 (55 %), Q8_0 9.79 -> 9.5 (61 %).
#pragma once

#include "kernels_decode.hip.h"

namespace bgk {

// dynamic LDS: the block terms [8 waves][32 rows][DEC_PS] f32, then the workgroup's weight scales [192 rows][32] (fp16 d, or half2 {d, m})
__host__ __device__ constexpr size_t lm_stream_smem_bytes(bool q81) { return (size_t)8 * 32 * DEC_PS * 4 + (size_t)192 * 32 * (q81 ? 4 : 2); }

// One workgroup = 8 waves = three 64-row blocks = 96 row pairs.  Waves 4-7 ("early": they only wait at the LayerNorm's barriers) take 16 pairs each -- blocks 0 and 1 --
// and ask for their weights at entry; waves 0-3 run the LayerNorm first (a wave is blocked while the compute unit's memory pipeline takes its loads) and then take 8
// pairs each -- block 2.  (12 + 12 pairs with the same order of requests measured the same, 5.8 us; every wave loading at entry 6.0.)
template <int WT>
__global__ __launch_bounds__(512) void lm_stream_kernel(const MatvecParams p) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant, "block-quantized weights");
    seq_forward(p.lineage, SEQ_LM_HEAD);
    constexpr int NW = 8, NB = 3, LE = 16, LL = 8;       // block units per lane of an early / a LayerNorm wave: 4 LE + 4 LL = 96 row pairs
    extern __shared__ __attribute__((aligned(16))) unsigned char lm_smem[];
    unsigned char *const s_sc = lm_smem + (size_t)NW * 32 * DEC_PS * 4;
    __shared__ double s_red[8];
    __shared__ __attribute__((aligned(16))) uint32_t s_xq[256];
    __shared__ float s_xd[32];
    __shared__ uint32_t s_xs[32];
    __shared__ float s_redf[NB * 4];
    __shared__ int s_redi[NB * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), sub = lane & 31, rsub = lane >> 5;
    const bool worker = tid < 256, early = wave >= 4;
    const int wq = wave & 3, pair0 = early ? 0 : 4 * LE;      // unit s of this wave: row pair pair0 + 4 s + wq, rows 2 pair + (lane >> 5)
    const int M = p.W.M, row0 = blockIdx.x * NB * 64;
    float *const part = reinterpret_cast<float *>(lm_smem) + wave * 32 * DEC_PS;
    BG_STAMP(0);
    // ---- the column and the LayerNorm vectors FIRST (a wave's loads return in order: behind its weights they would arrive last) ----
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), lnw = xv, lnb = xv;
    if (worker) {
        xv = reinterpret_cast<const float4 *>(p.x)[tid];
        lnw = reinterpret_cast<const float4 *>(p.ln_w)[tid]; lnb = reinterpret_cast<const float4 *>(p.ln_b)[tid];
    }
    asm volatile("" : "+v"(xv.x), "+v"(lnw.x), "+v"(lnb.x));
    // the rows' scales: one contiguous piece of the scale array (12 / 24 KB), 16 bytes per lane, through LDS -- a 2-byte load per unit costs the compute unit's
    // memory pipeline as much as the unit's 16 weight bytes (measured: 4500 cycles until a wave's 24 loads were issued)
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
    Unit<WT> wl[LE];
    auto load_weights = [&](auto nc) __attribute__((always_inline)) {
        constexpr int N = decltype(nc)::value;
#pragma unroll
        for (int s = 0; s < N; s++) {
            const int row = row0 + 2 * (pair0 + 4 * s + wq) + rsub;
            wl[s].q0 = make_uint4(0u, 0u, 0u, 0u); wl[s].q1 = wl[s].q0; wl[s].sc = 0u; wl[s].qh = 0u;
            if (row < M) {
                const int64_t idx = (int64_t)row * 32 + sub;
                wl[s].q0 = *reinterpret_cast<const uint4 *>(p.W.qs + idx * TI::qbytes);
                if (WT == W_Q8_0) wl[s].q1 = reinterpret_cast<const uint4 *>(p.W.qs + idx * TI::qbytes)[1];
                if (WT == W_Q5_0 || WT == W_Q5_1) wl[s].qh = p.W.qh[idx];
            }
        }
    };
    if (early) load_weights(std::integral_constant<int, LE>{});
    // (the scales were requested before the weights, so writing them to LDS now waits for them only)
#pragma unroll
    for (int j = 0; j < NSL; j++)
        if (tid + j * NW * 64 < n16) reinterpret_cast<uint4 *>(s_sc)[tid + j * NW * 64] = sc_stage[j];
    BG_STAMP(1);
    if (p.dbg & 32) { asm volatile("" :: "v"(xv.x)); BG_STAMP(2); }
    // ---- final LayerNorm + Q8 (waves 0-3; ends with a workgroup barrier) ----
    ln4_q8_1024<TI::q81>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
    if (!early) load_weights(std::integral_constant<int, LL>{});
    BG_STAMP(3);
    uint32_t ax[8];
    const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
    ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
    const float axd = s_xd[sub];
    const uint32_t axs = s_xs[sub];
    auto dots = [&](auto nc) __attribute__((always_inline)) {
        constexpr int N = decltype(nc)::value;
#pragma unroll
        for (int s = 0; s < N; s++) {
            const int lr = 2 * (pair0 + 4 * s + wq) + rsub;
            if (TI::q81) wl[s].sc = reinterpret_cast<const uint32_t *>(s_sc)[lr * 32 + sub];
            else wl[s].sc = reinterpret_cast<const uint16_t *>(s_sc)[lr * 32 + sub];
            part[(s * 2 + rsub) * DEC_PS + sub] = unit_dot_quant<WT>(wl[s], ax, axd, __uint_as_float(axs), (int)axs);
        }
    };
    if (early) dots(std::integral_constant<int, LE>{}); else dots(std::integral_constant<int, LL>{});
    BG_STAMP(4);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane < 2 N finishes the row 2 (pair0 + 4 (lane >> 1) + wq) + (lane & 1): 16 consecutive lanes hold rows of one 64-row block
    float best_val = -INFINITY, my_val = 0.0f;
    int best_idx = 0x7fffffff;
    const int my_pair = pair0 + 4 * (lane >> 1) + wq, my_row = row0 + 2 * my_pair + (lane & 1);
    const bool mine = lane < 2 * (early ? LE : LL) && my_row < M;
    if (mine) {
        my_val = sum32_in_order(part + lane * DEC_PS);
        best_val = my_val; best_idx = my_row;
    }
    BG_STAMP(5);
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const float ov = __shfl_xor(best_val, off, 64);
        const int oi = __shfl_xor(best_idx, off, 64);
        if (ov > best_val || (ov == best_val && oi < best_idx)) { best_val = ov; best_idx = oi; }
    }
    if (p.pmax_val == nullptr) { if (mine) p.out[my_row] = my_val; return; }
    if (lane < 2 * (early ? LE : LL) && (lane & 15) == 0) { const int bl = my_pair >> 5; s_redf[bl * 4 + wq] = best_val; s_redi[bl * 4 + wq] = best_idx; }
    __syncthreads();
    if (mine) p.out[my_row] = my_val;      // behind the barrier: in front of it the barrier would wait for the stores to land
    if (tid < NB) {
        float bv = s_redf[tid * 4];
        int bi = s_redi[tid * 4];
#pragma unroll
        for (int w = 1; w < 4; w++) {
            const float ov = s_redf[tid * 4 + w];
            const int oi = s_redi[tid * 4 + w];
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
