// The decode mat-vec over the MODEL'S OWN BYTES in one timing window (SURVEY section 7 "hard parts": "all 24 layers' same-shape mat-vecs batched in one timing window";
// north_star: ">= 70 % of the HBM roofline on the Q4_0 single-token decode mat-vec at d_model = 1024").
//
// Inside a decode step the mat-vecs of one token are a dependent chain of 97 small matrices (0.6 - 2.4 MB each, 24.4 MB for lm_head): no kernel that runs them in order
// can keep 8 TB/s busy, and the pipelined launch's 10 % of the roofline measures that chain's latency, not the mat-vec.  This kernel is the mat-vec ALONE: every block-
// quantized matrix of the arena -- 24 x {q/k/v, out_proj, fc1, fc2} + lm_head, 195.6 MB of Q4_0 -- times a resident Q8 activation vector of its shape, all of them
// independent, rows spread over the whole chip.  The arithmetic per row is the reference's: int8 block dots (unit_dot_quant), block terms added in block order.
//   unit  = 2048 weight blocks = 64 rows of K = 1024 or 16 rows of K = 4096 (36.9 KB of Q4_0), one workgroup of 512 threads; a thread owns 4 blocks, 512 apart:
//           every load of a wave is 1 KB contiguous; all of a thread's loads are issued before anything waits; the unit's scales come as ONE contiguous piece
//           (16 bytes per lane) through LDS, like the activation vector
//   then  the 2048 block terms through LDS (row pitch + 4: conflict-free row walks), one lane per row adds them in block order
#pragma once

#include "kernels_decode.hip.h"

namespace bgk {

struct SweepJob {
    DevMatrix W;
    const int8_t *xq; const float *xd; const uint32_t *xs;   // the activation of this matrix's shape: K int8, K/32 scales, K/32 block sums (Q8_0: int, Q8_1: d * sum)
    float *out;                                               // [M]
    int32_t unit0;                                            // index of the job's first unit in the launch
    int32_t pad;
};
struct SweepParams {
    const SweepJob *jobs;
    const int32_t *unit_job;                                  // [gridDim.x]: the job a unit belongs to
};
constexpr int SWEEP_UNIT = 2048;                              // blocks per unit

template <int WT>
__global__ __launch_bounds__(512) void matvec_sweep_kernel(const SweepParams p) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant, "block-quantized weights");
    constexpr int SB = TI::q81 ? 4 : 2, NI = SWEEP_UNIT / 512, NSC = SWEEP_UNIT * SB / 16 / 512 > 0 ? SWEEP_UNIT * SB / 16 / 512 : 1;
    __shared__ __attribute__((aligned(16))) unsigned char s_sc[SWEEP_UNIT * SB];
    __shared__ __attribute__((aligned(16))) uint32_t s_xq[4096 / 4];
    __shared__ float s_xd[128];
    __shared__ uint32_t s_xs[128];
    __shared__ __attribute__((aligned(16))) float s_part[64 * 36 > 16 * 132 ? 64 * 36 : 16 * 132];
    const int tid = threadIdx.x;
    const SweepJob j = p.jobs[p.unit_job[blockIdx.x]];
    const int K = j.W.K, M = j.W.M, BPR = K >> 5, bshift = K == 1024 ? 5 : 7;   // blocks per row: 32 or 128
    const int RU = SWEEP_UNIT >> bshift, pitch = BPR + 4;
    const int64_t idx0 = (int64_t)((int)blockIdx.x - j.unit0) * SWEEP_UNIT, nblk = (int64_t)M * BPR;
    // ---- everything this thread reads from memory, requested up front ----
    Unit<WT> wl[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int64_t idx = idx0 + tid + 512 * i;
        wl[i].q0 = make_uint4(0u, 0u, 0u, 0u); wl[i].q1 = wl[i].q0; wl[i].sc = 0u; wl[i].qh = 0u;
        if (idx < nblk) {
            wl[i].q0 = *reinterpret_cast<const uint4 *>(j.W.qs + idx * TI::qbytes);
            if (WT == W_Q8_0) wl[i].q1 = reinterpret_cast<const uint4 *>(j.W.qs + idx * TI::qbytes)[1];
            if (WT == W_Q5_0 || WT == W_Q5_1) wl[i].qh = j.W.qh[idx];
        }
    }
    uint4 sc_st[NSC];
#pragma unroll
    for (int i = 0; i < NSC; i++) {
        const int piece = tid + 512 * i;                       // 16-byte piece of the unit's scales
        sc_st[i] = make_uint4(0u, 0u, 0u, 0u);
        if (piece < SWEEP_UNIT * SB / 16 && (idx0 + (int64_t)piece * (16 / SB)) < nblk) sc_st[i] = reinterpret_cast<const uint4 *>(j.W.sc + idx0 * SB)[piece];
    }
    uint4 xq_st = make_uint4(0u, 0u, 0u, 0u);
    if (tid < K / 16) xq_st = reinterpret_cast<const uint4 *>(j.xq)[tid];
    float xd_st = 0.0f; uint32_t xs_st = 0u;
    if (tid < BPR) { xd_st = j.xd[tid]; xs_st = j.xs[tid]; }
#pragma unroll
    for (int i = 0; i < NSC; i++) if (tid + 512 * i < SWEEP_UNIT * SB / 16) reinterpret_cast<uint4 *>(s_sc)[tid + 512 * i] = sc_st[i];
    if (tid < K / 16) reinterpret_cast<uint4 *>(s_xq)[tid] = xq_st;
    if (tid < BPR) { s_xd[tid] = xd_st; s_xs[tid] = xs_st; }
    __syncthreads();
    // ---- block terms ----
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int bb = tid + 512 * i, blk = bb & (BPR - 1), r = bb >> bshift;
        wl[i].sc = TI::q81 ? reinterpret_cast<const uint32_t *>(s_sc)[bb] : (uint32_t)reinterpret_cast<const uint16_t *>(s_sc)[bb];
        const uint32_t xs = s_xs[blk];
        s_part[r * pitch + blk] = unit_dot_quant<WT>(wl[i], s_xq + blk * 8, s_xd[blk], __uint_as_float(xs), (int)xs);
    }
    __syncthreads();
    // ---- a lane per row: the terms in block order ----
    if (tid < RU) {
        const int64_t row = (idx0 >> bshift) + tid;
        if (row < M) {
            const float4 *p4 = reinterpret_cast<const float4 *>(s_part + tid * pitch);
            float s = 0.0f;
            for (int q = 0; q < BPR / 4; q++) {
                const float4 t = p4[q];
                s = __fadd_rn(s, t.x); s = __fadd_rn(s, t.y); s = __fadd_rn(s, t.z); s = __fadd_rn(s, t.w);
            }
            j.out[row] = s;
        }
    }
}

}  // namespace bgk
