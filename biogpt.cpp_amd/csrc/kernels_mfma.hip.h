// Block-quantized weights x many activation columns on the matrix cores (gfx950 v_mfma_i32_16x16x32_i8).
//
// With 32+ columns in flight (prompt passes of 128 tokens, dozens of sequences decoded together) the 8-column
// mat-vec kernels of kernels_fast.hip.h are instruction-bound: ~12 VALU instructions per (row, column, block),
// fc2 at 128 columns = 42 us.  One MFMA does the 32-element integer dot of a weight block against an activation
// block for 16 rows x 16 columns at once -- exactly ggml's per-block "sumi" -- and everything after it is the
// unchanged scalar arithmetic:
//
//   per wave = one 16 x 16 output tile, for block b = 0 .. K/32-1 IN ORDER:
//     A = 16 rows x 32 weights of block b      (int8 from the expanded image: Q4_0 / Q5_0 signed, q - 8 / q - 16)
//     B = 32 activations x 16 columns          (the producer's Q8_0 / Q8_1 blocks)
//     C = A x B (int32, zero-initialised)      -> sumi[row][col] of THIS block, 4 per lane
//     acc[row][col] += ggml's per-type term    (sumi - 8*sum(x)) * d_w * d_x  etc. (unit_dot_quant; sumi - 8*sum(x) IS the signed dot)
//
// The accumulation runs over the blocks in block order inside one lane -- the association of the reference's
// scalar vec_dot loop -- and the integer sums are exact: results are bit-identical to the VALU kernels and to
// the oracle.
//
// What makes it fast is where the operands come from (a first version that loaded them straight from the SoA
// weight arrays was bound by the texture addresser: every 8-byte operand load touched 16 cache lines):
//   * weights: a second, ROW-TILED copy of the matrix (retile_kernel, built once on first use): the 16 rows of
//     a tile are contiguous per block, so an A-operand load of a wave covers 256 contiguous bytes and the four
//     scales a lane needs are 8 contiguous bytes;
//   * activations: the 16 columns of the workgroup are staged in LDS with coalesced 16-byte loads (up to 2048
//     elements of K at a time: K = 4096 goes in two phases, 41 KB, three workgroups per compute unit) and shared
//     by its 4 waves (= 4 row tiles); the row pitch + 16 bytes makes the 8-byte operand reads conflict-free
//     (4 lanes per bank pair, the minimum for 512 bytes).
// Workgroup = 4 waves = 64 rows x 16 columns; grid = (M/64, ceil(N/16)).  A loads are batched 4 blocks at a
// time and double-buffered against the arithmetic.
#pragma once

#include "kernels_fast.hip.h"

namespace bgk {

using i32x4 = __attribute__((ext_vector_type(4))) int;

// weight batches (4 blocks each) in flight per wave: 2 (round 2) or 3 (round 4 experiment, profiles/prefill_mfma_depth_r4.txt)
#ifndef MFMA_DEPTH
#define MFMA_DEPTH 2
#endif

// ---- row-tiled, EXPANDED weight image (round 4) ---------------------------------------------------------------
// src (SoA arena): qs[(row*BPR + b) * QB], sc[(row*BPR + b)], qh[(row*BPR + b)]
// dst (image)    : index i = (tile*BPR + b)*16 + r  with tile = row / 16, r = row % 16   (M is a multiple of 16):
//                  q[i * 32 .. +31]  the block's 32 weights as int8 in element order -- Q4_0: q - 8, Q5_0: q - 16 (signed: the MFMA's integer dot is then
//                                    sum_j (q_j - 8) x_j itself, no block sum of the activations is needed), Q4_1 / Q5_1: q (0 .. 15 / 0 .. 31), Q8_0: as stored;
//                  s[i]              the block's scale as f32 (Q4_1 / Q5_1: {d, m} as two f32)
// Round 2's image kept the file's nibbles and fp16 scales (16 + 2 bytes per block) and every MFMA was followed by ~35 VALU instructions per lane, of which the
// operand unpack, the fp16 -> f32 conversions of four row scales and the "- 8 sum(x)" corrections were half; they are done once, here.  The image doubles (Q4: 36 bytes
// per block instead of 18): 340 MB for BioGPT-base, read once per prompt pass.
template <int WT>
__global__ __launch_bounds__(256) void retile_kernel(DevMatrix src, uint8_t *dq, uint8_t *ds) {
    using TI = TypeInfo<WT>;
    constexpr int QB = TI::qbytes;
    const int BPR = src.K / QK;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // source block index row*BPR + b
    if (idx >= (int64_t)src.M * BPR) return;
    const int row = (int)(idx / BPR), b = (int)(idx - (int64_t)row * BPR);
    const int64_t dst = ((int64_t)(row >> 4) * BPR + b) * 16 + (row & 15);
    const uint4 *q = reinterpret_cast<const uint4 *>(src.qs + idx * QB);
    uint4 *o = reinterpret_cast<uint4 *>(dq + dst * 32);
    if (WT == W_Q8_0) { o[0] = q[0]; o[1] = q[1]; }
    else {
        const uint4 v = q[0];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t lo[4], hi[4];
        uint32_t qh = 0u;
        if (WT == W_Q5_0 || WT == W_Q5_1) qh = src.qh[idx];
#pragma unroll
        for (int i = 0; i < 4; i++) {      // byte j of word i: low nibble = element 4 i + j, high nibble = element 16 + 4 i + j (expand_unit's unpacking)
            lo[i] = w[i] & 0x0F0F0F0Fu;
            hi[i] = (w[i] >> 4) & 0x0F0F0F0Fu;
            if (WT == W_Q5_0 || WT == W_Q5_1) { lo[i] |= spread4(qh >> (4 * i)); hi[i] |= spread4(qh >> (16 + 4 * i)); }
            if (WT == W_Q4_0) { lo[i] = (lo[i] + 0x78787878u) ^ 0x80808080u; hi[i] = (hi[i] + 0x78787878u) ^ 0x80808080u; }
            if (WT == W_Q5_0) { lo[i] = (lo[i] + 0x70707070u) ^ 0x80808080u; hi[i] = (hi[i] + 0x70707070u) ^ 0x80808080u; }
        }
        o[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        o[1] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    }
    if (TI::q81) {
        const uint32_t dm = reinterpret_cast<const uint32_t *>(src.sc)[idx];
        reinterpret_cast<float2 *>(ds)[dst] = make_float2(h2f((uint16_t)(dm & 0xFFFFu)), h2f((uint16_t)(dm >> 16)));
    } else {
        reinterpret_cast<float *>(ds)[dst] = h2f(reinterpret_cast<const uint16_t *>(src.sc)[idx]);
    }
}

// ggml's block term from the integer dot (the expressions of unit_dot_quant; the symmetric formats' dots come out of the signed image already corrected)
template <int WT>
__device__ __forceinline__ float mfma_block_term(int dot, float dw, float mw, float xd, uint32_t xs) {
    if (WT == W_Q8_0) return __fmul_rn((float)dot, __fmul_rn(dw, xd));
    if (WT == W_Q4_0) return __fmul_rn(__fmul_rn((float)dot, dw), xd);
    if (WT == W_Q5_0) return __fmul_rn(__fmul_rn(dw, xd), (float)dot);
    return __fadd_rn(__fmul_rn(__fmul_rn(dw, xd), (float)dot), __fmul_rn(mw, __uint_as_float(xs)));
}

template <int WT>
struct MfmaBatch {                       // one lane's weight-side operands for CH consecutive blocks
    static constexpr int CH = 4;         // blocks per load batch (8 costs ~60 more VGPRs and an occupancy step)
    static constexpr int SW = TypeInfo<WT>::q81 ? 8 : 4;   // dwords of the lane's 4 row scales per block (f32 d, or {d, m})
    uint2 q[CH];                         // the A operand itself: 8 int8 of row (lane & 15), elements 8 g .. 8 g + 7
    uint32_t sc[CH][SW];                 // scales of rows 4g .. 4g+3
};

// this lane's two image pointers at block 0 of its tile; block b is a CONSTANT stride further (512 bytes of weights, 64 / 128 bytes of scales),
// so a batch is one base address plus immediate offsets
template <int WT>
struct MfmaLanePtrs {
    const uint8_t *q, *sc;
};
template <int WT>
__device__ __forceinline__ MfmaLanePtrs<WT> mfma_lane_ptrs(const DevMatrix &img, int64_t base, int li, int g) {
    using TI = TypeInfo<WT>;
    MfmaLanePtrs<WT> p;
    p.q = img.qs + (base + li) * 32 + 8 * g;
    p.sc = img.sc + (base + 4 * g) * (TI::q81 ? 8 : 4);
    return p;
}
template <int WT>
__device__ __forceinline__ void mfma_load_batch(MfmaBatch<WT> &t, const MfmaLanePtrs<WT> &lp, int b0) {
    using TI = TypeInfo<WT>;
    constexpr int QS = 512, SS = TI::q81 ? 128 : 64;   // bytes per block of one tile
    const uint8_t *q = lp.q + (size_t)b0 * QS, *sc = lp.sc + (size_t)b0 * SS;
#pragma unroll
    for (int j = 0; j < MfmaBatch<WT>::CH; j++) {
        t.q[j] = *reinterpret_cast<const uint2 *>(q + j * QS);
        const uint4 v = *reinterpret_cast<const uint4 *>(sc + j * SS);
        t.sc[j][0] = v.x; t.sc[j][1] = v.y; t.sc[j][2] = v.z; t.sc[j][3] = v.w;
        if (TI::q81) {
            const uint4 w = *reinterpret_cast<const uint4 *>(sc + j * SS + 16);
            t.sc[j][4] = w.x; t.sc[j][5] = w.y; t.sc[j][6] = w.z; t.sc[j][7] = w.w;
        }
    }
}

template <int WT>
__device__ __forceinline__ long mfma_a_operand(const MfmaBatch<WT> &t, int j, int g) {
    return (long)(((unsigned long)t.q[j].y << 32) | t.q[j].x);
}

// NT = 16-column tiles per wave.  NT = 2: the weight-side work of a block (operand unpack, the four fp16 row scales) feeds
// two MFMAs and two sets of per-output terms, and the two tiles' dependency chains interleave.
__host__ __device__ inline int matmul_mfma_kp(int K, int nt) { const int cap = nt == 1 ? 2048 : 1024; return K > cap ? cap : K; }
__host__ __device__ inline size_t matmul_mfma_smem_bytes(int K, bool gelu_q8, int nt = 1) {
    const int kp = matmul_mfma_kp(K, nt), nc = 16 * nt;                        // staged K phase, columns per workgroup
    const size_t act = (size_t)nc * (kp + 16) + 2 * (size_t)nc * (kp / QK + 1) * 4;
    const size_t tail = gelu_q8 ? (size_t)nc * 64 * 4 : 0;
    return (act > tail ? act : tail) + 64;    // the GELU_Q8 exchange reuses the activation area after the last block
}

template <int WT, int EPI, int K, int NT = 1>
__global__ __launch_bounds__(256) void matmul_mfma_kernel(const MatvecParams p, const DevMatrix img) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant, "block-quantized weights");
    static_assert(NT == 1 || NT == 2, "column tiles per wave");
    // KP: columns of K staged in LDS at a time (K = 4096 goes in phases: 41 KB instead of 82 KB, so three workgroups
    // fit a compute unit instead of one; the accumulators simply carry over, block order is unchanged)
    constexpr int KP = (NT == 1) ? (K > 2048 ? 2048 : K) : (K > 1024 ? 1024 : K), NPH = K / KP, BPP = KP / QK, NC = 16 * NT;
    constexpr int CH = MfmaBatch<WT>::CH, BPR = K / QK, NB = BPP / CH, PITCH = KP + 16, SP = BPP + 1;   // SP: per-column pitch of the scale arrays (bank skew)
    static_assert(NB % 2 == 0, "K must be a multiple of 512");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint8_t *const s_q = smem_raw;                                              // [NC columns][PITCH] int8
    float *const s_d = reinterpret_cast<float *>(smem_raw + NC * PITCH);        // [NC][SP] activation block scales
    uint32_t *const s_s = reinterpret_cast<uint32_t *>(s_d + NC * SP);          // [NC][SP] block sums (Q8_0: int, Q8_1: d*sum)
    float *const s_tail = reinterpret_cast<float *>(smem_raw);                  // GELU_Q8: [NC columns][64 rows], after the loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int M = p.W.M;
    const int tile = blockIdx.x * 4 + wave;                                     // 16-row tile of this wave
    const int row0 = tile * 16, col0 = blockIdx.y * NC;
    const bool tile_ok = row0 < M;
    const int tile_c = tile_ok ? tile : 0;                                      // waves past the last row keep the barriers company
    const int64_t base = (int64_t)tile_c * BPR * 16;
    const int orow = row0 + 4 * g;                                              // outputs: rows 4g .. 4g+3, columns col0 + 16t + (lane & 15)

    const MfmaLanePtrs<WT> lp = mfma_lane_ptrs<WT>(img, base, li, g);
    MfmaBatch<WT> t0, t1;
#if MFMA_DEPTH == 3
    MfmaBatch<WT> t2;
#endif
    mfma_load_batch<WT>(t0, lp, 0);
    // epilogue inputs (independent loads); M is a multiple of 4 everywhere
    const int orc = min(orow, M - 4);
    float4 e_bias = make_float4(0.f, 0.f, 0.f, 0.f), e_res[NT];
    if (EPI != EPI_LOGITS) e_bias = *reinterpret_cast<const float4 *>(p.bias + orc);
    int e_npast[NT], e_seq[NT], colv[NT];
    bool col_ok[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        colv[t] = col0 + 16 * t + li;
        col_ok[t] = colv[t] < p.N;
        const int colc = min(colv[t], p.N - 1);
        e_res[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI == EPI_RESID) e_res[t] = *reinterpret_cast<const float4 *>(p.resid + (size_t)colc * p.ldr + orc);
        e_npast[t] = 0; e_seq[t] = 0;
        if (EPI == EPI_QKV) {
            e_npast[t] = p.seq ? p.seq[colc].n_past : p.st->n_past + colc;
            e_seq[t] = (p.seq && p.col_mode) ? p.seq[colc].seq_id : colc;
        }
    }

    const uint8_t *bq = s_q + li * PITCH + 8 * g;                               // B operand: column = lane & 15 (+ 16 t), k-group g
    const float *bd = s_d + li * SP;
    const uint32_t *bs = s_s + li * SP;
    float acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[t][r] = 0.0f;
    auto consume = [&](const MfmaBatch<WT> &tb, int b0) {                       // b0: block index inside the staged phase
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int b = b0 + j;
            const long aop = mfma_a_operand<WT>(tb, j, g);
            float dwr[4], mwr[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                dwr[r] = __uint_as_float(TI::q81 ? tb.sc[j][2 * r] : tb.sc[j][r]);
                mwr[r] = TI::q81 ? __uint_as_float(tb.sc[j][2 * r + 1]) : 0.0f;
            }
            i32x4 c[NT];
            float xd[NT];
            uint32_t xs[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const long bop = *reinterpret_cast<const long *>(bq + t * 16 * PITCH + b * QK);
                xd[t] = bd[t * 16 * SP + b];
                xs[t] = TI::q81 ? bs[t * 16 * SP + b] : 0u;      // the block sums only serve the min term of Q4_1 / Q5_1
                const i32x4 zero = {0, 0, 0, 0};
                c[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(aop, bop, zero, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[t][r] = __fadd_rn(acc[t][r], mfma_block_term<WT>(c[t][r], dwr[r], mwr[r], xd[t], xs[t]));
        }
    };
#pragma unroll 1
    for (int ph = 0; ph < NPH; ph++) {
        // ---- stage this phase of the NC activation columns in LDS (coalesced 16-byte pieces) ----
        if (ph > 0) __syncthreads();                                            // the previous phase has been consumed
        {
            // The activation bytes go from global memory STRAIGHT into LDS (global_load_lds_dwordx4, gfx950: 16 bytes per lane, a wave-instruction moves 1 KB of ONE
            // column to 1 KB of that column's LDS row): no registers, every piece of the phase in flight at once.  Round 4: written as "load 16 bytes, store them to
            // LDS" the compiler waited for each piece before it asked for the next (s_waitcnt vmcnt(0) between them, whatever the source order: it sinks every
            // load to its LDS write) -- nine dependent L2 round trips per phase, a third of fc2's 26 us (profiles/prefill_mfma_staging_r4.txt).
            constexpr int PPC = KP / 16;                                        // 16-byte pieces per column
            constexpr int NPIECE = NC * PPC / 256, NSC = (NC * BPP + 255) / 256;
            static_assert(PPC % 64 == 0, "a wave's 64 pieces lie in one column");
            typedef __attribute__((address_space(1))) const void gl_ptr;
            typedef __attribute__((address_space(3))) void lds_ptr;
            const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
            for (int i = 0; i < NPIECE; i++) {
                const int pc0 = 64 * wv + 256 * i, c = pc0 / PPC, o0 = (pc0 - c * PPC) * 16;      // wave-uniform: the lane's piece is pc0 + lane
                const int cc = min(col0 + c, p.N - 1);                          // idle columns re-read the last one
                __builtin_amdgcn_global_load_lds((gl_ptr *)(p.aq_q + (size_t)cc * K + ph * KP + o0 + lane * 16), (lds_ptr *)(s_q + c * PITCH + o0), 16, 0, 0);
            }
            float sd[NSC];
            uint32_t ss[NSC];
#pragma unroll
            for (int i = 0; i < NSC; i++) {
                const int e = min(tid + 256 * i, NC * BPP - 1);
                const int c = e / BPP, b = e - c * BPP;
                const int cc = min(col0 + c, p.N - 1);
                sd[i] = p.aq_d[(size_t)cc * BPR + ph * BPP + b];
                ss[i] = TI::q81 ? p.aq_s[(size_t)cc * BPR + ph * BPP + b] : 0u;
            }
#pragma unroll
            for (int i = 0; i < NSC; i++) {
                const int e = tid + 256 * i;
                if (e < NC * BPP) {
                    const int c = e / BPP, b = e - c * BPP;
                    s_d[c * SP + b] = sd[i];
                    if (TI::q81) s_s[c * SP + b] = ss[i];
                }
            }
        }
        if (ph > 0) mfma_load_batch<WT>(t0, lp, ph * BPP);
        __builtin_amdgcn_s_waitcnt(0x0f70);                                     // vmcnt(0): the direct-to-LDS loads of this wave have landed (the barrier publishes them)
        __syncthreads();
#if MFMA_DEPTH == 3
        // three weight batches in flight: a batch is requested two consume steps (8 blocks) before its MFMAs
        mfma_load_batch<WT>(t1, lp, ph * BPP + CH);
#pragma unroll 1
        for (int nb = 0; nb < NB; nb += 3) {
            if (nb + 2 < NB) mfma_load_batch<WT>(t2, lp, ph * BPP + (nb + 2) * CH);
            consume(t0, nb * CH);
            if (nb + 1 < NB) {
                if (nb + 3 < NB) mfma_load_batch<WT>(t0, lp, ph * BPP + (nb + 3) * CH);
                consume(t1, (nb + 1) * CH);
            }
            if (nb + 2 < NB) {
                if (nb + 4 < NB) mfma_load_batch<WT>(t1, lp, ph * BPP + (nb + 4) * CH);
                consume(t2, (nb + 2) * CH);
            }
        }
#else
#pragma unroll 1
        for (int nb = 0; nb < NB; nb += 2) {
            mfma_load_batch<WT>(t1, lp, ph * BPP + (nb + 1) * CH);
            consume(t0, nb * CH);
            if (nb + 2 < NB) mfma_load_batch<WT>(t0, lp, ph * BPP + (nb + 2) * CH);
            consume(t1, (nb + 1) * CH);
        }
#endif
    }

    if (EPI == EPI_GELU_Q8) {
        __syncthreads();                                                        // everyone is done reading the activation area
#pragma unroll
        for (int t = 0; t < NT; t++) {
            float4 v;
            v.x = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias.x, acc[t][0]))]); v.y = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias.y, acc[t][1]))]);
            v.z = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias.z, acc[t][2]))]); v.w = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias.w, acc[t][3]))]);
            *reinterpret_cast<float4 *>(s_tail + (16 * t + li) * 64 + wave * 16 + 4 * g) = v;
        }
        __syncthreads();
        // the workgroup's 64 rows are two Q8 blocks of fc2's activation row per column: quantize_row_q8_0 / _q8_1,
        // a half-wave per (column, block)
        for (int u = wave * 2 + (lane >> 5); u < 2 * NC; u += 8) {
            const int c = u >> 1, half = u & 1;
            if (col0 + c >= p.N) continue;
            const float v1 = s_tail[c * 64 + half * 32 + (lane & 31)];
            float amax = fabsf(v1);
            amax = fmaxf(amax, dpp_f<DPP_QUAD_XOR1>(amax)); amax = fmaxf(amax, dpp_f<DPP_QUAD_XOR2>(amax));
            amax = fmaxf(amax, dpp_f<DPP_ROW_HALF_MIRROR>(amax)); amax = fmaxf(amax, dpp_f<DPP_ROW_MIRROR>(amax));
            amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
            const float d = amax / 127.0f;
            const float id = (d != 0.0f) ? 1.0f / d : 0.0f;
            const int q = (int)roundf(__fmul_rn(v1, id));
            int isum = q;
            isum += dpp_i<DPP_QUAD_XOR1>(isum); isum += dpp_i<DPP_QUAD_XOR2>(isum);
            isum += dpp_i<DPP_ROW_HALF_MIRROR>(isum); isum += dpp_i<DPP_ROW_MIRROR>(isum);
            isum += __shfl_xor(isum, 16, 64);
            const size_t blk = (size_t)(col0 + c) * (M / 32) + blockIdx.x * 2 + half;   // column-major [N][d_ff/32]
            p.oq_q[blk * 32 + (lane & 31)] = (int8_t)q;
            if ((lane & 31) == 0) {
                if (TI::q81) { p.oq_d[blk] = d; p.oq_s[blk] = __float_as_uint(__fmul_rn((float)isum, d)); }
                else { p.oq_d[blk] = h2f(f2h(d)); p.oq_s[blk] = (uint32_t)isum; }
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (!(col_ok[t] && tile_ok)) continue;
        const int col = colv[t];
        if (EPI == EPI_QKV) {
            float4 v;
            v.x = __fadd_rn(e_bias.x, acc[t][0]); v.y = __fadd_rn(e_bias.y, acc[t][1]); v.z = __fadd_rn(e_bias.z, acc[t][2]); v.w = __fadd_rn(e_bias.w, acc[t][3]);
            const int which = orow / K, rr = orow - which * K;             // d_model == K for the q/k/v projection
            if (which == 0) {
                v.x = __fmul_rn(v.x, p.q_scale); v.y = __fmul_rn(v.y, p.q_scale); v.z = __fmul_rn(v.z, p.q_scale); v.w = __fmul_rn(v.w, p.q_scale);
                *reinterpret_cast<float4 *>(p.q_out + (size_t)col * K + rr) = v;
            } else {
                float *cache = ((which == 1) ? p.kcache : p.vcache) + (p.seq ? (size_t)e_seq[t] * p.kv_seq_stride : 0);
                const int hh = rr >> p.dk_log2, dd = rr & (p.dk - 1);       // head-major cache: [H][P][dk]; 4 | dk
                *reinterpret_cast<float4 *>(cache + (((size_t)hh * p.P + e_npast[t]) << p.dk_log2) + dd) = v;
            }
        } else if (EPI == EPI_RESID) {
            float4 v;
            v.x = __fadd_rn(__fadd_rn(acc[t][0], e_bias.x), e_res[t].x); v.y = __fadd_rn(__fadd_rn(acc[t][1], e_bias.y), e_res[t].y);
            v.z = __fadd_rn(__fadd_rn(acc[t][2], e_bias.z), e_res[t].z); v.w = __fadd_rn(__fadd_rn(acc[t][3], e_bias.w), e_res[t].w);
            *reinterpret_cast<float4 *>(p.out + (size_t)col * p.ldo + orow) = v;
        } else {
            *reinterpret_cast<float4 *>(p.out + (size_t)col * p.ldo + orow) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        }
    }
}

}  // namespace bgk
