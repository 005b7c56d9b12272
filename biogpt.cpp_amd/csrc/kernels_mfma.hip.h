// Block-quantized weights x many activation columns on the matrix cores (gfx950 v_mfma_i32_16x16x32_i8).
//
// With 48+ columns in flight (prompt passes, dozens of sequences decoded together) the 8-column mat-vec kernels of
// kernels_fast.hip.h are instruction-bound.  One MFMA does the 32-element integer dot of a weight block against an
// activation block for 16 rows x 16 columns at once -- exactly ggml's per-block "sumi" -- and everything after it is
// the unchanged scalar arithmetic:
//
//   per wave = one 16 x 16 output tile, for block b = 0 .. K/32-1 IN ORDER:
//     A = 16 rows x 32 weights of block b      (int8 from the expanded image: Q4_0 / Q5_0 signed, q - 8 / q - 16)
//     B = 32 activations x 16 columns          (the producer's Q8_0 / Q8_1 blocks)
//     C = A x B (int32, zero-initialised)      -> sumi[row][col] of THIS block, 4 per lane
//     acc[row][col] += ggml's per-type term    (mfma_block_term: the reference's mul / mul / add, unfused)
//
// The accumulation runs over the blocks in block order inside one lane -- the association of the reference's scalar
// vec_dot loop -- and the integer sums are exact: results are bit-identical to the VALU kernels and to the oracle.
//
// Round 5: the loop is a SOFTWARE PIPELINE.  Across blocks the only true dependence is the one f32 add per output; a
// 1024-row matrix at 512 columns is 2048 wave tiles = two waves per SIMD, so nothing but the wave itself can hide its
// latencies.  Round 4's loop (LDS read -> MFMA -> cvt -> mul -> mul -> add, batch by batch) ran ~340 cycles per block
// against ~40 of arithmetic.  Now, per batch of 4 blocks, three stages are in flight in one wave:
//     global A-operand loads        3 batches ahead of their MFMAs  (L2-hit latency ~200-330 cycles),
//     LDS reads (B operand, scales) 1 batch ahead of their use,
//     the 4 MFMAs of batch n+1      issued among the cvt/mul/mul/add of batch n.
// and what a lane loads per block shrank: the four row scales a lane needs used to be a 16-byte GLOBAL load per lane
// (1 KB through the texture addresser per wave and block for 64 unique bytes -- twice the weight bytes themselves);
// they now go global -> LDS by DMA once per phase (each wave its own tile's 2 KB) and are read back as one
// ds_read_b128 broadcast.  The activation block scales ride the same DMA.
//
// Operand sources:
//   * weights: a second, ROW-TILED, EXPANDED copy of the matrix (retile_kernel, built once on first use): the 16 rows
//     of a tile are contiguous per block, so an A-operand load of a wave covers 512 contiguous bytes;
//   * activations: the 16 columns of the workgroup are staged in LDS by global_load_lds (no registers), 1024 elements
//     of K per phase, DOUBLE-BUFFERED: phase ph + 1 lands while phase ph is consumed (K = 4096: four phases, 54 KB;
//     K = 1024: one phase, 27 KB); the row pitch + 16 bytes makes the 8-byte operand reads conflict-free.
// Workgroup = 4 waves = 64 rows x 16 columns; grid = (M/64, ceil(N/16)); linear workgroup id mod 8 = blockIdx.x mod 8
// (M/64 is a multiple of 8 for every BioGPT matrix but lm_head), so an XCD's L2 holds 1/8 of the weight image.
#pragma once

#include "kernels_fast.hip.h"
#include <type_traits>

namespace bgk {

using i32x4 = __attribute__((ext_vector_type(4))) int;
typedef float mm_f2 __attribute__((ext_vector_type(2)));

// ---- row-tiled, EXPANDED weight image -------------------------------------------------------------------------
// src (SoA arena): qs[(row*BPR + b) * QB], sc[(row*BPR + b)], qh[(row*BPR + b)]
// dst (image)    : (tile, b, r) with tile = row / 16, r = row % 16   (M is a multiple of 16), i = (tile*BPR + b)*16 + r:
//                  q[i * 32 .. +31]   the block's 32 weights as int8 in element order -- Q4_0: q - 8, Q5_0: q - 16 (signed: the MFMA's integer dot is then
//                                     sum_j (q_j - 8) x_j itself, no block sum of the activations is needed), Q4_1 / Q5_1: q (0 .. 15 / 0 .. 31), Q8_0: as stored;
//                  scales             f32, per (tile, b) one 64-byte group d[16 rows]; Q4_1 / Q5_1: 128 bytes, d[16 rows] then m[16 rows]
//                                     (a lane's rows 4g .. 4g+3 are one aligned 16-byte piece of each)
template <int WT>
__global__ __launch_bounds__(256) void retile_kernel(DevMatrix src, uint8_t *dq, uint8_t *ds) {
    using TI = TypeInfo<WT>;
    constexpr int QB = TI::qbytes;
    const int BPR = src.K / QK;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // source block index row*BPR + b
    if (idx >= (int64_t)src.M * BPR) return;
    const int row = (int)(idx / BPR), b = (int)(idx - (int64_t)row * BPR);
    const int64_t grp = (int64_t)(row >> 4) * BPR + b;                 // (tile, b)
    const int64_t dst = grp * 16 + (row & 15);
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
        float *const s = reinterpret_cast<float *>(ds) + grp * 32 + (row & 15);
        s[0] = h2f((uint16_t)(dm & 0xFFFFu));
        s[16] = h2f((uint16_t)(dm >> 16));
    } else {
        reinterpret_cast<float *>(ds)[dst] = h2f(reinterpret_cast<const uint16_t *>(src.sc)[idx]);
    }
}

// ggml's block term from the integer dot (the expressions of unit_dot_quant; the symmetric formats' dots come out of the signed image already corrected)
template <int WT>
__device__ __forceinline__ float mfma_block_term(float dot, float dw, float mw, float xd, float xs) {    // dot: the integer dot, converted (exact)
    if (WT == W_Q8_0) return __fmul_rn(dot, __fmul_rn(dw, xd));
    if (WT == W_Q4_0) return __fmul_rn(__fmul_rn(dot, dw), xd);
    if (WT == W_Q5_0) return __fmul_rn(__fmul_rn(dw, xd), dot);
    return __fadd_rn(__fmul_rn(__fmul_rn(dw, xd), dot), __fmul_rn(mw, xs));
}

// LDS of one staged phase of the workgroup (KP = 1024 elements of K for 16 columns and the 4 waves' 16-row tiles)
template <bool Q81, int J = 1>
struct MfmaLds {
    static constexpr int KP = 1024, BPP = KP / QK;
    static constexpr int PITCH = KP + 16;                    // bytes per activation column: the 8-byte operand reads of 32 lanes cover all 64 banks
    static constexpr int SWF = Q81 ? 32 : 16;                // floats per (tile, block) of weight scales
    static constexpr int OFF_Q = 0;
    static constexpr int OFF_D = OFF_Q + 16 * PITCH;         // [8 batches][16 columns][4 blocks] activation block scales d: a batch's read is 256 contiguous bytes
    static constexpr int OFF_S = OFF_D + 16 * BPP * 4;       // the same for the Q8_1 block sums d * sum(q) (Q4_1 / Q5_1 only)
    static constexpr int OFF_W = OFF_S + (Q81 ? 16 * BPP * 4 : 0);   // [4 waves][J tiles][BPP][SWF] weight scales of the waves' tiles
    static constexpr int BYTES = OFF_W + 4 * J * BPP * SWF * 4;
};
__host__ __device__ inline size_t matmul_mfma_smem_bytes(int K, bool q81, bool gelu_q8, int J = 1) {
    const size_t buf = J == 2 ? (q81 ? (size_t)MfmaLds<true, 2>::BYTES : (size_t)MfmaLds<false, 2>::BYTES) : (q81 ? (size_t)MfmaLds<true>::BYTES : (size_t)MfmaLds<false>::BYTES);
    const size_t act = buf * (K > 1024 ? 2 : 1);
    const size_t tail = gelu_q8 ? (size_t)16 * 64 * 4 : 0;   // the GELU_Q8 exchange reuses the area after the last block
    return (act > tail ? act : tail);
}
// K = 1024 is one phase: 4 computing waves.  Longer rows add the staging wave (phases 1 .. are requested while the computing waves consume the one before).
__host__ __device__ constexpr int mfma_threads(int K) { return K > 1024 ? 320 : 256; }

// MFMA_STAMPS (tools/microbench22.hip only): shader-clock stamps per wave at four points (0 entry, 1 phase-0 DMAs issued, 2 loop done, 3 epilogue done), p.tstamp[(workgroup * 5 + wave) * 8 + k]; k >= 4: the 100 MHz wall clock (4 entry, 5 loop done, 7 epilogue done; 6: XCC_ID / HW_ID)
#ifdef MFMA_STAMPS
#define MFMA_STAMP(k) do { if (lane == 0) p.tstamp[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 5 + wv) * 8 + (k)] = (k) >= 4 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MFMA_STAMP(k) do { } while (0)
#endif

typedef __attribute__((address_space(1))) const void mfma_gl_ptr;
typedef __attribute__((address_space(3))) void mfma_lds_ptr;

// registers: 4 waves per SIMD (<= 128) for the formats without a min term -- at 136 the second workgroup of a compute unit only started when the first one's staging wave
// had left (5-wave workgroups at 3 waves per SIMD do not pack: fc2 18 us instead of 10, tools/microbench22.hip); Q4_1 / Q5_1 hold twice the scales: 3 waves per SIMD (<= 168)
// J = 2 (K = 1024 only; the host asks for it when 32 | M): a wave WALKS two row tiles (rows 32 w' .. 32 w' + 31 of the matrix) against the workgroup's 16 stationary activation
// columns -- one prologue (arguments, the activations' DMA flight: ~4000 cycles before the first MFMA) and one dispatch for 14000 cycles of loop instead of 7000; fc1's 2048
// workgroups (two rounds of four per compute unit, each with its own prologue, epilogue and 3 us until the freed place is taken) become 1024 = ONE round.  The tiles of an image
// are contiguous, so the A-operand queue simply runs on across the tile boundary; both tiles' weight scales are staged in phase 0 (4 KB per wave); no barrier after phase 0.
// A wave's two tiles are one 32-row Q8 block of fc2's activation: the GELU_Q8 epilogue is wave-local (no LDS exchange, no barrier).
template <int WT, int EPI, int K, int J = 1>
__global__ __launch_bounds__(mfma_threads(K), TypeInfo<WT>::q81 ? 3 : 4) void matmul_mfma_kernel(const MatvecParams p, const DevMatrix img) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant, "block-quantized weights");
    static_assert(J == 1 || (J == 2 && K == 1024 && !TI::q81 && (EPI == EPI_GELU_Q8 || EPI == EPI_QKV)), "two tiles per wave: one-phase rows, the fc1 and q/k/v sites, the formats without a min term (the others do not fit their register budget without scratch -- and any scratch at all costs these launches their dispatch rate)");
    // q/k/v walking two tiles: its epilogue inputs (two bias quads, the column's cache row) are requested BEHIND the loop -- inside it they cost the registers the second tile's
    // parked sums took (18 - 31 spilled registers when they were requested in the last steps as in the one-tile kernel); a workgroup waits one L2 round trip for them, once
    constexpr bool EPI_LATE = J == 2 && EPI == EPI_QKV;
    constexpr bool Q81 = TI::q81;
    using L = MfmaLds<Q81, J>;
    constexpr int KP = L::KP, NPH = K / KP, BPP = L::BPP, BPR = K / QK, CH = 4, NB = BPP / CH, NBT = NPH * NB, NBL = NB * J;
    constexpr int PITCH = L::PITCH, SWF = L::SWF;
    constexpr int QDEPTH = (J == 2 && (WT != W_Q4_0 || EPI == EPI_QKV)) ? 2 : 3; // A-operand batches requested ahead of their MFMAs (the walking kernels only fit 128 registers without scratch at depth 2 -- all but Q4_0's fc1)
    constexpr int NW = mfma_threads(K) / 64;                 // waves of the workgroup
    static_assert(K % KP == 0 && NB == 8, "phases of 1024 elements, 8 batches of 4 blocks");
    static_assert(NBT > QDEPTH, "the prologue requests QDEPTH batches");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *const s_tail = reinterpret_cast<float *>(smem_raw);                  // GELU_Q8: [16 columns][64 rows], after the loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int M = p.W.M;
    const int col0 = blockIdx.y * 16;
    MFMA_STAMP(0); MFMA_STAMP(4);
#ifdef MFMA_STAMPS
    if (lane == 0) p.tstamp[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 5 + wv) * 8 + 6] = ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));   // XCC_ID, HW_ID
#endif

    // ---- staging: a phase of the workgroup goes global -> LDS by DMA (global_load_lds: no registers), in NITEMS wave-instructions of 1 KB ----
    //   items 0 .. 15: the 16 activation columns (1 KB of int8 each); then the four tiles' weight scales (2 / 4 KB contiguous in the image each);
    //   then the activation block scales (and Q8_1 sums): 16-byte piece (batch j, column c) to [j][c], a lane per piece
    // A wave issues a DMA every ~130 cycles (microbench22: 26 of them 3000 - 4300 cycles, as long as a phase's arithmetic), so phase 0 -- which nothing overlaps --
    // is requested by ALL waves, an item each in turn.
    constexpr int NI_W = 4 * J * (SWF / 8), NI_X = Q81 ? 4 : 2, NITEMS = 16 + NI_W + NI_X;
    const int ntile = M / 16;
    auto stage_item = [&](int ph, int it, int lane) {
        unsigned char *const base = smem_raw + (ph & 1) * L::BYTES;
        if (it < 16) {
            const int cc = min(col0 + it, p.N - 1);                             // idle columns re-read the last one
            __builtin_amdgcn_global_load_lds((mfma_gl_ptr *)(p.aq_q + (size_t)cc * K + ph * KP + lane * 16), (mfma_lds_ptr *)(base + L::OFF_Q + it * PITCH), 16, 0, 0);
        } else if (it < 16 + NI_W) {
            const int w = (it - 16) / (J * (SWF / 8)), i = (it - 16) % (J * (SWF / 8));
            const int t = min(((int)blockIdx.x * 4 + w) * J, ntile - J);        // (J = 2: the wave's two tiles are contiguous in the image, one piece of 2 x 2 KB)
            const uint8_t *const src = img.sc + ((size_t)t * BPR + (size_t)ph * BPP) * (SWF * 4) + lane * 16;
            __builtin_amdgcn_global_load_lds((mfma_gl_ptr *)(src + i * 1024), (mfma_lds_ptr *)(base + L::OFF_W + w * (J * BPP * SWF * 4) + i * 1024), 16, 0, 0);
        } else {
            const int x = it - 16 - NI_W, i = x & 1;
            const int pos = lane + 64 * i, j = pos >> 4, c = pos & 15;
            const int cc = min(col0 + c, p.N - 1);
            if (x < 2) __builtin_amdgcn_global_load_lds((mfma_gl_ptr *)(p.aq_d + (size_t)cc * BPR + ph * BPP + 4 * j), (mfma_lds_ptr *)(base + L::OFF_D + i * 1024), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((mfma_gl_ptr *)(p.aq_s + (size_t)cc * BPR + ph * BPP + 4 * j), (mfma_lds_ptr *)(base + L::OFF_S + i * 1024), 16, 0, 0);
        }
    };
#pragma unroll 1
    for (int it = wv; it < NITEMS; it += NW) stage_item(0, it, lane);
    MFMA_STAMP(1);

    // ---- the staging wave (K > 1024): phases 1 .. NPH - 1, one ahead of the computing waves ----
    // Barrier ph (one per phase, all waves): a wave arrives when ITS DMAs of phase ph have landed and its LDS reads of phase ph - 1 are done; phase ph + 1 is
    // requested right behind it into the buffer phase ph - 1 occupied.  The DMAs of the later phases live in a wave of their own because every wait on vmcnt is in
    // order: inside a computing wave a wait for an A operand would also wait for whatever DMA was issued before it (and the compiler, seeing LDS written by VMEM,
    // drains vmcnt before the next ds_read of any address).
    if (NW == 5 && wv == 4) {
        int ls = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(ls));                 // (the compiler lays this block out BEHIND the computing waves' code and would hold the lane index over all of it)
#pragma unroll 1
        for (int ph = 0; ph < NPH; ph++) {
            __builtin_amdgcn_s_waitcnt(0x0f70);                                 // vmcnt(0): landed
            if (ph == 0) MFMA_STAMP(2);
            __builtin_amdgcn_s_barrier();
            if (ph + 1 < NPH) {
#pragma unroll
                for (int it = 0; it < NITEMS; it++) stage_item(ph + 1, it, ls);
            }
        }
        MFMA_STAMP(3); MFMA_STAMP(5);
        return;
    }

    const int li = lane & 15, g = lane >> 4;
    const int tile = (blockIdx.x * 4 + wave) * J;                               // (first) 16-row tile of this wave
    const int row0 = tile * 16;
    const bool tile_ok = row0 < M;
    const int tile_c = tile_ok ? tile : 0;                                      // waves past the last row keep the barriers company
    // outputs of this lane: rows row0 + 4g .. + 3, column col0 + (lane & 15)

    // ---- A operands: 8 int8 of row (lane & 15), elements 8 g .. 8 g + 7 of each block; block b of the tile is 512 bytes further ----
    const uint8_t *const aq = img.qs + ((size_t)tile_c * BPR * 16 + li) * 32 + 8 * g;
    uint2 qa[4][CH];                                                            // batch gb lives in qa[gb & 3]
    auto load_a = [&](int gb, uint2 (&dst)[CH]) {
#if defined(MFMA_ABLATE) && MFMA_ABLATE == 3      // (tools/microbench22 only: the loop without one of its ingredients -- 1 weight-scale reads, 2 B-operand reads, 3 A-operand loads, 4 MFMAs, 5 block arithmetic)
#pragma unroll
        for (int j = 0; j < CH; j++) { dst[j] = make_uint2((unsigned)gb + lane, (unsigned)j); asm volatile("" : "+v"(dst[j].x), "+v"(dst[j].y)); }
        return;
#endif
#pragma unroll
        for (int j = 0; j < CH; j++) dst[j] = *reinterpret_cast<const uint2 *>(aq + (size_t)(gb * CH + j) * 512);
    };
    load_a(0, qa[0]); load_a(1, qa[1]); load_a(2, qa[2]);

    // epilogue inputs (independent loads); M is a multiple of 4 everywhere
    float4 e_bias = make_float4(0.f, 0.f, 0.f, 0.f), e_res = make_float4(0.f, 0.f, 0.f, 0.f);   // requested inside the LAST phase: held from here they are spilled over the loop
    float4 e_bias1 = make_float4(0.f, 0.f, 0.f, 0.f);                           // J = 2: the second tile's
    const int colc = min(col0 + li, p.N - 1);
    int e_npast = 0, e_seq = 0;
    if (EPI == EPI_QKV && !EPI_LATE) {
        e_npast = p.seq ? p.seq[colc].n_past : p.st->n_past + colc;
        e_seq = (p.seq && p.col_mode) ? p.seq[colc].seq_id : colc;
    }

    // this lane's LDS addresses inside a buffer
    const int o_b = L::OFF_Q + li * PITCH + 8 * g;                              // B operand: column lane & 15, k-group g; block b is 32 bytes further
    const int o_d = L::OFF_D + li * 16, o_s = L::OFF_S + li * 16;               // the column's block scales / sums; batch n is 256 bytes further
    const int o_w = L::OFF_W + wv * (J * BPP * SWF * 4) + 16 * g;               // rows 4g .. 4g+3 of the tile's scales; block b is SWF * 4 bytes further (the second tile's follow the first's)

    mm_f2 acc2[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};                               // the lane's four sums, as two register pairs (packed f32 arithmetic)
    mm_f2 accA[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};                               // J = 2: the first tile's sums, parked while the second tile runs
    long sb[2][CH];                                                             // B operands of batch n: sb[n & 1]
    float4 sw[2][CH], sm[2][CH], xd[2], xs[2];                                  // weight scales / mins, activation scales / sums of batch n
    i32x4 cc[CH];                                                               // integer dots: block j of batch n until step n has converted them, then block j of batch n + 1
    // The matrix core starts every block's integer dot at 0x4B400000 = the bit pattern of 12582912.0f (1.5 x 2^23): for |dot| < 2^22 (a block of 32 int8 products is below
    // 2^19.1) the int32 result READ AS A FLOAT is 12582912 + dot exactly, and (that - 12582912.0f) is (float)dot -- one full-rate v_pk_add_f32 for two conversions where
    // v_cvt_f32_i32 issues at half rate (tools/microbench24: 3.84 against 2.04 cycles per wave-instruction and SIMD at four waves).  Same value, bit for bit.
    i32x4 zero = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
    asm volatile("" : "+v"(zero));                                              // (held in registers: the C operand of an MFMA is a register quad)
    const mm_f2 bias2 = {__int_as_float(zero[0]), __int_as_float(zero[1])};    // 12582912.0f twice: the registers of the C operand, read as floats (no constant of its own)

    // one phase; LAST (compile time): the epilogue's inputs are requested inside it (the last phase is peeled off the loop so that they are not loop-carried)
    auto phase = [&](const int ph, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const unsigned char *const base = smem_raw + (ph & 1) * L::BYTES;
        // barrier ph: phase ph has landed, this wave is done with phase ph - 1 (a bare s_barrier: __syncthreads() carries a release fence that drains EVERY
        // load, the A-operand prefetch included)
        if (ph == 0) __builtin_amdgcn_s_waitcnt(0x0070);                        // vmcnt(0) lgkmcnt(0): this wave's share of phase 0 has landed (and the first A operands with it)
        else __builtin_amdgcn_s_waitcnt(0xc07f);                                // lgkmcnt(0); vmcnt / expcnt untouched
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        auto lds_b = [&](int n, long (&dst)[CH]) {
#if defined(MFMA_ABLATE) && MFMA_ABLATE == 2
#pragma unroll
            for (int j = 0; j < CH; j++) { dst[j] = (long)n * 0x0101010101010101L + lane; asm volatile("" : "+v"(dst[j])); }
            return;
#endif
#pragma unroll
            for (int j = 0; j < CH; j++) dst[j] = *reinterpret_cast<const long *>(base + o_b + ((n & (NB - 1)) * CH + j) * QK);
        };
        auto lds_s = [&](int n, int s) {
#pragma unroll
            for (int j = 0; j < CH; j++) {
#if defined(MFMA_ABLATE) && MFMA_ABLATE == 1
                sw[s][j] = make_float4(1.0f + n, 2.0f, 3.0f + j, 4.0f); asm volatile("" : "+v"(sw[s][j].x), "+v"(sw[s][j].y), "+v"(sw[s][j].z), "+v"(sw[s][j].w));
                continue;
#endif
                sw[s][j] = *reinterpret_cast<const float4 *>(base + o_w + (n * CH + j) * (SWF * 4));
                if (Q81) sm[s][j] = *reinterpret_cast<const float4 *>(base + o_w + (n * CH + j) * (SWF * 4) + 64);
            }
            xd[s] = *reinterpret_cast<const float4 *>(base + o_d + (n & (NB - 1)) * 256);
            if (Q81) xs[s] = *reinterpret_cast<const float4 *>(base + o_s + (n & (NB - 1)) * 256);
        };
        // pipeline fill: operands of batches 0 and 1, scales of batch 0, the MFMAs of batch 0
        lds_b(0, sb[0]); lds_s(0, 0); lds_b(1, sb[1]);
#pragma unroll
        for (int j = 0; j < CH; j++)
            cc[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)(((unsigned long)qa[0][j].y << 32) | qa[0][j].x), sb[0][j], zero, 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NBL; n++) {
            const int gb = ph * NB + n;                                         // batch of the whole row (J = 2: of the wave's two rows of tiles, one after the other in the image)
            __builtin_amdgcn_sched_barrier(0);                                  // the stages of one step stay in their step (left alone the scheduler hoists every load of the phase and spills)
            if (!(LAST && n + QDEPTH >= NBL)) load_a(gb + QDEPTH, qa[(n + QDEPTH) & 3]);   // (compile time: the row ends with this phase)
            if (n + 1 < NBL) lds_s(n + 1, (n + 1) & 1);
            if (n + 2 < NBL) lds_b(n + 2, sb[n & 1]);                          // batch n's operands went into its MFMAs one step ago
            if (n == NBL - QDEPTH && LAST && !EPI_LATE) {                       // the A-operand registers of the batches past the end are free from here
                int t2 = wv * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(t2));   // indices recomputed from the wave's SGPR and the lane counter, not held over the loop (the pipeline fills all 128 registers; threadIdx.x itself would be one more)
                const int orc2 = min((int)(blockIdx.x * 4 + (t2 >> 6)) * (16 * J) + ((t2 >> 2) & 12), M - (J == 2 ? 20 : 4)), colc2 = min(col0 + (t2 & 15), p.N - 1);
                if (EPI != EPI_LOGITS) e_bias = *reinterpret_cast<const float4 *>(p.bias + orc2);
                if (J == 2) e_bias1 = *reinterpret_cast<const float4 *>(p.bias + orc2 + 16);
                if (EPI == EPI_RESID) e_res = *reinterpret_cast<const float4 *>(p.resid + (size_t)colc2 * p.ldr + orc2);
            }
            // The arithmetic of the step in stages that the scheduler may not mix (sched_barrier): inside a stage every instruction is independent of its neighbours
            // (left to itself the compiler walks block by block -- cvt, mul, mul, add back to back, each waiting out the latency of the one before).
            mm_f2 t[CH][2];
#if defined(MFMA_ABLATE) && MFMA_ABLATE == 5
#pragma unroll
            for (int j = 0; j < CH; j++) {
                acc2[0].x += __int_as_float(cc[j][0] ^ cc[j][1]) + sw[n & 1][j].x; acc2[1].x += __int_as_float(cc[j][2] ^ cc[j][3]) + xd[n & 1].x;
                if (n + 1 < NBL) { const uint2 a = qa[(n + 1) & 3][j]; cc[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)(((unsigned long)a.y << 32) | a.x), sb[(n + 1) & 1][j], zero, 0, 0, 0); }
            }
            asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]));
            continue;
#endif
#pragma unroll
            for (int j = 0; j < CH; j++) {                                      // stage 1: the 16 conversions as 8 packed adds; the dots' registers are free
                const mm_f2 lo = {__int_as_float(cc[j][0]), __int_as_float(cc[j][1])}, hi = {__int_as_float(cc[j][2]), __int_as_float(cc[j][3])};
                t[j][0] = lo - bias2; t[j][1] = hi - bias2;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CH; j++) {                                      // stage 2: the next batch's MFMAs, one per four first products
                if (n + 1 < NBL) {
                    const uint2 a = qa[(n + 1) & 3][j];
#if defined(MFMA_ABLATE) && MFMA_ABLATE == 4
                    cc[j] = i32x4{(int)a.x, (int)a.y, (int)sb[(n + 1) & 1][j], (int)(sb[(n + 1) & 1][j] >> 32)} | zero;
#else
                    cc[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)(((unsigned long)a.y << 32) | a.x), sb[(n + 1) & 1][j], zero, 0, 0, 0);
#endif
                }
                const float4 dw = sw[n & 1][j];
                const float xdj = j == 0 ? xd[n & 1].x : j == 1 ? xd[n & 1].y : j == 2 ? xd[n & 1].z : xd[n & 1].w;
                const mm_f2 dw2[2] = {{dw.x, dw.y}, {dw.z, dw.w}}, xd2 = {xdj, xdj};
                if (WT == W_Q4_0) {
                    t[j][0] = t[j][0] * dw2[0]; t[j][1] = t[j][1] * dw2[1];                     // (dot * d_w) ...
                } else {                                                        // d_w * d_x first: both products here
                    const mm_f2 dx0 = dw2[0] * xd2, dx1 = dw2[1] * xd2;
                    t[j][0] = t[j][0] * dx0; t[j][1] = t[j][1] * dx1;                           // dot * (d_w d_x) / (d_w d_x) * dot: the same product
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CH; j++) {                                      // stage 3: Q4_0's second product; the min term of Q4_1 / Q5_1
                const float xdj = j == 0 ? xd[n & 1].x : j == 1 ? xd[n & 1].y : j == 2 ? xd[n & 1].z : xd[n & 1].w;
                const float xsj = !Q81 ? 0.0f : j == 0 ? xs[n & 1].x : j == 1 ? xs[n & 1].y : j == 2 ? xs[n & 1].z : xs[n & 1].w;
                const float4 mw = Q81 ? sm[n & 1][j] : make_float4(0.f, 0.f, 0.f, 0.f);
                const mm_f2 xd2 = {xdj, xdj}, xs2 = {xsj, xsj}, mw2[2] = {{mw.x, mw.y}, {mw.z, mw.w}};
                if (WT == W_Q4_0) { t[j][0] = t[j][0] * xd2; t[j][1] = t[j][1] * xd2; }         // ... * d_x
                if (Q81) { t[j][0] = t[j][0] + mw2[0] * xs2; t[j][1] = t[j][1] + mw2[1] * xs2; }   // + m_w * s_x (-ffp-contract=off: a product and a sum)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CH; j++) {                                      // stage 4: the sums, in block order (the one true dependence across blocks)
                acc2[0] = acc2[0] + t[j][0]; acc2[1] = acc2[1] + t[j][1];
            }
            if (J == 2 && n == NB - 1) {      // the first tile's row is done: its sums wait for the epilogue (named like the running sums below: see there)
                asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]));
                accA[0] = acc2[0]; accA[1] = acc2[1]; acc2[0] = mm_f2{0.0f, 0.0f}; acc2[1] = mm_f2{0.0f, 0.0f};
                asm volatile("" : "+v"(accA[0]), "+v"(accA[1]));
            }
            // the step's arithmetic is DONE in the step: without a side effect that names the sums, instruction selection sinks all of a one-phase kernel's
            // cvt / mul / add behind its last MFMA (the sums are only used by the epilogue) and every block's integer dots and scales stay live -- spills
            asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]));
        }
    };
#pragma unroll 1
    for (int ph = 0; ph < NPH - 1; ph++) phase(ph, std::false_type{});
    phase(NPH - 1, std::true_type{});

    MFMA_STAMP(2); MFMA_STAMP(5);
    if (EPI == EPI_GELU_Q8 && J == 2) {
        // the wave's 32 rows x 16 columns: per column ONE Q8 block of fc2's activation row (quantize_row_q8_0 / _q8_1), its 32 elements in this lane quad-of-rows (g = 0 .. 3)
        // x two tiles x four rows: element 16 jt + 4 g + r.  No exchange through LDS, no barrier.
        const float b0[4] = {e_bias.x, e_bias.y, e_bias.z, e_bias.w}, b1[4] = {e_bias1.x, e_bias1.y, e_bias1.z, e_bias1.w};
        const float a0[4] = {accA[0].x, accA[0].y, accA[1].x, accA[1].y}, a1[4] = {acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y};
        float v[2][4];
#pragma unroll
        for (int r = 0; r < 4; r++) { v[0][r] = h2f(p.gelu_tab[f2h(__fadd_rn(b0[r], a0[r]))]); v[1][r] = h2f(p.gelu_tab[f2h(__fadd_rn(b1[r], a1[r]))]); }
        float amax = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) amax = fmaxf(amax, fmaxf(fabsf(v[0][r]), fabsf(v[1][r])));
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64)); amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const float d = amax / 127.0f;
        const float id = (d != 0.0f) ? 1.0f / d : 0.0f;
        int q[2][4], isum = 0;
#pragma unroll
        for (int jt = 0; jt < 2; jt++)
#pragma unroll
            for (int r = 0; r < 4; r++) { q[jt][r] = (int)roundf(__fmul_rn(v[jt][r], id)); isum += q[jt][r]; }
        isum += __shfl_xor(isum, 16, 64); isum += __shfl_xor(isum, 32, 64);
        if (col0 + li < p.N && tile_ok) {
            const size_t blk = (size_t)(col0 + li) * (M / 32) + (size_t)(blockIdx.x * 4 + wv);   // column-major [N][d_ff/32]
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
                *reinterpret_cast<uint32_t *>(p.oq_q + blk * 32 + 16 * jt + 4 * g) =
                    (uint32_t)(q[jt][0] & 0xFF) | ((uint32_t)(q[jt][1] & 0xFF) << 8) | ((uint32_t)(q[jt][2] & 0xFF) << 16) | ((uint32_t)(q[jt][3] & 0xFF) << 24);
            if (g == 0) {
                if (Q81) { p.oq_d[blk] = d; p.oq_s[blk] = __float_as_uint(__fmul_rn((float)isum, d)); }
                else { p.oq_d[blk] = h2f(f2h(d)); p.oq_s[blk] = (uint32_t)isum; }
            }
        }
        MFMA_STAMP(3); MFMA_STAMP(7);
        return;
    }
    const float acc[4] = {acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y};
    if (EPI == EPI_GELU_Q8) {
        __syncthreads();                                                        // everyone is done reading the activation area
        {
            float4 v;
            v.x = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias.x, acc[0]))]); v.y = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias.y, acc[1]))]);
            v.z = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias.z, acc[2]))]); v.w = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias.w, acc[3]))]);
            *reinterpret_cast<float4 *>(s_tail + li * 64 + wave * 16 + 4 * g) = v;
        }
        __syncthreads();
        // the workgroup's 64 rows are two Q8 blocks of fc2's activation row per column: quantize_row_q8_0 / _q8_1,
        // a half-wave per (column, block)
        for (int u = wave * 2 + (lane >> 5); u < 32; u += 8) {
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
                if (Q81) { p.oq_d[blk] = d; p.oq_s[blk] = __float_as_uint(__fmul_rn((float)isum, d)); }
                else { p.oq_d[blk] = h2f(f2h(d)); p.oq_s[blk] = (uint32_t)isum; }
            }
        }
        MFMA_STAMP(3); MFMA_STAMP(7);
        return;
    }
    int t3 = wv * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(t3));   // (as above)
    const int col = col0 + (t3 & 15);
    if (EPI_LATE) {
        const int orc3 = min((int)(blockIdx.x * 4 + (t3 >> 6)) * (16 * J) + ((t3 >> 2) & 12), M - 20), colc3 = min(col, p.N - 1);
        e_bias = *reinterpret_cast<const float4 *>(p.bias + orc3);
        e_bias1 = *reinterpret_cast<const float4 *>(p.bias + orc3 + 16);
        e_npast = p.seq ? p.seq[colc3].n_past : p.st->n_past + colc3;
        e_seq = (p.seq && p.col_mode) ? p.seq[colc3].seq_id : colc3;
    }
#pragma unroll
    for (int jt = 0; jt < J; jt++) {
        const int orow = (int)(blockIdx.x * 4 + (t3 >> 6)) * (16 * J) + 16 * jt + ((t3 >> 2) & 12);
        if (!(col < p.N && orow < M)) return;
        const mm_f2 s0 = (J == 2 && jt == 0) ? accA[0] : acc2[0], s1 = (J == 2 && jt == 0) ? accA[1] : acc2[1];
        const float4 eb = (J == 2 && jt == 1) ? e_bias1 : e_bias;
        const float av[4] = {s0.x, s0.y, s1.x, s1.y};
        if (EPI == EPI_QKV) {
            float4 v;
            v.x = __fadd_rn(eb.x, av[0]); v.y = __fadd_rn(eb.y, av[1]); v.z = __fadd_rn(eb.z, av[2]); v.w = __fadd_rn(eb.w, av[3]);
            const int which = orow / K, rr = orow - which * K;                 // d_model == K for the q/k/v projection
            if (which == 0) {
                v.x = __fmul_rn(v.x, p.q_scale); v.y = __fmul_rn(v.y, p.q_scale); v.z = __fmul_rn(v.z, p.q_scale); v.w = __fmul_rn(v.w, p.q_scale);
                *reinterpret_cast<float4 *>(p.q_out + (size_t)col * K + rr) = v;
            } else {
                float *cache = ((which == 1) ? p.kcache : p.vcache) + (p.seq ? (size_t)e_seq * p.kv_seq_stride : 0);
                const int hh = rr >> p.dk_log2, dd = rr & (p.dk - 1);           // head-major cache: [H][P][dk]; 4 | dk
                *reinterpret_cast<float4 *>(cache + (((size_t)hh * p.P + e_npast) << p.dk_log2) + dd) = v;
            }
        } else if (EPI == EPI_RESID) {
            float4 v;
            v.x = __fadd_rn(__fadd_rn(av[0], eb.x), e_res.x); v.y = __fadd_rn(__fadd_rn(av[1], eb.y), e_res.y);
            v.z = __fadd_rn(__fadd_rn(av[2], eb.z), e_res.z); v.w = __fadd_rn(__fadd_rn(av[3], eb.w), e_res.w);
            *reinterpret_cast<float4 *>(p.out + (size_t)col * p.ldo + orow) = v;
        } else {
            *reinterpret_cast<float4 *>(p.out + (size_t)col * p.ldo + orow) = make_float4(av[0], av[1], av[2], av[3]);
        }
    }
    MFMA_STAMP(3); MFMA_STAMP(7);
}

}  // namespace bgk
