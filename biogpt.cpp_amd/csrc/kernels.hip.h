// Hand-written HIP kernels (gfx950 / CDNA4, wave64) for the BioGPT decoder forward pass.
//
// What the reference computes on this path is the ggml op sequence of biogpt_graph
// (biogpt.cpp:624-810).  The kernels here restate that arithmetic for the MI355X:
//
//   embed_kernel      get_rows + scale + get_rows + add            biogpt.cpp:664-686
//   matvec_kernel     [LayerNorm] -> activation Q8 quantize -> block-quantized W x  -> epilogue
//                       QKV   : + bias, Q * 1/sqrt(dk), K/V appended to the F32 cache  :691-727
//                       RESID : + bias + residual (out_proj, fc2)                      :767-772, :790-795
//                       GELU  : + bias, GELU through the fp16 table (fc1)              :779-787
//                       LOGITS: final LayerNorm + lm_head (+ fused partial arg-max)    :799-803
//   attn_kernel       QK^T over all T cached keys, softmax (fp16-table exp, no mask), PV  :729-764
//   argmax_kernel     greedy sampler (top_k = 1) + token feedback for the device loop  main.cpp:109-128
//
// Numerics follow ggml's CPU path: weights stay block-quantized, the F32
// activation is quantized to Q8_0 / Q8_1 per 32-block and the dot is an int8 dot (v_dot4_i32_i8)
// scaled by d_w * d_x; LayerNorm statistics and the F32 attention dots accumulate in double;
// GELU / softmax-exp go through 65536-entry fp16 tables uploaded by the host.
//
// Device data layout (ours; the file format is only the drop-in boundary): every quantized matrix
// is repacked at load time into structure-of-arrays form so that a wave reads 16 aligned bytes per
// lane:  qs[row][block] = 16 B of nibbles (32 B of int8 for Q8_0), sc[row][block] = fp16 d
// (half2 {d,m} for Q4_1/Q5_1), qh[row][block] = the 32 fifth bits (Q5_x).
//
// Mat-vec decomposition: a "unit" is 16 B of quants (one 32-element block; 32 B for Q8_0) or 16 B
// of f16/f32 values.  LPR = min(64, pow2ceil(units per row)) lanes share a row, each lane handles
// NIT units of it, 64/LPR rows per wave step; partial sums are combined with xor-shuffles.  The
// weight loads of the first step are issued BEFORE the activation prologue so that the HBM latency
// of the weight stream hides behind the LayerNorm/quantize work (the streams are independent).
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bgk {

constexpr int QK = 32;

enum WType : int { W_F32 = 0, W_F16 = 1, W_Q4_0 = 2, W_Q4_1 = 3, W_Q5_0 = 6, W_Q5_1 = 7, W_Q8_0 = 8 };
enum Prologue : int { PRO_PLAIN = 0, PRO_LN = 1 };
enum Epilogue : int { EPI_QKV = 0, EPI_RESID = 1, EPI_GELU = 2, EPI_LOGITS = 3 };

// Per-context mutable device state: the decode position and the token ring the kernels read
// their inputs from (so a captured graph can advance itself without host involvement).
struct DevState {
    int32_t n_past;   // tokens already in the KV cache
    int32_t n_gen;    // number of ids written to gen_ids by argmax_kernel
    int32_t causal;   // 0: reference behaviour (no intra-chunk mask, F1); 1: opt-in causal mask
    int32_t pad;
    // followed in memory by: int32 tokens[n_positions]; int32 gen_ids[n_positions]
};
__device__ __forceinline__ const int32_t *state_tokens(const DevState *st) { return reinterpret_cast<const int32_t *>(st + 1); }
__device__ __forceinline__ int32_t *state_tokens(DevState *st) { return reinterpret_cast<int32_t *>(st + 1); }

struct DevMatrix {
    const uint8_t *qs;   // quants / float values
    const uint8_t *sc;   // per-block scales (see header comment); unused for float types
    const uint32_t *qh;  // Q5 fifth bits; unused otherwise
    int32_t type;
    int32_t M;           // rows
    int32_t K;           // row length (elements)
};

// ---- small helpers --------------------------------------------------------------------------------
__device__ __forceinline__ float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ uint16_t f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }

template <typename T>
__device__ __forceinline__ T wave_xor_sum(T v, int width) {  // sum over aligned groups of `width` lanes
    for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-wide sum of doubles; every thread gets the same value. scratch: >= 16 doubles of LDS.
__device__ __forceinline__ double block_sum_f64(double v, double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_xor_sum(v, 64);
    __syncthreads();  // scratch may still be read from a previous reduction
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; w++) t += scratch[w];
    return t;
}
__device__ __forceinline__ float block_max_f32(float v, float *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = scratch[0];
    for (int w = 1; w < nw; w++) t = fmaxf(t, scratch[w]);
    return t;
}

// ---- dequantize one element of a repacked matrix (embedding gather) ------------------------------
__device__ __forceinline__ float dequant_elem(const DevMatrix &m, int64_t row, int col) {
    const int bpr = m.K / QK;
    if (m.type == W_F32) return reinterpret_cast<const float *>(m.qs)[row * m.K + col];
    if (m.type == W_F16) return h2f(reinterpret_cast<const uint16_t *>(m.qs)[row * m.K + col]);
    const int64_t blk = row * bpr + col / QK;
    const int j = col % QK;
    if (m.type == W_Q8_0) {
        const int8_t q = reinterpret_cast<const int8_t *>(m.qs)[blk * 32 + j];
        return __fmul_rn((float)q, h2f(reinterpret_cast<const uint16_t *>(m.sc)[blk]));
    }
    const uint8_t byte = m.qs[blk * 16 + (j & 15)];
    int q = (j < 16) ? (byte & 0x0F) : (byte >> 4);
    if (m.type == W_Q5_0 || m.type == W_Q5_1) q |= (int)((m.qh[blk] >> j) & 1u) << 4;
    if (m.type == W_Q4_0) return __fmul_rn((float)(q - 8), h2f(reinterpret_cast<const uint16_t *>(m.sc)[blk]));
    if (m.type == W_Q5_0) return __fmul_rn((float)(q - 16), h2f(reinterpret_cast<const uint16_t *>(m.sc)[blk]));
    const uint16_t *dm = reinterpret_cast<const uint16_t *>(m.sc) + blk * 2;  // Q4_1 / Q5_1
    return __fadd_rn(__fmul_rn((float)q, h2f(dm[0])), h2f(dm[1]));
}

// x[n][d] = dequant(embed_tokens[tok[n]])[d] * sqrt(D) + dequant(embed_pos[n_past + n + 2])[d]
// (biogpt.cpp:664-686; position offset +2 at :672; embedding scale F7)
__global__ void embed_kernel(DevMatrix tok_emb, DevMatrix pos_emb, const DevState *st, float embed_scale,
                             float *x, int D) {
    const int n = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const int32_t tok = state_tokens(st)[n];
    const int32_t pos = st->n_past + n + 2;
    const float te = __fmul_rn(dequant_elem(tok_emb, tok, d), embed_scale);
    const float pe = dequant_elem(pos_emb, pos, d);
    x[(size_t)n * D + d] = __fadd_rn(te, pe);
}

// ---- mat-vec ---------------------------------------------------------------------------------------
struct MatvecParams {
    DevMatrix W;
    int32_t upr;       // units per row
    int32_t lpr_log2;  // log2(lanes per row)
    int32_t nit;       // units per lane per row (ceil(upr / lpr))
    int32_t rpw;       // rows per wave (multiple of 64 / lpr)
    // activations: N columns of K floats
    const float *x;
    int32_t ldx;
    int32_t N;
    const float *ln_w;
    const float *ln_b;
    float eps;
    const float *bias;   // [M]
    const float *resid;  // EPI_RESID: [N][ldr]
    int32_t ldr;
    float *out;          // RESID/GELU/LOGITS: [N][ldo]
    int32_t ldo;
    // EPI_QKV
    float *q_out;        // [N][D]
    float *kcache;       // layer slice of memory_k: [P][D]
    float *vcache;
    int32_t D;
    float q_scale;
    const DevState *st;
    // EPI_GELU
    const uint16_t *gelu_tab;
    // EPI_LOGITS: fused partial arg-max (N == 1 only); may be null
    float *pmax_val;
    int32_t *pmax_idx;
};

template <int WT> struct TypeInfo;
template <> struct TypeInfo<W_F32>  { static constexpr bool quant = false; static constexpr int qbytes = 16; static constexpr int elems = 4; static constexpr bool q81 = false; };
template <> struct TypeInfo<W_F16>  { static constexpr bool quant = false; static constexpr int qbytes = 16; static constexpr int elems = 8; static constexpr bool q81 = false; };
template <> struct TypeInfo<W_Q4_0> { static constexpr bool quant = true;  static constexpr int qbytes = 16; static constexpr int elems = 32; static constexpr bool q81 = false; };
template <> struct TypeInfo<W_Q4_1> { static constexpr bool quant = true;  static constexpr int qbytes = 16; static constexpr int elems = 32; static constexpr bool q81 = true; };
template <> struct TypeInfo<W_Q5_0> { static constexpr bool quant = true;  static constexpr int qbytes = 16; static constexpr int elems = 32; static constexpr bool q81 = false; };
template <> struct TypeInfo<W_Q5_1> { static constexpr bool quant = true;  static constexpr int qbytes = 16; static constexpr int elems = 32; static constexpr bool q81 = true; };
template <> struct TypeInfo<W_Q8_0> { static constexpr bool quant = true;  static constexpr int qbytes = 32; static constexpr int elems = 32; static constexpr bool q81 = false; };

// registers holding one weight unit
template <int WT>
struct Unit {
    uint4 q0;
    uint4 q1;      // second half of a Q8_0 block
    uint32_t sc;   // fp16 d (low 16 bits) or half2 {d, m}
    uint32_t qh;
};

template <int WT>
__device__ __forceinline__ void load_unit(Unit<WT> &u, const DevMatrix &W, int64_t idx) {
    using TI = TypeInfo<WT>;
    const uint4 *q = reinterpret_cast<const uint4 *>(W.qs + idx * TI::qbytes);
    u.q0 = q[0];
    if (WT == W_Q8_0) u.q1 = q[1];
    if (TI::quant) {
        if (TI::q81) u.sc = reinterpret_cast<const uint32_t *>(W.sc)[idx];
        else u.sc = reinterpret_cast<const uint16_t *>(W.sc)[idx];
        if (WT == W_Q5_0 || WT == W_Q5_1) u.qh = W.qh[idx];
    }
}

__device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }

// spread 4 bits of t to bit 4 of each byte
__device__ __forceinline__ uint32_t spread4(uint32_t t) { return (((t & 0xFu) * 0x00204081u) & 0x01010101u) << 4; }

// One unit against one activation column. Quant types: xq = the 32 int8 of the matching
// activation block (8 dwords in LDS), xd = activation scale, xs = integer sum (Q8_0) or d*sum (Q8_1).
template <int WT>
__device__ __forceinline__ float unit_dot_quant(const Unit<WT> &u, const uint32_t *xq, float xd, float xs_f, int xs_i) {
    const uint4 xa = *reinterpret_cast<const uint4 *>(xq);      // elements 0..15
    const uint4 xb = *reinterpret_cast<const uint4 *>(xq + 4);  // elements 16..31
    int s = 0;
    if (WT == W_Q8_0) {
        s = dot4(u.q0.x, xa.x, s); s = dot4(u.q0.y, xa.y, s); s = dot4(u.q0.z, xa.z, s); s = dot4(u.q0.w, xa.w, s);
        s = dot4(u.q1.x, xb.x, s); s = dot4(u.q1.y, xb.y, s); s = dot4(u.q1.z, xb.z, s); s = dot4(u.q1.w, xb.w, s);
        const float dw = h2f((uint16_t)u.sc);
        return __fmul_rn((float)s, __fmul_rn(dw, xd));  // sumi*(d_w*d_x)
    }
    uint32_t lo[4] = {u.q0.x & 0x0F0F0F0Fu, u.q0.y & 0x0F0F0F0Fu, u.q0.z & 0x0F0F0F0Fu, u.q0.w & 0x0F0F0F0Fu};
    uint32_t hi[4] = {(u.q0.x >> 4) & 0x0F0F0F0Fu, (u.q0.y >> 4) & 0x0F0F0F0Fu, (u.q0.z >> 4) & 0x0F0F0F0Fu, (u.q0.w >> 4) & 0x0F0F0F0Fu};
    if (WT == W_Q5_0 || WT == W_Q5_1) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            lo[i] |= spread4(u.qh >> (4 * i));
            hi[i] |= spread4(u.qh >> (16 + 4 * i));
        }
    }
    s = dot4(lo[0], xa.x, s); s = dot4(lo[1], xa.y, s); s = dot4(lo[2], xa.z, s); s = dot4(lo[3], xa.w, s);
    s = dot4(hi[0], xb.x, s); s = dot4(hi[1], xb.y, s); s = dot4(hi[2], xb.z, s); s = dot4(hi[3], xb.w, s);
    if (WT == W_Q4_0) {
        s -= 8 * xs_i;
        return __fmul_rn(__fmul_rn((float)s, h2f((uint16_t)u.sc)), xd);  // sumi*d_w*d_x
    }
    if (WT == W_Q5_0) {
        s -= 16 * xs_i;
        return __fmul_rn(__fmul_rn(h2f((uint16_t)u.sc), xd), (float)s);  // (d_w*d_x)*sumi
    }
    // Q4_1 / Q5_1: (d_w*d_x)*sumi + m_w*s_x
    const float dw = h2f((uint16_t)(u.sc & 0xFFFFu)), mw = h2f((uint16_t)(u.sc >> 16));
    return __fadd_rn(__fmul_rn(__fmul_rn(dw, xd), (float)s), __fmul_rn(mw, xs_f));
}

template <int WT>
__device__ __forceinline__ double unit_dot_float(const Unit<WT> &u, const float *xf) {
    double acc = 0.0;
    if (WT == W_F32) {
        const float w[4] = {__uint_as_float(u.q0.x), __uint_as_float(u.q0.y), __uint_as_float(u.q0.z), __uint_as_float(u.q0.w)};
        const float4 xv = *reinterpret_cast<const float4 *>(xf);
        acc += (double)__fmul_rn(w[0], xv.x); acc += (double)__fmul_rn(w[1], xv.y);
        acc += (double)__fmul_rn(w[2], xv.z); acc += (double)__fmul_rn(w[3], xv.w);
    } else {
        const uint32_t p[4] = {u.q0.x, u.q0.y, u.q0.z, u.q0.w};
        const float4 x0 = *reinterpret_cast<const float4 *>(xf);
        const float4 x1 = *reinterpret_cast<const float4 *>(xf + 4);
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            acc += (double)__fmul_rn(h2f((uint16_t)(p[i] & 0xFFFFu)), xv[2 * i]);
            acc += (double)__fmul_rn(h2f((uint16_t)(p[i] >> 16)), xv[2 * i + 1]);
        }
    }
    return acc;
}

// LDS carve for the mat-vec kernel (all offsets multiples of 16 bytes)
struct MatvecSmem {
    float *stage;     // [K]      raw / normalised column being processed
    uint32_t *xq;     // [NC][K/4] int8 activations (quant) -- or float [NC][K] for float types
    float *xd;        // [NC][K/32]
    float *xsf;       // [NC][K/32]
    int *xsi;         // [NC][K/32]
    double *red;      // [16]
};

__host__ __device__ inline size_t matvec_smem_bytes(int wtype, int K, int NC) {
    const bool quant = !(wtype == W_F32 || wtype == W_F16);
    size_t b = (size_t)K * 4;                                    // stage
    b += quant ? (size_t)NC * K : (size_t)NC * K * 4;            // xq / xf
    b += 3 * (size_t)NC * (K / QK + 4) * 4;                      // xd, xsf, xsi (padded)
    b += 16 * 8;                                                 // red
    return (b + 255) & ~(size_t)255;
}

template <int WT, int PRO, int EPI, int NC, bool SEQ = true>
__global__ __launch_bounds__(256) void matvec_kernel(const MatvecParams p) {
    using TI = TypeInfo<WT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = p.W.K, M = p.W.M;
    const int nblk = K / QK;
    const int nblk_pad = nblk + 4;

    MatvecSmem sm;
    {
        unsigned char *ptr = smem_raw;
        sm.stage = reinterpret_cast<float *>(ptr); ptr += (size_t)K * 4;
        sm.xq = reinterpret_cast<uint32_t *>(ptr); ptr += TI::quant ? (size_t)NC * K : (size_t)NC * K * 4;
        sm.xd = reinterpret_cast<float *>(ptr); ptr += (size_t)NC * nblk_pad * 4;
        sm.xsf = reinterpret_cast<float *>(ptr); ptr += (size_t)NC * nblk_pad * 4;
        sm.xsi = reinterpret_cast<int *>(ptr); ptr += (size_t)NC * nblk_pad * 4;
        sm.red = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(ptr) + 15) & ~(uintptr_t)15);
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int lpr = 1 << p.lpr_log2;
    const int rps = 64 >> p.lpr_log2;               // rows per wave step
    const int sub = lane & (lpr - 1);               // lane's position inside its row
    const int rsub = lane >> p.lpr_log2;            // row inside the step
    const int nsteps = p.rpw / rps;
    const int64_t row_base = ((int64_t)blockIdx.x * nwaves + wave) * p.rpw;
    const int col0 = blockIdx.y * NC;
    const int ncols = min(NC, p.N - col0);

    // ---- issue the first work item's weight loads (independent of the activations) ------------
    // A work item = (row step, chunk of MAXIT units per lane); rows longer than MAXIT*LPR units
    // (f16/f32 weights with K = d_ff) take several items per step.
    constexpr int MAXIT = 4;
    const int nitc = (p.nit + MAXIT - 1) / MAXIT;
    const int nitems = nsteps * nitc;
    Unit<WT> cur[MAXIT];
#define BG_LOAD_ITEM(dst, w)                                                                     \
    do {                                                                                         \
        const int stp_ = (w) / nitc, itc_ = (w) - stp_ * nitc;                                   \
        const int64_t row_ = row_base + (int64_t)stp_ * rps + rsub;                              \
        _Pragma("unroll") for (int it = 0; it < MAXIT; it++) {                                   \
            const int itg_ = itc_ * MAXIT + it, uu_ = sub + itg_ * lpr;                          \
            if (itg_ < p.nit && uu_ < p.upr && row_ < M) load_unit<WT>(dst[it], p.W, row_ * p.upr + uu_); \
        }                                                                                        \
    } while (0)
    BG_LOAD_ITEM(cur, 0);

    // ---- prologue: [LayerNorm] + activation conversion into LDS, one column at a time --------
    for (int c = 0; c < ncols; c++) {
        const float *xcol = p.x + (size_t)(col0 + c) * p.ldx;
        const int nchunks = K / 4;
        // stage the column (each thread re-reads only what it wrote: no barrier needed)
        for (int ch = tid; ch < nchunks; ch += nthreads)
            reinterpret_cast<float4 *>(sm.stage)[ch] = reinterpret_cast<const float4 *>(xcol)[ch];
        if (PRO == PRO_LN) {
            // ggml_norm: mean and variance in double, y = (x-mean) * 1/sqrtf(var+eps); then *w, +b
            double s = 0.0;
            for (int ch = tid; ch < nchunks; ch += nthreads) {
                const float4 v = reinterpret_cast<const float4 *>(sm.stage)[ch];
                s += (double)v.x; s += (double)v.y; s += (double)v.z; s += (double)v.w;
            }
            const float mean = (float)(block_sum_f64(s, sm.red) / (double)K);
            double s2 = 0.0;
            for (int ch = tid; ch < nchunks; ch += nthreads) {
                float4 v = reinterpret_cast<const float4 *>(sm.stage)[ch];
                v.x = __fsub_rn(v.x, mean); v.y = __fsub_rn(v.y, mean); v.z = __fsub_rn(v.z, mean); v.w = __fsub_rn(v.w, mean);
                reinterpret_cast<float4 *>(sm.stage)[ch] = v;
                s2 += (double)__fmul_rn(v.x, v.x); s2 += (double)__fmul_rn(v.y, v.y);
                s2 += (double)__fmul_rn(v.z, v.z); s2 += (double)__fmul_rn(v.w, v.w);
            }
            const float var = (float)(block_sum_f64(s2, sm.red) / (double)K);
            const float scale = 1.0f / sqrtf(__fadd_rn(var, p.eps));
            for (int ch = tid; ch < nchunks; ch += nthreads) {
                float4 v = reinterpret_cast<const float4 *>(sm.stage)[ch];
                const float4 w = reinterpret_cast<const float4 *>(p.ln_w)[ch];
                const float4 b = reinterpret_cast<const float4 *>(p.ln_b)[ch];
                v.x = __fadd_rn(__fmul_rn(w.x, __fmul_rn(v.x, scale)), b.x);
                v.y = __fadd_rn(__fmul_rn(w.y, __fmul_rn(v.y, scale)), b.y);
                v.z = __fadd_rn(__fmul_rn(w.z, __fmul_rn(v.z, scale)), b.z);
                v.w = __fadd_rn(__fmul_rn(w.w, __fmul_rn(v.w, scale)), b.w);
                reinterpret_cast<float4 *>(sm.stage)[ch] = v;
            }
        }
        if (TI::quant) {
            // quantize_row_q8_0 / q8_1: a 32-block = 8 consecutive chunks = 8 consecutive lanes
            for (int ch = tid; ch < nchunks; ch += nthreads) {
                const float4 v = reinterpret_cast<const float4 *>(sm.stage)[ch];
                float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
                const float d = amax / 127.0f;
                const float id = (d != 0.0f) ? 1.0f / d : 0.0f;
                const int q0 = (int)roundf(__fmul_rn(v.x, id)), q1 = (int)roundf(__fmul_rn(v.y, id));
                const int q2 = (int)roundf(__fmul_rn(v.z, id)), q3 = (int)roundf(__fmul_rn(v.w, id));
                int isum = q0 + q1 + q2 + q3;
                isum += __shfl_xor(isum, 1, 64);
                isum += __shfl_xor(isum, 2, 64);
                isum += __shfl_xor(isum, 4, 64);
                sm.xq[(size_t)c * (K / 4) + ch] = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) |
                                                   ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
                if ((ch & 7) == 0) {
                    const int b = ch >> 3;
                    if (TI::q81) {
                        sm.xd[c * nblk_pad + b] = d;                         // Q8_1 keeps f32 d
                        sm.xsf[c * nblk_pad + b] = __fmul_rn((float)isum, d);  // s = sum * d
                    } else {
                        sm.xd[c * nblk_pad + b] = h2f(f2h(d));               // Q8_0 stores fp16 d
                        sm.xsi[c * nblk_pad + b] = isum;
                    }
                }
            }
        } else {
            float *xf = reinterpret_cast<float *>(sm.xq) + (size_t)c * K;
            for (int ch = tid; ch < nchunks; ch += nthreads) {
                float4 v = reinterpret_cast<const float4 *>(sm.stage)[ch];
                if (WT == W_F16) {  // src1 row converted to f16 (ggml_fp32_to_fp16_row)
                    v.x = h2f(f2h(v.x)); v.y = h2f(f2h(v.y)); v.z = h2f(f2h(v.z)); v.w = h2f(f2h(v.w));
                }
                reinterpret_cast<float4 *>(xf)[ch] = v;
            }
        }
    }
    __syncthreads();

    // ---- main loop over work items ---------------------------------------------------------------
    float best_val = -INFINITY;
    int best_idx = 0x7fffffff;
    float carry[NC];   // quant types: running row sum in block order (ggml's scalar vec_dot order)
    double accd[NC];   // float types: double accumulation (order-insensitive at f32 output precision)
    for (int w = 0; w < nitems; w++) {
        Unit<WT> nxt[MAXIT];
        if (w + 1 < nitems) BG_LOAD_ITEM(nxt, w + 1);
        const int stp = w / nitc, itc = w - stp * nitc;
        const int64_t row = row_base + (int64_t)stp * rps + rsub;
        if (itc == 0) {
#pragma unroll
            for (int c = 0; c < NC; c++) { carry[c] = 0.0f; accd[c] = 0.0; }
        }
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
            const int itg = itc * MAXIT + it, uu = sub + itg * lpr;
            if (itg >= p.nit) continue;  // wave-uniform
            const bool live = uu < p.upr && row < M;
            if (TI::quant) {
                // per-block contributions, then an in-order chain over the lanes of the row so that the
                // f32 sum is associated exactly like the reference's scalar loop: ((c0 + c1) + c2) + ...
                float cc[NC], acc[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    cc[c] = 0.0f;
                    if (live && c < ncols) {
                        const uint32_t *xq = sm.xq + (size_t)c * (K / 4) + uu * 8;
                        cc[c] = unit_dot_quant<WT>(cur[it], xq, sm.xd[c * nblk_pad + uu], sm.xsf[c * nblk_pad + uu], sm.xsi[c * nblk_pad + uu]);
                    }
                    acc[c] = (sub == 0) ? __fadd_rn(carry[c], cc[c]) : cc[c];
                }
                if (SEQ) {
                    for (int s2 = 1; s2 < lpr; s2++) {
#pragma unroll
                        for (int c = 0; c < NC; c++) {
                            // lane l takes lane l-1's running sum (DPP wave_shr:1) and adds its own block
                            const float t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[c]), 0x138, 0xf, 0xf, true));
                            acc[c] = (sub == 0) ? acc[c] : __fadd_rn(t, cc[c]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NC; c++) carry[c] = __shfl(acc[c], lane | (lpr - 1), 64);
                } else {
#pragma unroll
                    for (int c = 0; c < NC; c++) carry[c] = wave_xor_sum(acc[c], lpr);
                }
            } else if (live) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if (c < ncols) {
                        const float *xf = reinterpret_cast<const float *>(sm.xq) + (size_t)c * K + uu * TI::elems;
                        accd[c] += unit_dot_float<WT>(cur[it], xf);
                    }
                }
            }
        }
        if (itc == nitc - 1) {
            float res[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (TI::quant) res[c] = carry[c];
                else res[c] = (float)wave_xor_sum(accd[c], lpr);
            }
            if (sub == 0 && row < M) {
                const int r = (int)row;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if (c >= ncols) continue;
                    const int col = col0 + c;
                    float v = res[c];
                    if (EPI == EPI_QKV) {
                        v = __fadd_rn(p.bias[r], v);
                        const int which = r / p.D, rr = r - which * p.D;
                        if (which == 0) {
                            p.q_out[(size_t)col * p.D + rr] = __fmul_rn(v, p.q_scale);
                        } else {
                            float *cache = (which == 1) ? p.kcache : p.vcache;
                            cache[(size_t)(p.st->n_past + col) * p.D + rr] = v;
                        }
                    } else if (EPI == EPI_RESID) {
                        v = __fadd_rn(v, p.bias[r]);
                        p.out[(size_t)col * p.ldo + r] = __fadd_rn(v, p.resid[(size_t)col * p.ldr + r]);
                    } else if (EPI == EPI_GELU) {
                        v = __fadd_rn(p.bias[r], v);
                        p.out[(size_t)col * p.ldo + r] = h2f(p.gelu_tab[f2h(v)]);
                    } else {
                        p.out[(size_t)col * p.ldo + r] = v;
                        if (c == 0 && (v > best_val || (v == best_val && r < best_idx))) { best_val = v; best_idx = r; }
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < MAXIT; it++) cur[it] = nxt[it];
    }
#undef BG_LOAD_ITEM

    if (EPI == EPI_LOGITS && p.pmax_val != nullptr) {
        // per-block partial arg-max (lowest index wins ties), finished by argmax_kernel
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best_val, off, 64);
            const int oi = __shfl_xor(best_idx, off, 64);
            if (ov > best_val || (ov == best_val && oi < best_idx)) { best_val = ov; best_idx = oi; }
        }
        __syncthreads();
        float *sv = reinterpret_cast<float *>(sm.red);
        int *si = reinterpret_cast<int *>(sm.red) + 8;
        if (lane == 0) { sv[wave] = best_val; si[wave] = best_idx; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < nwaves; w++)
                if (sv[w] > best_val || (sv[w] == best_val && si[w] < best_idx)) { best_val = sv[w]; best_idx = si[w]; }
            p.pmax_val[blockIdx.x] = best_val;
            p.pmax_idx[blockIdx.x] = best_idx;
        }
    }
}

// ---- attention -------------------------------------------------------------------------------------
struct AttnParams {
    const float *q;       // [N][D]  (already scaled)
    const float *kcache;  // layer slice [P][D]
    const float *vcache;
    float *out;           // [N][D]
    const DevState *st;
    const uint16_t *exp_tab;
    int32_t N, D, dk, P;
};

// One workgroup per (head, query token).  scores -> softmax -> PV exactly in the order of
// biogpt.cpp:741-764: S_j = K_j . q ; p_j = f16tab_exp(S_j - max) ; p_j *= 1/sum(double) ;
// o_d = sum_j V_jd * p_j.  LDS: S[T] + q[dk] + reduction scratch.
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int h = blockIdx.x, i = blockIdx.y;
    const int n_past = p.st->n_past;
    int T = n_past + p.N;
    if (p.st->causal) T = n_past + i + 1;
    const int dk = p.dk, D = p.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;

    float *S = reinterpret_cast<float *>(smem_raw);                    // [P]
    float *qs = S + p.P;                                               // [dk]
    double *red = reinterpret_cast<double *>(qs + dk);                 // [16]
    double *pv = red + 16;                                             // [nthreads]

    for (int d = tid; d < dk; d += nthreads) qs[d] = p.q[(size_t)i * D + (size_t)h * dk + d];
    __syncthreads();

    // scores: dk/4 lanes per key, float4 each
    const int lpk = dk >> 2;              // lanes per key (power of two, <= 64)
    const int kpw = 64 / lpk;             // keys per wave step
    const int ksub = lane % lpk, kidx = lane / lpk;
    const float4 qv = reinterpret_cast<const float4 *>(qs)[ksub];
    for (int j0 = wave * kpw; j0 < T; j0 += nwaves * kpw) {
        const int j = j0 + kidx;
        double acc = 0.0;
        if (j < T) {
            const float4 kv = *reinterpret_cast<const float4 *>(p.kcache + (size_t)j * D + (size_t)h * dk + 4 * ksub);
            acc += (double)__fmul_rn(kv.x, qv.x); acc += (double)__fmul_rn(kv.y, qv.y);
            acc += (double)__fmul_rn(kv.z, qv.z); acc += (double)__fmul_rn(kv.w, qv.w);
        }
        acc = wave_xor_sum(acc, lpk);
        if (ksub == 0 && j < T) S[j] = (float)acc;
    }
    __syncthreads();

    // softmax (ggml_soft_max: fp16-table exp, double sum, scale by (float)(1/sum))
    float mx = -INFINITY;
    for (int j = tid; j < T; j += nthreads) mx = fmaxf(mx, S[j]);
    mx = block_max_f32(mx, reinterpret_cast<float *>(red));
    double sum = 0.0;
    for (int j = tid; j < T; j += nthreads) {
        const float val = h2f(p.exp_tab[f2h(__fsub_rn(S[j], mx))]);
        S[j] = val;
        sum += (double)val;
    }
    sum = block_sum_f64(sum, red);
    const float inv = (float)(1.0 / sum);
    for (int j = tid; j < T; j += nthreads) S[j] = __fmul_rn(S[j], inv);
    __syncthreads();

    // PV: thread = (slice, d); each slice strides over the keys
    const int nsl = nthreads / dk;
    const int d = tid % dk, sl = tid / dk;
    double acc = 0.0;
    if (sl < nsl)
        for (int j = sl; j < T; j += nsl) acc += (double)__fmul_rn(p.vcache[(size_t)j * D + (size_t)h * dk + d], S[j]);
    pv[tid] = acc;
    __syncthreads();
    if (tid < dk) {
        double t = 0.0;
        for (int s2 = 0; s2 < nsl; s2++) t += pv[s2 * dk + tid];
        p.out[(size_t)i * D + (size_t)h * dk + tid] = (float)t;
    }
}

__host__ __device__ inline size_t attn_smem_bytes(int P, int dk, int nthreads) {
    return (((size_t)P + dk) * 4 + 15) / 16 * 16 + 16 * 8 + (size_t)nthreads * 8 + 64;
}

// ---- greedy sampler + token feedback (main.cpp:109-128 with top_k = 1) ----------------------------
// Finishes the arg-max over the per-block partials of the lm_head kernel, appends the id to
// gen_ids, makes it the next input token and advances n_past by n_eval (the tokens just evaluated).
__global__ __launch_bounds__(256) void argmax_kernel(const float *pmax_val, const int32_t *pmax_idx, int nparts,
                                                     DevState *st, int n_eval, int n_positions) {
    __shared__ float sv[256];
    __shared__ int si[256];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = threadIdx.x; k < nparts; k += blockDim.x) {
        const float v = pmax_val[k];
        const int ix = pmax_idx[k];
        if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float ov = sv[threadIdx.x + off];
            const int oi = si[threadIdx.x + off];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int32_t *tokens = state_tokens(st);
        int32_t *gen = tokens + n_positions;
        const int g = st->n_gen;
        if (g < n_positions) gen[g] = si[0];
        st->n_gen = g + 1;
        tokens[0] = si[0];
        st->n_past = st->n_past + n_eval;
    }
}

}  // namespace bgk
