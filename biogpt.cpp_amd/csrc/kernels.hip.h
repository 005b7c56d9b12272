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
// KV cache: F32 like the reference (F4) but HEAD-MAJOR [layer][head][position][dk] -- the reference's
// [layer][position][d_model] puts one head's rows 4 KB apart, which serialises on a memory channel.
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
enum Prologue : int { PRO_PLAIN = 0, PRO_LN = 1, PRO_Q8IN = 2 };  // Q8IN: activation already quantized by the producer kernel
enum Epilogue : int { EPI_QKV = 0, EPI_RESID = 1, EPI_GELU = 2, EPI_LOGITS = 3, EPI_GELU_Q8 = 4 };  // GELU_Q8: fc1 writes its output as Q8 blocks

// Per-context mutable device state: the decode position and the token ring the kernels read
// their inputs from (so a captured graph can advance itself without host involvement).
struct DevState {
    int32_t n_past;   // tokens already in the KV cache
    int32_t n_gen;    // number of ids written to gen_ids by argmax_kernel
    int32_t causal;   // 0: reference behaviour (no intra-chunk mask, F1); 1: opt-in causal mask
    int32_t chunk;    // > 0: the N columns are consecutive reference chunks of this many tokens evaluated in one
                      // pass: column i sees the keys up to the end of ITS chunk (what n_batch-sized evals would show it)
    // followed in memory by: int32 tokens[n_positions]; int32 gen_ids[n_positions]
};
// keys visible to query column i of an N-column pass (biogpt.cpp:729-764 has no mask inside one eval, F1)
__device__ __forceinline__ int visible_keys(const DevState *st, int i, int N) {
    const int n_past = st->n_past;
    if (st->causal) return n_past + i + 1;
    const int nb = st->chunk;
    if (nb <= 0) return n_past + N;
    const int end = (i / nb + 1) * nb;
    return n_past + (end < N ? end : N);
}
__device__ __forceinline__ const int32_t *state_tokens(const DevState *st) { return reinterpret_cast<const int32_t *>(st + 1); }
__device__ __forceinline__ int32_t *state_tokens(DevState *st) { return reinterpret_cast<int32_t *>(st + 1); }

// Per-sequence decode state for batched multi-sequence decode (one activation column per sequence):
// its own position, current token and output count; kernels index it by column.
struct SeqState {
    int32_t n_past;   // position of this column's token
    int32_t token;
    int32_t n_gen;
    int32_t seq_id;   // which sequence's KV cache (batched decode: column i = sequence i)
    int32_t t_vis;    // keys this column may see; 0 = n_past + 1 (decode).  Prompt columns of several sequences
    int32_t pad[3];   //   travelling together carry the end of their own reference chunk here (col_mode = 1)
};

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

// ---- cross-lane primitives: DPP only (no LDS crossbar / ds_bpermute on the latency-critical paths) ----
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each 8
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <-> 15-i inside each 16
constexpr int DPP_WAVE_SHR1 = 0x138;       // lane i <- lane i-1 across the whole wave (GFX9)
constexpr int DPP_ROW_ROR4 = 0x124;        // lane i <- lane (i + 4) mod 16 inside each row of 16 (rotation: every lane gets a value)
constexpr int DPP_ROW_ROR8 = 0x128;

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// after the xor1/xor2 steps every lane of a quad holds the quad total, so the mirror permutations
// (which land in the other quad / the other half) complete the butterfly
__device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v)); v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
    return v;
}
__device__ __forceinline__ int group8_sum(int v) {
    v += dpp_i<DPP_QUAD_XOR1>(v); v += dpp_i<DPP_QUAD_XOR2>(v); v += dpp_i<DPP_ROW_HALF_MIRROR>(v);
    return v;
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// wave-wide sum of doubles, result uniform in all lanes
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_d<DPP_QUAD_XOR1>(v); v += dpp_d<DPP_QUAD_XOR2>(v); v += dpp_d<DPP_ROW_HALF_MIRROR>(v); v += dpp_d<DPP_ROW_MIRROR>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
// sum over aligned groups of `width` (power of two <= 64) lanes; result valid in every lane of the group
template <typename T>
__device__ __forceinline__ T wave_xor_sum(T v, int width) {
    for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-wide reductions: one DPP wave reduction + one LDS exchange; every thread gets the result.
__device__ __forceinline__ double block_sum_f64(double v, double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum_f64(v);
    if (nw == 1) return v;
    __syncthreads();  // scratch may still be read from a previous reduction
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; w++) t += scratch[w];
    return t;
}
__device__ __forceinline__ float block_max_f32(float v, float *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max_f32(v);
    if (nw == 1) return v;
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
                             float *x, int D, const SeqState *seq) {
    const int n = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    // seq != null: column n is an independent sequence with its own token / position
    const int32_t tok = seq ? seq[n].token : state_tokens(st)[n];
    const int32_t pos = (seq ? seq[n].n_past : st->n_past + n) + 2;
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
    float *kcache;       // layer slice of memory_k, head-major: [H][P][dk]
    float *vcache;
    int32_t dk;          // head size (power of two)
    int32_t dk_log2;
    int32_t P;           // n_positions
    int32_t D;
    float q_scale;
    const DevState *st;
    const SeqState *seq;       // batched decode: per-column position (null: columns are consecutive positions of one sequence)
    int32_t col_mode;          // 1: columns are prompt tokens of several sequences (use seq_id / t_vis of the column state)
    int64_t kv_seq_stride;     // floats between two sequences' caches
    // EPI_GELU
    const uint16_t *gelu_tab;
    // EPI_LOGITS: fused partial arg-max (N == 1 only); may be null
    float *pmax_val;
    int32_t *pmax_idx;
    // producer-quantized activations (single-token fast chain): 32 int8 per block + scale + block sum
    const int8_t *aq_q; const float *aq_d; const uint32_t *aq_s;   // PRO_Q8IN input
    int8_t *oq_q; float *oq_d; uint32_t *oq_s;                     // EPI_GELU_Q8 output
    uint32_t *lineage;   // EPI_LOGITS, single-token eval graphs: the call's lineage words (SEQ_* below); null: off
    DevState *st_adv;    // EPI_LOGITS, fused decode step: block 0 moves the device-side position / id count on by `adv`
    int32_t adv;         //   after the step (the sampler runs in the next step's first kernel); null / 0: off
    double inv_k;        // 1.0 / K
    int32_t k_pow2;      // K is a power of two: sum * inv_k == sum / K exactly (skips two f64 divisions)
    unsigned long long *tstamp;  // profiling: [2][grid][8] shader-clock stamps when dbg & 32
    int32_t dbg;         // profiling ablations (bench only; 0 in production): 1 skip x loads, 2 skip LN stats, 4 skip chain, 8 skip weights, 16 skip epilogue table
};

// Lineage of a replayed single-token eval (the five-launch layer captured as a graph; biogpt_hip_eval): the graph's FIRST node writes the call's sequence number, the
// LAST layer's last kernel forwards it when it starts, the lm_head kernel forwards that, the row's copy to pinned host memory carries it out.  A kernel that ran before its
// predecessor forwards the previous call's number: the host sees a row that is not this call's and repeats the call on eager launches (profiles/two_contexts_r4.txt: a
// replayed graph returned the row of the call before, bit for bit, beside another context's draining persistent kernel).
enum : int { SEQ_FETCHED = 0, SEQ_LAST_LAYER = 1, SEQ_LM_HEAD = 2 };
__device__ __forceinline__ void seq_forward(uint32_t *seq, int to) {
    if (seq != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) seq[to] = seq[to - 1];
}

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
    // (weights live in device memory: GLOBAL loads spelled out -- a DevMatrix copied out of a table in memory carries generic pointers, i.e. flat loads)
    typedef uint32_t lu_u4 __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(1))) lu_u4 *q = (const __attribute__((address_space(1))) lu_u4 *)(W.qs + idx * TI::qbytes);
    { const lu_u4 t = q[0]; u.q0 = make_uint4(t.x, t.y, t.z, t.w); }
    if (WT == W_Q8_0) { const lu_u4 t = q[1]; u.q1 = make_uint4(t.x, t.y, t.z, t.w); }
    if (TI::quant) {
        if (TI::q81) u.sc = ((const __attribute__((address_space(1))) uint32_t *)W.sc)[idx];
        else u.sc = ((const __attribute__((address_space(1))) uint16_t *)W.sc)[idx];
        if (WT == W_Q5_0 || WT == W_Q5_1) u.qh = ((const __attribute__((address_space(1))) uint32_t *)W.qh)[idx];
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

// A 4- / 5-bit unit unpacked IN REGISTERS to 32 bytes (q0: elements 0..15, q1: elements 16..31, still unsigned 0..15 / 0..31): the
// XCD-pipelined decode kernel does this while the weights wait for their layer's turn, so the dependent chain of a stage holds
// 8 dot4 per unit instead of 8 dot4 + 16 (Q4) or 64 (Q5) unpack instructions.  unit_dot_expanded == unit_dot_quant bit for bit.
template <int WT>
__device__ __forceinline__ void expand_unit(Unit<WT> &u) {
    static_assert(WT == W_Q4_0 || WT == W_Q4_1 || WT == W_Q5_0 || WT == W_Q5_1, "nibble formats");
    const uint32_t q[4] = {u.q0.x, u.q0.y, u.q0.z, u.q0.w};
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        lo[i] = q[i] & 0x0F0F0F0Fu;
        hi[i] = (q[i] >> 4) & 0x0F0F0F0Fu;
        if (WT == W_Q5_0 || WT == W_Q5_1) {
            lo[i] |= spread4(u.qh >> (4 * i));
            hi[i] |= spread4(u.qh >> (16 + 4 * i));
        }
    }
    u.q0 = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    u.q1 = make_uint4(hi[0], hi[1], hi[2], hi[3]);
}
// ... and the fp16 scale converted to f32 once (u.sc = bits of d; not for Q4_1 / Q5_1).  Q8_0 units take part too (settle_unit only
// converts their scale).
// Round 4: the symmetric nibble formats are settled SIGNED -- q - 8 (Q4_0) / q - 16 (Q5_0) as int8, so that the block dot is sum_j (q_j - 8) x_j itself, the same
// integer as sum_j q_j x_j - 8 sum_j x_j: the dot then needs no block sum of the activations (one instruction per unit, and the producers of Q8_0 activations no
// longer compute, publish and poll the sums).  Bytes 0 .. 15 (0 .. 31): (b + 0x78) ^ 0x80 = b - 8 ((b + 0x70) ^ 0x80 = b - 16) in every byte, no carry between bytes.
template <int WT>
__device__ __forceinline__ void settle_unit(Unit<WT> &u) {
    if constexpr (WT != W_Q8_0) expand_unit<WT>(u);
    if constexpr (WT == W_Q4_0 || WT == W_Q5_0) {
        constexpr uint32_t B = WT == W_Q4_0 ? 0x78787878u : 0x70707070u;
        u.q0.x = (u.q0.x + B) ^ 0x80808080u; u.q0.y = (u.q0.y + B) ^ 0x80808080u; u.q0.z = (u.q0.z + B) ^ 0x80808080u; u.q0.w = (u.q0.w + B) ^ 0x80808080u;
        u.q1.x = (u.q1.x + B) ^ 0x80808080u; u.q1.y = (u.q1.y + B) ^ 0x80808080u; u.q1.z = (u.q1.z + B) ^ 0x80808080u; u.q1.w = (u.q1.w + B) ^ 0x80808080u;
    }
    // Q4_1 / Q5_1 keep {d, m} packed: two f32 would be a tenth register per unit (8-9 spilled VGPRs measured in the 256-key variant)
    if constexpr (WT != W_Q4_1 && WT != W_Q5_1) u.sc = __float_as_uint(h2f((uint16_t)u.sc));
}
// unit_dot_quant on a settled unit, bit for bit
template <int WT>
__device__ __forceinline__ float unit_dot_settled(const Unit<WT> &u, const uint32_t *xq, float xd, float xs_f, int xs_i) {
    const uint4 xa = *reinterpret_cast<const uint4 *>(xq);
    const uint4 xb = *reinterpret_cast<const uint4 *>(xq + 4);
    int s = 0;
    s = dot4(u.q0.x, xa.x, s); s = dot4(u.q0.y, xa.y, s); s = dot4(u.q0.z, xa.z, s); s = dot4(u.q0.w, xa.w, s);
    s = dot4(u.q1.x, xb.x, s); s = dot4(u.q1.y, xb.y, s); s = dot4(u.q1.z, xb.z, s); s = dot4(u.q1.w, xb.w, s);
    if (WT == W_Q4_1 || WT == W_Q5_1) {
        const float d1 = h2f((uint16_t)(u.sc & 0xFFFFu)), m1 = h2f((uint16_t)(u.sc >> 16));
        return __fadd_rn(__fmul_rn(__fmul_rn(d1, xd), (float)s), __fmul_rn(m1, xs_f));
    }
    const float dw = __uint_as_float(u.sc);
    if (WT == W_Q8_0) return __fmul_rn((float)s, __fmul_rn(dw, xd));
    if (WT == W_Q4_0) return __fmul_rn(__fmul_rn((float)s, dw), xd);      // (signed units: s is already sum_j (q_j - 8) x_j; xs_i is not used)
    return __fmul_rn(__fmul_rn(dw, xd), (float)s);      // Q5_0
}

template <int WT>
__device__ __forceinline__ float unit_dot_expanded(const Unit<WT> &u, const uint32_t *xq, float xd, float xs_f, int xs_i) {
    const uint4 xa = *reinterpret_cast<const uint4 *>(xq);
    const uint4 xb = *reinterpret_cast<const uint4 *>(xq + 4);
    int s = 0;
    s = dot4(u.q0.x, xa.x, s); s = dot4(u.q0.y, xa.y, s); s = dot4(u.q0.z, xa.z, s); s = dot4(u.q0.w, xa.w, s);
    s = dot4(u.q1.x, xb.x, s); s = dot4(u.q1.y, xb.y, s); s = dot4(u.q1.z, xb.z, s); s = dot4(u.q1.w, xb.w, s);
    if (WT == W_Q4_0) {
        s -= 8 * xs_i;
        return __fmul_rn(__fmul_rn((float)s, h2f((uint16_t)u.sc)), xd);
    }
    if (WT == W_Q5_0) {
        s -= 16 * xs_i;
        return __fmul_rn(__fmul_rn(h2f((uint16_t)u.sc), xd), (float)s);
    }
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

// LDS layout of the mat-vec kernel: [xq | xd | xs | tail], all offsets multiples of 16 bytes.
//   quant types : xq = int8 activations [NC][K], xd = per-block scale [NC][K/32], xs = per-block
//                 integer sum (Q8_0) or d*sum (Q8_1), stored as raw 32-bit words
//   float types : xq = f32 activations [NC][K] (rounded through f16 for F16 weights)
//   part        : per wave [rpw][NC][stride] f32 block terms c_b of every row, summed IN BLOCK ORDER by one
//                 lane per (row, column) -- the association of the reference's scalar vec_dot loop
__host__ __device__ inline int matvec_part_stride(int upr) { return ((upr + 3) & ~3) + 4; }  // floats; 16-B aligned, bank-skewed
__host__ __device__ inline size_t matvec_smem_bytes(int wtype, int K, int NC, int upr, int rpw, int nwaves) {
    const bool quant = !(wtype == W_F32 || wtype == W_F16);
    size_t b = quant ? (size_t)NC * K : (size_t)NC * K * 4;
    b += 2 * (size_t)NC * (K / QK + 4) * 4;
    b += 256;                                                                   // tail (arg-max exchange)
    b += (size_t)nwaves * rpw * NC * (quant ? matvec_part_stride(upr) : 4) * 4;  // per-wave block terms
    return (b + 255) & ~(size_t)255;
}

// The activation column is held in registers as float4 "chunks": chunk id = jj*64 + lane.  For the
// LayerNorm prologue every wave loads the WHOLE column (statistics are computed redundantly per wave
// with DPP reductions -- no block barrier), but converts/quantizes only its share (jj % nwaves == wave).
// For the plain prologue a wave loads only its share.  KCH = register chunks per lane (4 or 16).
#ifdef BIOGPT_HIP_PROFILE_HOOKS   // make EXTRA=-DBIOGPT_HIP_PROFILE_HOOKS: in-kernel timestamps (tools/sweep_matvec.py)
#define BG_STAMP(k)                                                                                  \
    do {                                                                                            \
        if ((p.dbg & 32) && threadIdx.x == 0)                                                       \
            p.tstamp[(((size_t)((p.dbg >> 8) & 1) * gridDim.x + blockIdx.x) * 8) + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define BG_STAMP(k) do {} while (0)
#endif

template <int WT, int PRO, int EPI, int NC, int KCH>
__global__ __launch_bounds__(256) void matvec_kernel(const MatvecParams p) {
    using TI = TypeInfo<WT>;
    BG_STAMP(0);
    if (EPI == EPI_LOGITS) seq_forward(p.lineage, SEQ_LM_HEAD);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = p.W.K, M = p.W.M;
    const int nblk_pad = K / QK + 4;
    uint32_t *const s_xq = reinterpret_cast<uint32_t *>(smem_raw);
    float *const s_xd = reinterpret_cast<float *>(smem_raw + (TI::quant ? (size_t)NC * K : (size_t)NC * K * 4));
    uint32_t *const s_xs = reinterpret_cast<uint32_t *>(s_xd + (size_t)NC * nblk_pad);
    float *const s_tail = reinterpret_cast<float *>(s_xs + (size_t)NC * nblk_pad);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const int lpr = 1 << p.lpr_log2;
    const int rps = 64 >> p.lpr_log2;               // rows per wave step
    const int sub = lane & (lpr - 1);               // lane's position inside its row
    const int rsub = lane >> p.lpr_log2;            // row inside the step
    const int nsteps = p.rpw / rps;
    const int64_t row_base = ((int64_t)blockIdx.x * nwaves + wave) * p.rpw;
    const int col0 = blockIdx.y * NC;
    const int ncols = min(NC, p.N - col0);
    // finisher slots: output f = (row row_base + f % rpw, column col0 + f / rpw), f < rpw*ncols; lane l
    // takes f = l, l + 64, ... (one per lane for every BioGPT-base shape)
    const int pstride = TI::quant ? matvec_part_stride(p.upr) : 4;
    float *const s_part = s_tail + 64 + (size_t)wave * p.rpw * NC * pstride;
    const int nfin = p.rpw * ncols;
    const int f_row = lane % p.rpw, f_col = lane / p.rpw;
    const int64_t f_grow = row_base + f_row;
    const bool finisher = lane < nfin && f_grow < M;

    // ---- t = 0: issue every load that does not depend on another load ---------------------------
    // (a) weights of the first work item.  A work item = (row step, chunk of MAXIT units per lane);
    //     rows longer than MAXIT*LPR units (f16/f32 weights with K = d_ff) take several items per step.
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
            if (itg_ < p.nit && uu_ < p.upr && row_ < M && !(p.dbg & 8)) load_unit<WT>(dst[it], p.W, row_ * p.upr + uu_); \
        }                                                                                        \
    } while (0)
    BG_LOAD_ITEM(cur, 0);

    // (b) epilogue operands (bias / residual / decode position) of this lane's output element
    float e_bias = 0.0f, e_res = 0.0f;
    int e_npast = 0;
    if (finisher) {
        if (EPI != EPI_LOGITS) e_bias = p.bias[f_grow];
        if (EPI == EPI_RESID) e_res = p.resid[(size_t)(col0 + f_col) * p.ldr + f_grow];
        if (EPI == EPI_QKV) e_npast = p.st->n_past;
    }

    BG_STAMP(1);
    // ---- prologue: [LayerNorm] + activation conversion into LDS -------------------------------
    const int nchunks = K >> 2;
    const int njj = (nchunks + 63) >> 6;            // chunks per lane over the whole column
    // N = 1: the 4 waves share the column (each converts a quarter).  N > 1: a wave owns whole columns
    // (c = wave, wave + nwaves, ...) so the columns' load latencies and LayerNorms run side by side.
    constexpr bool COLWISE = NC > 1;
    for (int c = COLWISE ? wave : 0; c < ncols; c += COLWISE ? nwaves : 1) {
        const float4 *xcol = reinterpret_cast<const float4 *>(p.x + (size_t)(col0 + c) * p.ldx);
        float4 xr[KCH];
        // register slot i holds chunk jj(i): whole column (jj = i) for LN / column-wise; else the wave's share
#pragma unroll
        for (int i = 0; i < KCH; i++) {
            const int jj = (PRO == PRO_LN || COLWISE) ? i : wave + i * nwaves;
            const int ch = jj * 64 + lane;
            xr[i] = (jj < njj && ch < nchunks && !(p.dbg & 1)) ? xcol[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (p.dbg & 32) { asm volatile("" :: "v"(xr[0].x)); BG_STAMP(2); }
        if (PRO == PRO_LN && !(p.dbg & 2)) {
            // ggml_norm: mean and variance accumulate in double; y = (x-mean) * 1/sqrtf(var+eps); then *w, +b
            double s1 = 0.0;
#pragma unroll
            for (int i = 0; i < KCH; i++) {
                if (i < njj) s1 += ((double)xr[i].x + (double)xr[i].y) + ((double)xr[i].z + (double)xr[i].w);
            }
            s1 = wave_sum_f64(s1);
            const float mean = (float)(p.k_pow2 ? s1 * p.inv_k : s1 / (double)K);
            double s2 = 0.0;
#pragma unroll
            for (int i = 0; i < KCH; i++) {
                const bool live = i < njj && (i * 64 + lane) < nchunks;
                float4 v = xr[i];
                v.x = __fsub_rn(v.x, mean); v.y = __fsub_rn(v.y, mean); v.z = __fsub_rn(v.z, mean); v.w = __fsub_rn(v.w, mean);
                if (live)
                    s2 += ((double)__fmul_rn(v.x, v.x) + (double)__fmul_rn(v.y, v.y)) + ((double)__fmul_rn(v.z, v.z) + (double)__fmul_rn(v.w, v.w));
                xr[i] = v;
            }
            s2 = wave_sum_f64(s2);
            const float var = (float)(p.k_pow2 ? s2 * p.inv_k : s2 / (double)K);
            const float scale = 1.0f / sqrtf(__fadd_rn(var, p.eps));
#pragma unroll
            for (int i = 0; i < KCH; i++) {
                const int ch = i * 64 + lane;
                if (i < njj && ch < nchunks && (COLWISE || (i % nwaves) == wave)) {  // only what this wave converts
                    const float4 w = reinterpret_cast<const float4 *>(p.ln_w)[ch];
                    const float4 b = reinterpret_cast<const float4 *>(p.ln_b)[ch];
                    float4 v = xr[i];
                    v.x = __fadd_rn(__fmul_rn(w.x, __fmul_rn(v.x, scale)), b.x);
                    v.y = __fadd_rn(__fmul_rn(w.y, __fmul_rn(v.y, scale)), b.y);
                    v.z = __fadd_rn(__fmul_rn(w.z, __fmul_rn(v.z, scale)), b.z);
                    v.w = __fadd_rn(__fmul_rn(w.w, __fmul_rn(v.w, scale)), b.w);
                    xr[i] = v;
                }
            }
        }
        // convert this wave's share and publish it in LDS
#pragma unroll
        for (int i = 0; i < KCH; i++) {
            const int jj = (PRO == PRO_LN || COLWISE) ? i : wave + i * nwaves;
            const int ch = jj * 64 + lane;
            const bool mine = jj < njj && (COLWISE || PRO != PRO_LN || (i % nwaves) == wave);
            if (!mine) continue;                       // wave-uniform
            const bool live = ch < nchunks;
            const float4 v = xr[i];
            if (TI::quant) {
                // quantize_row_q8_0 / q8_1: a 32-block = 8 consecutive chunks = 8 consecutive lanes
                const float amax = group8_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                const float d = amax / 127.0f;
                const float id = (d != 0.0f) ? 1.0f / d : 0.0f;
                const int q0 = (int)roundf(__fmul_rn(v.x, id)), q1 = (int)roundf(__fmul_rn(v.y, id));
                const int q2 = (int)roundf(__fmul_rn(v.z, id)), q3 = (int)roundf(__fmul_rn(v.w, id));
                const int isum = group8_sum(q0 + q1 + q2 + q3);
                if (live) {
                    s_xq[(size_t)c * (K / 4) + ch] = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) |
                                                      ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
                    if ((ch & 7) == 0) {
                        const int b = ch >> 3;
                        if (TI::q81) {
                            s_xd[c * nblk_pad + b] = d;                                             // Q8_1 keeps f32 d
                            s_xs[c * nblk_pad + b] = __float_as_uint(__fmul_rn((float)isum, d));    // s = sum * d
                        } else {
                            s_xd[c * nblk_pad + b] = h2f(f2h(d));                                   // Q8_0 stores fp16 d
                            s_xs[c * nblk_pad + b] = (uint32_t)isum;
                        }
                    }
                }
            } else if (live) {
                float4 o = v;
                if (WT == W_F16) {  // src1 row converted to f16 (ggml_fp32_to_fp16_row)
                    o.x = h2f(f2h(o.x)); o.y = h2f(f2h(o.y)); o.z = h2f(f2h(o.z)); o.w = h2f(f2h(o.w));
                }
                reinterpret_cast<float4 *>(s_xq)[(size_t)c * (K / 4) + ch] = o;
            }
        }
    }
    BG_STAMP(3);
    __syncthreads();
    BG_STAMP(4);

    // ---- main loop over work items: block terms -> LDS ---------------------------------------------
    double accd[NC];   // float types: double accumulation (order-insensitive at f32 output precision)
    for (int w = 0; w < nitems; w++) {
        Unit<WT> nxt[MAXIT];
        if (w + 1 < nitems) BG_LOAD_ITEM(nxt, w + 1);
        const int stp = w / nitc, itc = w - stp * nitc;
        const int64_t row = row_base + (int64_t)stp * rps + rsub;
        const int wrow = stp * rps + rsub;               // row index inside this wave
        if (itc == 0) {
#pragma unroll
            for (int c = 0; c < NC; c++) accd[c] = 0.0;
        }
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
            const int itg = itc * MAXIT + it, uu = sub + itg * lpr;
            if (itg >= p.nit) continue;  // wave-uniform
            const bool live = uu < p.upr && row < M;
            if (!live) continue;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (c >= ncols) continue;
                if (TI::quant) {
                    const uint32_t *xq = s_xq + (size_t)c * (K / 4) + uu * 8;
                    const uint32_t xs = s_xs[c * nblk_pad + uu];
                    s_part[((size_t)wrow * NC + c) * pstride + uu] =
                        unit_dot_quant<WT>(cur[it], xq, s_xd[c * nblk_pad + uu], __uint_as_float(xs), (int)xs);
                } else {
                    const float *xf = reinterpret_cast<const float *>(s_xq) + (size_t)c * K + uu * TI::elems;
                    accd[c] += unit_dot_float<WT>(cur[it], xf);
                }
            }
        }
        if (!TI::quant && itc == nitc - 1) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const float r = (float)wave_xor_sum(accd[c], lpr);
                if (sub == 0 && row < M && c < ncols) s_part[((size_t)wrow * NC + c) * pstride] = r;
            }
        }
#pragma unroll
        for (int it = 0; it < MAXIT; it++) cur[it] = nxt[it];
    }
#undef BG_LOAD_ITEM
    BG_STAMP(5);
    // LDS is processed in order per wave: the finisher lanes' reads below see this wave's writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- finish: one lane per (row, column) adds the block terms in block order, then the epilogue ---
    float best_val = -INFINITY;
    int best_idx = 0x7fffffff;
    for (int f = lane; f < nfin; f += 64) {
        const int fr = f % p.rpw, fc = f / p.rpw;
        const int64_t grow = row_base + fr;
        if (grow >= M) continue;
        const float *part = s_part + ((size_t)fr * NC + fc) * pstride;
        float v;
        if (TI::quant) {
            // sumf = 0; for b: sumf += c_b   (ggml_vec_dot_q*_q8_* scalar order)
            float sumf = 0.0f;
            for (int b = 0; b < p.upr; b += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(part + b);
                sumf = __fadd_rn(sumf, t.x);
                if (b + 1 < p.upr) sumf = __fadd_rn(sumf, t.y);
                if (b + 2 < p.upr) sumf = __fadd_rn(sumf, t.z);
                if (b + 3 < p.upr) sumf = __fadd_rn(sumf, t.w);
            }
            v = sumf;
        } else {
            v = part[0];
        }
        const int r = (int)grow, col = col0 + fc;
        const bool first = (f == lane) && finisher;  // operands were fetched at kernel entry
        float bias = e_bias, res = e_res;
        int npast = e_npast;
        if (!first) {
            if (EPI != EPI_LOGITS) bias = p.bias[r];
            if (EPI == EPI_RESID) res = p.resid[(size_t)col * p.ldr + r];
            if (EPI == EPI_QKV) npast = p.st->n_past;
        }
        if (EPI == EPI_QKV) {
            v = __fadd_rn(bias, v);
            const int which = r / p.D, rr = r - which * p.D;
            if (which == 0) {
                p.q_out[(size_t)col * p.D + rr] = __fmul_rn(v, p.q_scale);
            } else {
                float *cache = (which == 1) ? p.kcache : p.vcache;
                const int hh = rr / p.dk, dd = rr - hh * p.dk;  // head-major cache: [H][P][dk]
                cache[((size_t)hh * p.P + (npast + col)) * p.dk + dd] = v;
            }
        } else if (EPI == EPI_RESID) {
            v = __fadd_rn(v, bias);
            p.out[(size_t)col * p.ldo + r] = __fadd_rn(v, res);
        } else if (EPI == EPI_GELU) {
            v = __fadd_rn(bias, v);
            p.out[(size_t)col * p.ldo + r] = (p.dbg & 16) ? v : h2f(p.gelu_tab[f2h(v)]);
        } else {
            p.out[(size_t)col * p.ldo + r] = v;
            if (fc == 0 && (v > best_val || (v == best_val && r < best_idx))) { best_val = v; best_idx = r; }
        }
    }

    BG_STAMP(6);
    if (EPI == EPI_LOGITS && p.pmax_val != nullptr) {
        // per-block partial arg-max (lowest index wins ties), finished by argmax_kernel
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best_val, off, 64);
            const int oi = __shfl_xor(best_idx, off, 64);
            if (ov > best_val || (ov == best_val && oi < best_idx)) { best_val = ov; best_idx = oi; }
        }
        float *sv = s_tail;
        int *si = reinterpret_cast<int *>(s_tail) + 8;
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
    int32_t t_cap;                // launch-time upper bound of the context (fast kernel load bound)
    int8_t *oq_q; float *oq_d; uint32_t *oq_s;  // optional Q8 copy of the output for the fast out_proj (null: off)
    int32_t q81;                  // 1: Q8_1 activation form (Q4_1 / Q5_1 weights), 0: Q8_0
    const SeqState *seq;          // batched decode: query row i belongs to sequence i (own cache + position)
    int32_t col_mode;             // 1: query rows are prompt tokens of several sequences (seq_id / t_vis per row)
    int64_t kv_seq_stride;
    float *sp_scores; float *sp_max; double *sp_pv; int32_t n_split;   // key-split decode attention scratch: [H][P], [H][16], [H][16][64]
    unsigned long long *tstamp;  // profiling (dbg & 32): [16 waves][8] stamps of block (0,0)
    int32_t dbg;
};

constexpr int ATTN_MAXK = 4;  // keys per thread held in registers: T <= ATTN_MAXK * blockDim

// One workgroup per (head, query token).  scores -> softmax -> PV in the order of biogpt.cpp:741-764:
//   S_j = K_j . q ; p_j = f16tab_exp(S_j - max) ; p_j *= (float)(1/sum_double) ; o_d = sum_j V_jd * p_j
// A thread owns whole keys (no cross-lane work in QK^T): its 16-byte loads of a 4*dk-byte key row are
// all in flight at once and the dot runs in the reference's element order with a double accumulator.
__global__ __launch_bounds__(1024) void attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int h = blockIdx.x, i = blockIdx.y;
    const int T = visible_keys(p.st, i, p.N);
    const int dk = p.dk, D = p.D;
    const int tid = threadIdx.x, nt = blockDim.x;

    float *S = reinterpret_cast<float *>(smem_raw);                               // [P] probabilities
    double *red = reinterpret_cast<double *>(smem_raw + (((size_t)p.P * 4 + 15) & ~(size_t)15));  // [16]
    double *pv = red + 16;                                                        // [nt]

    const float *__restrict__ qrow = p.q + (size_t)i * D + (size_t)h * dk;        // wave-uniform address
    const float *__restrict__ kbase = p.kcache + (size_t)h * p.P * dk;   // head-major cache: [H][P][dk]
    const float *__restrict__ vbase = p.vcache + (size_t)h * p.P * dk;

    // ---- scores ----
    float sc[ATTN_MAXK];
#pragma unroll
    for (int kk = 0; kk < ATTN_MAXK; kk++) {
        const int j = tid + kk * nt;
        sc[kk] = -INFINITY;
        if (j < T) {
            const float4 *kr = reinterpret_cast<const float4 *>(kbase + (size_t)j * dk);
            double acc = 0.0;
            for (int d4 = 0; d4 < (dk >> 2); d4++) {
                const float4 kv = kr[d4];
                const float4 qv = reinterpret_cast<const float4 *>(qrow)[d4];
                acc += (double)__fmul_rn(kv.x, qv.x); acc += (double)__fmul_rn(kv.y, qv.y);
                acc += (double)__fmul_rn(kv.z, qv.z); acc += (double)__fmul_rn(kv.w, qv.w);
            }
            sc[kk] = (float)acc;
        }
    }

    // ---- softmax (ggml_soft_max: fp16-table exp, double sum, scale by (float)(1/sum)) ----
    float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
    mx = block_max_f32(mx, reinterpret_cast<float *>(red));
    double sum = 0.0;
#pragma unroll
    for (int kk = 0; kk < ATTN_MAXK; kk++) {
        const int j = tid + kk * nt;
        if (j < T) {
            const float val = h2f(p.exp_tab[f2h(__fsub_rn(sc[kk], mx))]);
            sc[kk] = val;
            sum += (double)val;
        }
    }
    sum = block_sum_f64(sum, red);
    const float inv = (float)(1.0 / sum);
#pragma unroll
    for (int kk = 0; kk < ATTN_MAXK; kk++) {
        const int j = tid + kk * nt;
        if (j < T) S[j] = __fmul_rn(sc[kk], inv);
    }
    __syncthreads();

    // ---- PV: thread = (slice, d); each slice strides over the keys; double accumulation ----
    const int nsl = nt / dk;
    const int d = tid % dk, sl = tid / dk;
    double acc = 0.0;
    if (sl < nsl) {
        int j = sl;
        for (; j + 3 * nsl < T; j += 4 * nsl) {
            const float v0 = vbase[(size_t)j * dk + d], v1 = vbase[(size_t)(j + nsl) * dk + d];
            const float v2 = vbase[(size_t)(j + 2 * nsl) * dk + d], v3 = vbase[(size_t)(j + 3 * nsl) * dk + d];
            acc += (double)__fmul_rn(v0, S[j]); acc += (double)__fmul_rn(v1, S[j + nsl]);
            acc += (double)__fmul_rn(v2, S[j + 2 * nsl]); acc += (double)__fmul_rn(v3, S[j + 3 * nsl]);
        }
        for (; j < T; j += nsl) acc += (double)__fmul_rn(vbase[(size_t)j * dk + d], S[j]);
    }
    pv[tid] = acc;
    __syncthreads();
    if (tid < dk) {
        double t = 0.0;
        for (int s2 = 0; s2 < nsl; s2++) t += pv[s2 * dk + tid];
        p.out[(size_t)i * D + (size_t)h * dk + tid] = (float)t;
    }
}

__host__ __device__ inline size_t attn_smem_bytes(int P, int dk, int nthreads) {
    return (((size_t)P * 4 + 15) & ~(size_t)15) + 16 * 8 + (size_t)nthreads * 8 + 64;
}

// flat reference view [layer][pos][d_model] of a range of the head-major cache [layer][head][pos][dk] (biogpt_hip_read_kv)
__global__ __launch_bounds__(256) void kv_gather_kernel(const float *cache, float *out, unsigned long long offset, unsigned long long count,
                                                        int P, int D, int H, int dk) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const unsigned long long e = offset + i;
    const unsigned long long l = e / ((unsigned long long)P * D), pos = (e / D) % P, dm = e % D, h = dm / dk, dd = dm % dk;
    out[i] = cache[((l * H + h) * P + pos) * dk + dd];
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


// Batched decode sampler: one workgroup per sequence row of logits[rows][ld]; arg-max (lowest id wins ties),
// appended to the sequence's id list, becomes its next input token; advance = 1 moves its position on.
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const float *logits, int ld, int n_vocab, SeqState *seq, int seq0,
                                                           int32_t *gen_ids, int gen_stride, int advance) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    const float *lg = logits + (size_t)row * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = tid; k < n_vocab; k += blockDim.x) {
        const float v = lg[k];
        if (v > bv) { bv = v; bi = k; }   // ascending k: the first maximum is kept
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { sv[wv] = bv; si[wv] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < nw; w++)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        SeqState *s = seq + seq0 + row;
        gen_ids[(size_t)(seq0 + row) * gen_stride + s->n_gen] = bi;
        s->n_gen += 1;
        s->token = bi;
        s->n_past += advance;
    }
}

}  // namespace bgk
