// Single-token decode at BioGPT-base shapes (d_model 1024, d_ff 4096, 16 heads of 64, block-quantized weights,
// contexts up to 256 keys): five launches per layer like the first chain (kernels_fast.hip.h), but with the embedding
// and the sampler folded into the first one (122 -> 121 ... -> 5 x 24 + 1 launches per token), every kernel built for
// the SHORTEST dependent-instruction chain, and 16-wave workgroups that share the LayerNorm statistics.
//
// What bounds this path, measured on the MI355X (tools/microbench4-7.hip, profiles/microbench*_r2.txt):
//   * a dependent kernel boundary: 1.58 us for an empty kernel whatever its shape; last exit -> next first entry is
//     1.3-1.5 us for the real kernels (wall clock, profiles/decode_*_timeline_r2.txt);
//   * a wave issues ONE instruction per 4 cycles (2048 straight-line VALU instructions: 8324 cycles warm, 8750 cold --
//     the instruction cache is not the limit): a kernel's body is its per-wave instruction count, so the work is spread
//     over 16 waves per workgroup and nothing is recomputed per wave;
//   * cross-workgroup hand-offs INSIDE a launch are dearer than the boundary they would replace: last-arriver ticket
//     (sc1 stores -> drain -> agent atomic -> sc1 loads) +1.9 us at 16 arrivers; device-wide flag barrier 3.3 us at
//     256 workgroups (1.25 us at 32);
//   * one workgroup can pull ~100 GB/s: fusions that put a head's q/k/v rows (110 KB) + K/V + out_proj slice through
//     ONE compute unit were built and measured (13.3 us per layer for that kernel; profiles/decode_fused_per_head_*):
//     slower than separate launches.  Hence: every launch spreads its bytes over >= 32 compute units.
//
//   dec_qkv_kernel    [embedding (+ arg-max of the previous token's logits partials)] -> LayerNorm -> Q8 -> q/k/v rows
//                     -> Q scale, KV append                                                    biogpt.cpp:664-727
//   dec_attn_kernel   one workgroup per head: scores, fp16-table softmax, PV, Q8 of the 64 outputs  biogpt.cpp:729-764
//   dec_oproj_kernel  out_proj + bias + residual                                                    biogpt.cpp:767-772
//   dec_fc1_kernel    LayerNorm -> Q8 -> fc1 rows -> bias -> GELU table -> Q8 block(s) for fc2       biogpt.cpp:777-787
//   dec_fc2_kernel    fc2 + bias + residual, one row per wave                                        biogpt.cpp:790-795
//   (lm_head stays matvec_fast_kernel<EPI_LOGITS>; its block 0 advances the device-side position.)
//
// Arithmetic per element is that of kernels_fast.hip.h (and of the oracle): same Q8 activations, same integer block
// dots and scale expressions, same in-order f32 block sums, same fp16 tables, double LayerNorm / softmax / PV sums.
#pragma once

#include "kernels_fast.hip.h"

namespace bgk {

// ---- 32-lane and 64-lane exchanges on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap) -----------------
// permlane16_swap(v, v): result[0] holds, in lanes 16-31 of every 32, the values of lanes 0-15, result[1] holds in
// lanes 0-15 the values of lanes 16-31; the other halves are unchanged -- so op(result[0], result[1]) = op(v, v of
// lane ^ 16) in every lane.  Likewise permlane32_swap for lane ^ 32.
__device__ __forceinline__ float max_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ int sum_xor16(int v) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)r[0] + (int)r[1];
}
// max / integer sum over aligned groups of 32 lanes, result in every lane of the group
__device__ __forceinline__ float group32_max(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    return max_xor16(v);
}
__device__ __forceinline__ int group32_sum(int v) {
    v += dpp_i<DPP_QUAD_XOR1>(v); v += dpp_i<DPP_QUAD_XOR2>(v);
    v += dpp_i<DPP_ROW_HALF_MIRROR>(v); v += dpp_i<DPP_ROW_MIRROR>(v);
    return sum_xor16(v);
}

// kernel outputs that the NEXT launch reads.  Write-through (sc1) stores were measured against plain stores: 1.60 vs
// 1.77 us per producer launch in isolation (microbench6), but no gain in the decode chain (462 vs 459 us per token), so
// plain stores stay; -DBIOGPT_HIP_SC1_STORES rebuilds the other arm.
#ifdef BIOGPT_HIP_SC1_STORES
#define DEC_STORE_F32(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define DEC_STORE_F32(ptr, val) (*(ptr) = (val))
#endif

#ifdef BIOGPT_HIP_PROFILE_HOOKS   // make EXTRA=-DBIOGPT_HIP_PROFILE_HOOKS: per-segment shader-clock stamps of workgroup 0, wave 0
#define DEC_STAMP(k) do { if (p.tstamp && blockIdx.x == 0 && threadIdx.x == 0) p.tstamp[(k)] = __builtin_readcyclecounter(); } while (0)
// dbg & 64: entry / exit of EVERY workgroup on the constant 100 MHz clock, [slot][1024 workgroups][2] after the segment stamps
#define DEC_WALL(which) do { if (p.wall && threadIdx.x == 0) \
        p.wall[((size_t)p.wall_slot * 1024 + blockIdx.x) * 2 + (which)] = wall_clock64(); } while (0)
#else
#define DEC_STAMP(k) do {} while (0)
#define DEC_WALL(which) do {} while (0)
#endif

// The two divisions of quantize_row_q8_0 / q8_1 (d = amax / 127, id = 1 / d) sit on the dependent chain of every LayerNorm and of
// every activation hand-over.  Three-instruction forms that equal the IEEE quotients BIT FOR BIT, checked exhaustively on the
// MI355X (tools/microbench12.hip -> profiles/microbench12_q8_divisions_r2.txt): d' = fma(fma(-127, q, a), C, q) with q = a * C,
// C = RN(1 / 127), matches for every finite a >= 0 (2,139,095,040 values); id' = fma(fma(-d, r, 1), r, r) with r = v_rcp_f32(d)
// matches for 2^-100 <= a < 2^100 (it differs for a < 1.5e-36).  Outside that range the divisions themselves are used.
__device__ __forceinline__ void q8_scales(float amax, float &d, float &id) {
    if (amax >= 0x1p-100f && amax < 0x1p100f) {
        const float C = 1.0f / 127.0f;
        const float q = __fmul_rn(amax, C);
        d = __fmaf_rn(__fmaf_rn(-127.0f, q, amax), C, q);
        const float r = __builtin_amdgcn_rcpf(d);
        id = __fmaf_rn(__fmaf_rn(-d, r, 1.0f), r, r);
    } else {
        d = amax / 127.0f;
        id = (d != 0.0f) ? 1.0f / d : 0.0f;
    }
}

// LayerNorm (ggml_norm + affine, double statistics) and Q8_0 / Q8_1 quantization of ONE 1024-element column inside a
// 1024-thread workgroup.  Measured (profiles/decode_5kernel_timeline_r2.txt): with all 16 waves taking part (one element per
// thread) the double-precision adds of 16 waves contend for the SIMDs and every barrier waits for the slowest wave --
// 4300 cycles; so waves 0-3 do it alone (thread t < 256 holds elements 4t .. 4t+3, 8 lanes = one Q8 block; the
// arithmetic of lnq_kernel) while waves 4-15 are parked at the barriers.  Leaves the 32 activation blocks in LDS
// (s_xq / s_xd / s_xs); s_red: 8 doubles.  Every thread of the workgroup must call it; ends with a workgroup barrier.
#ifdef BIOGPT_HIP_PROFILE_HOOKS
#define LN_STAMP(k) do { if (ts && blockIdx.x == 0 && threadIdx.x == 0) ts[(k)] = __builtin_readcyclecounter(); } while (0)
#else
#define LN_STAMP(k) do {} while (0)
#endif
// SUM = false: the consumer's units are settled signed (kernels.hip.h, settle_unit) and need no block sums of Q8_0 activations (s_xs is then not written)
template <bool Q81, bool SUM = true>
__device__ __forceinline__ void ln4_q8_1024(float4 v, float4 lw, float4 lb, float eps, double *s_red, uint32_t *s_xq, float *s_xd,
                                            uint32_t *s_xs, unsigned long long *ts = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool worker = tid < 256;
    const double inv_k = 1.0 / 1024.0;
#ifdef BIOGPT_HIP_PROFILE_HOOKS
    asm volatile("" :: "v"(v.x));     // the column has arrived
#endif
    LN_STAMP(8);
#ifndef LN_FAKE_MEAN      // (timing experiment only, WRONG results: the mean taken as known -- the upper bound of what "LayerNorm partial sums from the producers" could save)
    if (worker) {
        const double s1 = wave_sum_f64(((double)v.x + (double)v.y) + ((double)v.z + (double)v.w));
        if (lane == 0) s_red[wave] = s1;
    }
    LN_STAMP(9);
    __syncthreads();
#endif
    LN_STAMP(10);
    float mean = 0.0f;
    float a = 0.0f, b = 0.0f, c = 0.0f, d4 = 0.0f;
    if (worker) {
#ifdef LN_FAKE_MEAN
        mean = 1.0e-3f * (float)inv_k;
#else
        mean = (float)(((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * inv_k);
#endif
        a = __fsub_rn(v.x, mean); b = __fsub_rn(v.y, mean); c = __fsub_rn(v.z, mean); d4 = __fsub_rn(v.w, mean);
        const double s2 = wave_sum_f64(((double)__fmul_rn(a, a) + (double)__fmul_rn(b, b)) + ((double)__fmul_rn(c, c) + (double)__fmul_rn(d4, d4)));
        if (lane == 0) s_red[4 + wave] = s2;
    }
    LN_STAMP(11);
    __syncthreads();
    LN_STAMP(12);
    if (worker) {
        const float var = (float)(((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])) * inv_k);
        const float scale = 1.0f / sqrtf(__fadd_rn(var, eps));
        a = __fadd_rn(__fmul_rn(lw.x, __fmul_rn(a, scale)), lb.x);
        b = __fadd_rn(__fmul_rn(lw.y, __fmul_rn(b, scale)), lb.y);
        c = __fadd_rn(__fmul_rn(lw.z, __fmul_rn(c, scale)), lb.z);
        d4 = __fadd_rn(__fmul_rn(lw.w, __fmul_rn(d4, scale)), lb.w);
        LN_STAMP(13);
        // quantize_row_q8_0 / q8_1: one block = 8 consecutive lanes x 4 values
        const float amax = group8_max(fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d4))));
        float d, id;
        q8_scales(amax, d, id);
        const int q0 = (int)roundf(__fmul_rn(a, id)), q1 = (int)roundf(__fmul_rn(b, id));
        const int q2 = (int)roundf(__fmul_rn(c, id)), q3 = (int)roundf(__fmul_rn(d4, id));
        s_xq[tid] = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
        if constexpr (Q81 || SUM) {
            const int isum = group8_sum(q0 + q1 + q2 + q3);
            if ((lane & 7) == 0) {
                const int blk = tid >> 3;
                if (Q81) { s_xd[blk] = d; s_xs[blk] = __float_as_uint(__fmul_rn((float)isum, d)); }
                else { s_xd[blk] = h2f(f2h(d)); s_xs[blk] = (uint32_t)isum; }
            }
        } else {
            if ((lane & 7) == 0) s_xd[tid >> 3] = h2f(f2h(d));
        }
    }
    LN_STAMP(14);
    __syncthreads();
}

// 32 block terms of one row, added in block order (the association of the reference's scalar ggml_vec_dot_q*_q8_*)
__device__ __forceinline__ float sum32_in_order(const float *part) {
    const float4 *p4 = reinterpret_cast<const float4 *>(part);
    float4 t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = p4[j];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        s = __fadd_rn(s, t[j].x); s = __fadd_rn(s, t[j].y); s = __fadd_rn(s, t[j].z); s = __fadd_rn(s, t[j].w);
    }
    return s;
}

// quantize_row_q8_0 / q8_1 of 32 values held one per lane by an aligned group of 32 lanes
__device__ __forceinline__ void q8_block32(float v, bool q81, int8_t &q_out, float &d_out, uint32_t &s_out, bool want_sum = true) {
    const float amax = group32_max(fabsf(v));
    float d, id;
    q8_scales(amax, d, id);
    const int q = (int)roundf(__fmul_rn(v, id));
    q_out = (int8_t)q;
    if (!q81 && !want_sum) { d_out = h2f(f2h(d)); s_out = 0u; return; }      // the consumer's units are settled signed (kernels.hip.h): no block sum
    const int isum = group32_sum(q);
    if (q81) { d_out = d; s_out = __float_as_uint(__fmul_rn((float)isum, d)); }
    else { d_out = h2f(f2h(d)); s_out = (uint32_t)isum; }
}

// ---- A: [embedding (+ sampler of the previous token)] -> LayerNorm -> Q8 -> q/k/v rows -> KV append ---------------
struct DecQkvParams {
    // layer input: tok_src 0 = x[1024] from memory; 1 = embedding of the state's token; 2 = arg-max of the lm_head
    // partials of the previous token (recorded in the state), then its embedding.  tok_src != 0: workgroup 0 writes x_out.
    const float *x;
    float *x_out;
    DevMatrix tok_emb, pos_emb;
    float embed_scale;
    int32_t tok_src;
    const float *pmax_val; const int32_t *pmax_idx; int32_t nparts;
    DevState *st;
    int32_t n_positions, n_vocab;
    const float *ln_w, *ln_b;
    float eps;
    DevMatrix Wqkv;            // [3*1024][1024] row-stacked q, k, v
    const float *bqkv;
    float q_scale;
    float *q_out;              // [1024] scaled queries
    float *kcache, *vcache;    // layer slice, head-major [H][P][64]
    int32_t P;
    unsigned long long *tstamp;
    unsigned long long *wall;
    int32_t wall_slot;
};

constexpr int DEC_PS = 36;     // floats between two rows of block terms in LDS (32 + 4: float4 aligned, skewed)
__host__ __device__ inline size_t dec_qkv_smem_bytes() { return 1536 + 128 + 128 + (size_t)16 * 2 * DEC_PS * 4; }

// NW waves per workgroup (>= 4: the LayerNorm workers), 2 rows per wave: grid = 3072 / (2 NW); wave w owns rows 2 NW b + 2 w, +1
template <int WT, int NW>
__global__ __launch_bounds__(NW * 64) void dec_qkv_kernel(const DecQkvParams p) {
    static_assert(NW >= 4 && NW <= 16, "waves per workgroup");
    using TI = TypeInfo<WT>;
    constexpr int D = 1024, DK = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *const s_xq = reinterpret_cast<uint32_t *>(smem);
    float *const s_xd = reinterpret_cast<float *>(smem + 1024);
    uint32_t *const s_xs = reinterpret_cast<uint32_t *>(smem + 1152);
    double *const s_red = reinterpret_cast<double *>(smem + 1280);
    float *const s_amv = reinterpret_cast<float *>(smem + 1536);
    int *const s_ami = reinterpret_cast<int *>(smem + 1536 + 128);
    float *const s_part = reinterpret_cast<float *>(smem + 1536 + 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 31, rsub = lane >> 5;
    DEC_STAMP(0);
    DEC_WALL(0);

    // ---- t = 0: the column first (the LayerNorm chain starts from it), then this lane's weight unit ----
    const int n_past = p.st->n_past;
    const bool worker = tid < 256;                                        // waves 0-3 hold the column, 4 elements per thread
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), lnw = xv, lnb = xv;
    if (worker) {
        if (p.tok_src == 0) xv = reinterpret_cast<const float4 *>(p.x)[tid];
        lnw = reinterpret_cast<const float4 *>(p.ln_w)[tid];
        lnb = reinterpret_cast<const float4 *>(p.ln_b)[tid];
    }
    const int row = blockIdx.x * 2 * NW + wave * 2 + rsub;                // row of the stacked [q; k; v] matrix
    Unit<WT> wq;
    load_unit<WT>(wq, p.Wqkv, (int64_t)row * 32 + sub);
    const int frow = blockIdx.x * 2 * NW + wave * 2 + (lane & 1);         // finisher lanes 0, 1
    float e_bias = 0.0f;
    if (lane < 2) e_bias = p.bqkv[frow];
    if (p.tok_src != 0) {
        int tok;
        if (p.tok_src == 2) {
            // greedy sampler of the PREVIOUS token (main.cpp:109-128, top_k = 1): finish the arg-max over the lm_head
            // kernel's per-workgroup partials (lowest id wins ties); workgroup 0 records it
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int k = tid; k < p.nparts; k += NW * 64) {
                const float v = p.pmax_val[k];
                const int ix = p.pmax_idx[k];
                if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(bv, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) { s_amv[wave] = bv; s_ami[wave] = bi; }
            __syncthreads();
            bv = s_amv[0]; bi = s_ami[0];
#pragma unroll
            for (int w = 1; w < NW; w++)
                if (s_amv[w] > bv || (s_amv[w] == bv && s_ami[w] < bi)) { bv = s_amv[w]; bi = s_ami[w]; }
            tok = bi;
            if (tok < 0 || tok >= p.n_vocab) tok = 0;      // partials never written (first replay of a fresh context)
            if (blockIdx.x == 0 && tid == 0) {
                int32_t *tokens = state_tokens(p.st);
                const int g = p.st->n_gen;
                if (g < p.n_positions) tokens[p.n_positions + g] = tok;
                tokens[0] = tok;
            }
        } else {
            tok = state_tokens(p.st)[0];
        }
        // biogpt.cpp:664-686: embed_tokens[tok] * sqrt(D) + embed_positions[n_past + 2]
        if (worker) {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                e[j] = __fadd_rn(__fmul_rn(dequant_elem(p.tok_emb, tok, 4 * tid + j), p.embed_scale), dequant_elem(p.pos_emb, n_past + 2, 4 * tid + j));
            xv = make_float4(e[0], e[1], e[2], e[3]);
            if (blockIdx.x == 0) reinterpret_cast<float4 *>(p.x_out)[tid] = xv;
        }
    }
    DEC_STAMP(1);
    if (TI::q81) ln4_q8_1024<true>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs, p.tstamp);
    else ln4_q8_1024<false>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs, p.tstamp);
    DEC_STAMP(2);
    {
        uint32_t ax[8];
        const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
        ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
        const uint32_t axs = s_xs[sub];
        float *const part = s_part + wave * 2 * DEC_PS;
        part[rsub * DEC_PS + sub] = unit_dot_quant<WT>(wq, ax, s_xd[sub], __uint_as_float(axs), (int)axs);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 2) {
            const float v = __fadd_rn(e_bias, sum32_in_order(part + lane * DEC_PS));
            const int which = frow >> 10, rr = frow & (D - 1);
            if (which == 0) {
                DEC_STORE_F32(p.q_out + rr, __fmul_rn(v, p.q_scale));   // Q scaled AFTER the bias (biogpt.cpp:708-710)
            } else {                                                      // KV append (biogpt.cpp:721-727), head-major cache
                float *cache = (which == 1) ? p.kcache : p.vcache;
                DEC_STORE_F32(cache + ((size_t)(rr >> 6) * p.P + n_past) * DK + (rr & 63), v);
            }
        }
    }
    DEC_STAMP(3);
    DEC_WALL(1);
}

// ---- B: attention of one head over T = n_past + 1 keys (biogpt.cpp:729-764, no mask needed for one query) --------
struct DecAttnParams {
    const float *q;            // [1024] scaled queries
    const float *kcache, *vcache;
    const DevState *st;
    int32_t P, t_cap;          // t_cap: launch-time bound of the context (<= 256) for the cache loads
    const uint16_t *exp_tab;
    int8_t *oq_q; float *oq_d; uint32_t *oq_s;   // attention output as 32 Q8 blocks (out_proj's activation row)
    float *att_out;            // optional F32 copy [1024] (null: off)
    int32_t q81;
    unsigned long long *tstamp;
    unsigned long long *wall;
    int32_t wall_slot;
};

// grid = 16 heads, NW waves.  Scores: LPK lanes per key (16 / 8 / 4 for contexts up to 64 / 128 / 256 keys; 2 lanes per key
// for 512 keys was measured: 256 KB of K / V through one compute unit, 633-649 us per token against 599 with the key-split
// launches of kernels_fast.hip.h), every
// lane 64 / LPK dims, double partial sums reduced with DPP inside the key's lane group -- all 1024 threads work whatever
// the context.  PV: 16 key slices x 64 dims.  The query row goes through LDS once (16 lanes load it) instead of being
// fetched by every quad (1024 x 64 B through one texture addresser).
template <int LPK, int NW>
__global__ __launch_bounds__(NW * 64) void dec_attn_kernel(const DecAttnParams p) {
    constexpr int DK = 64, NF4 = 16 / LPK;          // float4 per lane of a key row
    static_assert(LPK == 4 || LPK == 8 || LPK == 16, "lanes per key");
    __shared__ __attribute__((aligned(16))) float s_q[DK];
    __shared__ float s_S[NW * 64 / LPK];          // contexts up to NW * 64 / LPK keys
    __shared__ float s_redf[NW];
    __shared__ double s_redd[NW];
    __shared__ double s_pv[NW * 64];
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksub = tid & (LPK - 1), kidx = tid / LPK;
    const int dd = tid & (DK - 1), sl = tid >> 6;
    const int t_cap = p.t_cap;
    DEC_STAMP(0);
    DEC_WALL(0);
    const int n_past = p.st->n_past;
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 16) q4 = reinterpret_cast<const float4 *>(p.q + h * DK)[tid];
    float4 kr[NF4];
    if (kidx < t_cap) {
        const float4 *kbase = reinterpret_cast<const float4 *>(p.kcache + (size_t)h * p.P * DK) + (size_t)kidx * (DK / 4) + ksub;
#pragma unroll
        for (int m = 0; m < NF4; m++) kr[m] = kbase[LPK * m];       // float4 #(LPK*m + ksub): one load of a lane group covers 16*LPK contiguous bytes
    }
    constexpr int NV = 64 / LPK;                     // keys per slice: t_cap <= NW * NV
    float vr[NV];
    {
        const float *vbase = p.vcache + (size_t)h * p.P * DK + dd;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int j = sl + NW * k;
            if (j < t_cap) vr[k] = vbase[(size_t)j * DK];
        }
    }
    if (tid < 16) reinterpret_cast<float4 *>(s_q)[tid] = q4;
    const int T = n_past + 1;
    __syncthreads();
    DEC_STAMP(1);
    float sc = -INFINITY;
    if ((tid & ~63) < LPK * T) {        // whole waves past the context skip the double-precision work
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int m = 0; m < NF4; m++) {
            const float4 qm = *reinterpret_cast<const float4 *>(s_q + 4 * (LPK * m + ksub));
            a0 += (double)__fmul_rn(kr[m].x, qm.x); a1 += (double)__fmul_rn(kr[m].y, qm.y);
            a2 += (double)__fmul_rn(kr[m].z, qm.z); a3 += (double)__fmul_rn(kr[m].w, qm.w);
        }
        double acc = (a0 + a1) + (a2 + a3);
        acc += dpp_d<DPP_QUAD_XOR1>(acc);
        acc += dpp_d<DPP_QUAD_XOR2>(acc);
        if (LPK >= 8) acc += dpp_d<DPP_ROW_HALF_MIRROR>(acc);
        if (LPK >= 16) acc += dpp_d<DPP_ROW_MIRROR>(acc);
        if (kidx < T) sc = (float)acc;
    }
    float mx = wave_max_f32(sc);
    if (lane == 0) s_redf[wave] = mx;
    __syncthreads();
    mx = s_redf[0];
#pragma unroll
    for (int w = 1; w < NW; w++) mx = fmaxf(mx, s_redf[w]);
    DEC_STAMP(2);
    double sum = 0.0;
    if (kidx < T && ksub == 0) {
        const float val = h2f(p.exp_tab[f2h(__fsub_rn(sc, mx))]);
        s_S[kidx] = val;
        sum = (double)val;
    }
    sum = wave_sum_f64(sum);
    if (lane == 0) s_redd[wave] = sum;
    __syncthreads();
    sum = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) sum += s_redd[w];
    const float inv = inv_sum_f32(sum);
    DEC_STAMP(3);
    {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < NV; k += 2) {
            const int j0 = sl + NW * k, j1 = j0 + NW;
            if (j0 < T) a0 += (double)__fmul_rn(vr[k], __fmul_rn(s_S[j0], inv));
            if (j1 < T) a1 += (double)__fmul_rn(vr[k + 1], __fmul_rn(s_S[j1], inv));
        }
        s_pv[tid] = a0 + a1;
    }
    __syncthreads();
    if (tid < DK) {
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < NW; s2 += 2) { t0 += s_pv[s2 * DK + tid]; t1 += s_pv[(s2 + 1) * DK + tid]; }
        const float o = (float)(t0 + t1);
        if (p.att_out) p.att_out[h * DK + tid] = o;
        int8_t q8; float d8; uint32_t s8;
        q8_block32(o, p.q81 != 0, q8, d8, s8);
        const int blk = h * 2 + (tid >> 5);
        p.oq_q[blk * 32 + (tid & 31)] = q8;
        if ((tid & 31) == 0) { p.oq_d[blk] = d8; p.oq_s[blk] = s8; }
    }
    DEC_STAMP(4);
    DEC_WALL(1);
}

// ---- C: out_proj + bias + residual (biogpt.cpp:767-772); Q8 activation row from the attention kernel --------------
struct DecOprojParams {
    DevMatrix Wo;              // [1024][1024]
    const int8_t *aq_q; const float *aq_d; const uint32_t *aq_s;   // 32 Q8 blocks
    const float *bias;
    const float *resid;        // x
    float *out;                // x1
    unsigned long long *tstamp;
    unsigned long long *wall;
    int32_t wall_slot;
};
__host__ __device__ inline size_t dec_oproj_smem_bytes(int waves) { return (size_t)waves * 2 * DEC_PS * 4; }

// NW waves per workgroup, 2 rows per wave (lane = block): grid = 1024 / (2 * NW)
template <int WT, int NW>
__global__ __launch_bounds__(NW * 64) void dec_oproj_kernel(const DecOprojParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *const s_part = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 31, rsub = lane >> 5;
    const int row = (blockIdx.x * NW + wave) * 2 + rsub;
    DEC_STAMP(0);
    DEC_WALL(0);
    Unit<WT> wq;
    load_unit<WT>(wq, p.Wo, (int64_t)row * 32 + sub);
    uint32_t ax[8];
    {
        const uint4 *aq = reinterpret_cast<const uint4 *>(p.aq_q) + sub * 2;
        const uint4 a = aq[0], b = aq[1];
        ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
    }
    const float axd = p.aq_d[sub];
    const uint32_t axs = p.aq_s[sub];
    const int frow = (blockIdx.x * NW + wave) * 2 + (lane & 1);
    float e_bias = 0.0f, e_res = 0.0f;
    if (lane < 2) { e_bias = p.bias[frow]; e_res = p.resid[frow]; }
    DEC_STAMP(1);
    float *const part = s_part + wave * 2 * DEC_PS;
    part[rsub * DEC_PS + sub] = unit_dot_quant<WT>(wq, ax, axd, __uint_as_float(axs), (int)axs);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    DEC_STAMP(2);
    if (lane < 2) DEC_STORE_F32(p.out + frow, __fadd_rn(__fadd_rn(sum32_in_order(part + lane * DEC_PS), e_bias), e_res));
    DEC_STAMP(3);
    DEC_WALL(1);
}

// ---- D: LayerNorm -> Q8 -> fc1 rows -> bias -> GELU table -> Q8 block(s) for fc2 (biogpt.cpp:777-787) -------------
struct DecFc1Params {
    const float *x1;           // [1024] x + out_proj(...) + b_o
    const float *ln_w, *ln_b;
    float eps;
    DevMatrix W1;              // [4096][1024]
    const float *b1;
    const uint16_t *gelu_tab;
    int8_t *oq_q; float *oq_d; uint32_t *oq_s;   // fc1 output as Q8 blocks [4096/32]
    int32_t q81;
    unsigned long long *tstamp;
    unsigned long long *wall;
    int32_t wall_slot;
};

template <int NB> __host__ __device__ inline size_t dec_fc1_smem_bytes() { return 1536 + 256 + (size_t)32 * NB * DEC_PS * 4; }

// NB = Q8 output blocks (of 32 rows) per workgroup: grid = 4096 / (32 * NB); NW waves, each 2 rows per step, 16 NB / NW steps
template <int WT, int NB, int NW>
__global__ __launch_bounds__(NW * 64) void dec_fc1_kernel(const DecFc1Params p) {
    using TI = TypeInfo<WT>;
    static_assert(NB == 1 || NB == 2, "one wave quantizes the workgroup's output blocks");
    static_assert(NW >= 4 && (16 * NB) % NW == 0, "waves per workgroup");
    constexpr int STEPS = 16 * NB / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *const s_xq = reinterpret_cast<uint32_t *>(smem);
    float *const s_xd = reinterpret_cast<float *>(smem + 1024);
    uint32_t *const s_xs = reinterpret_cast<uint32_t *>(smem + 1152);
    double *const s_red = reinterpret_cast<double *>(smem + 1280);
    float *const s_g = reinterpret_cast<float *>(smem + 1536);     // [32 * NB] GELU outputs
    float *const s_part = reinterpret_cast<float *>(smem + 1536 + 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 31, rsub = lane >> 5;
    const int row0 = blockIdx.x * 32 * NB;
    DEC_STAMP(0);
    DEC_WALL(0);

    float4 x1 = make_float4(0.f, 0.f, 0.f, 0.f), lnw = x1, lnb = x1;      // the column first: the LayerNorm chain starts from it
    if (tid < 256) {
        x1 = reinterpret_cast<const float4 *>(p.x1)[tid];
        lnw = reinterpret_cast<const float4 *>(p.ln_w)[tid];
        lnb = reinterpret_cast<const float4 *>(p.ln_b)[tid];
    }
    Unit<WT> wq[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; s++) load_unit<WT>(wq[s], p.W1, (int64_t)(row0 + s * 2 * NW + wave * 2 + rsub) * 32 + sub);
    float e_bias = 0.0f;
    if (lane < 2 * STEPS) e_bias = p.b1[row0 + (lane >> 1) * 2 * NW + wave * 2 + (lane & 1)];
    DEC_STAMP(1);

    if (TI::q81) ln4_q8_1024<true>(x1, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs, p.tstamp);
    else ln4_q8_1024<false>(x1, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs, p.tstamp);
    DEC_STAMP(2);

    {
        uint32_t ax[8];
        const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
        ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
        const float axd = s_xd[sub];
        const uint32_t axs = s_xs[sub];
        float *const part = s_part + wave * 2 * STEPS * DEC_PS;
#pragma unroll
        for (int s = 0; s < STEPS; s++)
            part[(s * 2 + rsub) * DEC_PS + sub] = unit_dot_quant<WT>(wq[s], ax, axd, __uint_as_float(axs), (int)axs);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 2 * STEPS) {
            const float v = __fadd_rn(e_bias, sum32_in_order(part + lane * DEC_PS));
            s_g[(lane >> 1) * 2 * NW + wave * 2 + (lane & 1)] = h2f(p.gelu_tab[f2h(v)]);   // ggml_gelu: fp16 table
        }
    }
    __syncthreads();
    DEC_STAMP(3);
    // the workgroup's 32 * NB outputs = NB Q8 blocks of fc2's activation row (quantize_row_q8_0 / q8_1)
    if (tid < 32 * NB) {
        int8_t q8; float d8; uint32_t s8;
        q8_block32(s_g[tid], p.q81 != 0, q8, d8, s8);
        const int blk = blockIdx.x * NB + (tid >> 5);
        p.oq_q[(size_t)blk * 32 + (tid & 31)] = q8;
        if ((tid & 31) == 0) { p.oq_d[blk] = d8; p.oq_s[blk] = s8; }
    }
    DEC_STAMP(4);
    DEC_WALL(1);
}

// ---- E: fc2 + bias + residual (biogpt.cpp:790-795) ------------------------------------------------------------------
struct DecFc2Params {
    DevMatrix W2;              // [1024][4096]
    const int8_t *aq_q; const float *aq_d; const uint32_t *aq_s;   // fc1 output as 128 Q8 blocks
    const float *bias;
    const float *resid;        // x1
    float *out;                // next layer's input x
    unsigned long long *tstamp;
    unsigned long long *wall;
    int32_t wall_slot;
    uint32_t *seq;             // last layer only: the eval's lineage words (kernels.hip.h, SEQ_*); null: off
};

constexpr int DEC_PS2 = 132;   // 128 block terms + 4
__host__ __device__ inline size_t dec_fc2_smem_bytes(int waves) { return (size_t)waves * DEC_PS2 * 4; }

// one row per wave: lane holds blocks lane and lane + 64; lane 0 adds the 128 block terms in block order
template <int WT, int NW>
__global__ __launch_bounds__(NW * 64) void dec_fc2_kernel(const DecFc2Params p) {
    constexpr int BPR = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *const s_part = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x * NW + wave;
    DEC_STAMP(0);
    DEC_WALL(0);
    seq_forward(p.seq, SEQ_LAST_LAYER);
    Unit<WT> wq[2];
    uint32_t ax[2][8];
    float axd[2];
    uint32_t axs[2];
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int u = lane + 64 * it;
        load_unit<WT>(wq[it], p.W2, (int64_t)row * BPR + u);
        const uint4 *aq = reinterpret_cast<const uint4 *>(p.aq_q) + u * 2;
        const uint4 a = aq[0], b = aq[1];
        ax[it][0] = a.x; ax[it][1] = a.y; ax[it][2] = a.z; ax[it][3] = a.w; ax[it][4] = b.x; ax[it][5] = b.y; ax[it][6] = b.z; ax[it][7] = b.w;
        axd[it] = p.aq_d[u];
        axs[it] = p.aq_s[u];
    }
    float e_bias = 0.0f, e_res = 0.0f;
    if (lane == 0) { e_bias = p.bias[row]; e_res = p.resid[row]; }
    DEC_STAMP(1);
    float *const part = s_part + wave * DEC_PS2;
#pragma unroll
    for (int it = 0; it < 2; it++)
        part[lane + 64 * it] = unit_dot_quant<WT>(wq[it], ax[it], axd[it], __uint_as_float(axs[it]), (int)axs[it]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    DEC_STAMP(2);
    if (lane == 0) {
        const float4 *p4 = reinterpret_cast<const float4 *>(part);
        float sumf = 0.0f;
#pragma unroll
        for (int b0 = 0; b0 < BPR / 4; b0 += 8) {
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; j++) t[j] = p4[b0 + j];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                sumf = __fadd_rn(sumf, t[j].x); sumf = __fadd_rn(sumf, t[j].y);
                sumf = __fadd_rn(sumf, t[j].z); sumf = __fadd_rn(sumf, t[j].w);
            }
        }
        DEC_STORE_F32(p.out + row, __fadd_rn(__fadd_rn(sumf, e_bias), e_res));      // biogpt.cpp:790-795
    }
    DEC_STAMP(3);
    DEC_WALL(1);
}

// ---- device-side top-k of the logits row (the sampler needs k <= 64 candidates, not 170 KB over PCIe) ----
// Exact selection of the k largest logits, best first (equal logits: lower id first), one 1024-thread workgroup:
//   1. the lm_head kernel left one maximum per workgroup; the k-th largest of (groups of 8 of) them is a lower bound T0 of the
//      k-th largest logit (k workgroups each hold a logit >= it), so every member of the top k is >= T0;
//   2. one sweep over the row collects the logits >= T0 (a few dozen) in LDS;
//   3. every candidate counts the candidates that beat it: its rank; ranks < k are the answer, already ordered.
// out_n[0] = k, or -1 if more than TOPK_CAP logits reach T0 (degenerate rows: the caller falls back to the full row).
// biogpt_sample_top_k_top_p (biogpt.cpp:908-980) only ever looks at these k logits.
constexpr int TOPK_CAP = 4096;
__global__ __launch_bounds__(1024) void topk_kernel(const float *logits, int V, int k, const float *pmax_val, int nparts, float *out_val,
                                                    int32_t *out_idx, int32_t *out_n) {
    __shared__ float s_gm[128];
    __shared__ float s_cv[TOPK_CAP];
    __shared__ int s_ci[TOPK_CAP];
    __shared__ float s_t0;
    __shared__ int s_n;
    const int tid = threadIdx.x;
    if (tid == 0) { s_t0 = -INFINITY; s_n = 0; }
    // 0. the row's loads first (they do not depend on the threshold): up to 42 per thread, all in flight while step 1 runs -- the
    //    row was written by other compute units, every round of loads is a trip to the memory side
    constexpr int TK_U = 42;
    float x[TK_U];
#pragma unroll
    for (int u = 0; u < TK_U; u++) x[u] = (tid + 1024 * u < V) ? logits[tid + 1024 * u] : -INFINITY;
    // 1. threshold: the partial maxima in 128 groups of 8; the k-th largest GROUP maximum is still the maximum of some
    //    workgroup, so at least k logits reach it -- a lower bound of the k-th largest logit, found with 128 x 128 compares
    //    (8 threads per group, 16 compares each)
    const int np = min(nparts, 1024);
    float gm = (tid < np) ? pmax_val[tid] : -INFINITY;
    gm = group8_max(gm);
    if ((tid & 7) == 0) s_gm[tid >> 3] = gm;
    __syncthreads();
    const int ngroups = (np + 7) >> 3;
    if (k <= ngroups && nparts <= 1024) {
        const int gi = tid >> 3, j0 = (tid & 7) * 16;
        const float mine = s_gm[gi];
        int beat = 0;
#pragma unroll
        for (int j = j0; j < j0 + 16; j++) {
            const float o = s_gm[j];
            beat += (o > mine || (o == mine && j < gi)) ? 1 : 0;
        }
        beat = group8_sum(beat);
        if ((tid & 7) == 0 && beat == k - 1) s_t0 = mine;
    }
    __syncthreads();
    const float t0 = s_t0;
    // 2. the candidates: every logit that reaches the threshold
#pragma unroll
    for (int u = 0; u < TK_U; u++) {
        if (tid + 1024 * u < V && x[u] >= t0) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < TOPK_CAP) { s_cv[slot] = x[u]; s_ci[slot] = tid + 1024 * u; }
        }
    }
    for (int i0 = tid + TK_U * 1024; i0 < V; i0 += 1024) {      // vocabularies beyond 43008 entries
        const float xv = logits[i0];
        if (xv >= t0) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < TOPK_CAP) { s_cv[slot] = xv; s_ci[slot] = i0; }
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n > TOPK_CAP || n < k) { if (tid == 0) out_n[0] = -1; return; }   // n < k only if the row holds NaNs
    // 3. rank of every candidate among the candidates (equal logits: lower id first); ranks < k are the answer, in order
    for (int c = tid; c < n; c += 1024) {
        const float x = s_cv[c];
        const int ix = s_ci[c];
        int beat = 0;
        for (int j = 0; j < n; j++) {
            const float o = s_cv[j];
            beat += (o > x || (o == x && s_ci[j] < ix)) ? 1 : 0;
        }
        if (beat < k) { out_val[beat] = x; out_idx[beat] = ix; }
    }
    if (tid == 0) out_n[0] = k;
}

// biogpt_hip_eval's logits row: written by the device straight into pinned host memory as the last node of the replayed
// graph (a device-to-host copy command behind the graph costs 60-120 us of extra latency per token on this runtime)
__global__ __launch_bounds__(256) void logits_to_host_kernel(const float *src, float *dst_pinned, int n, const uint32_t *seq, int stamp_at) {
    if (seq != nullptr && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<uint32_t *>(dst_pinned)[stamp_at] = seq[SEQ_LM_HEAD];   // the row's lineage (kernels.hip.h)
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) *reinterpret_cast<float4 *>(dst_pinned + i) = *reinterpret_cast<const float4 *>(src + i);
    else for (int j = i; j < n; j++) dst_pinned[j] = src[j];
}

// Single-token evals through the C API: the host drops {n_past, causal, token} into a ring of pinned slots and replays the
// captured step; this first node of the graph pulls the next slot into the device state (one PCIe read instead of a copy
// command + its blit kernel in front of every token).  ctr counts the replays; the host mirrors it.
__global__ void fetch_state_kernel(const int32_t *mbox, uint32_t *ctr, DevState *st, uint32_t *seq) {
    const uint32_t n = *ctr;
    const int32_t *slot = mbox + (size_t)(n & 63u) * 8;
    if (seq) seq[SEQ_FETCHED] = (uint32_t)slot[3];       // the call's sequence number: forwarded by the last layer and the lm_head, returned beside the row
    st->n_past = slot[0];
    st->n_gen = 0;
    st->causal = slot[1];
    st->chunk = 0;
    state_tokens(st)[0] = slot[2];
    *ctr = n + 1u;
}

// the device-side position moves on without a sampler launch: after a prompt pass, before the first fused step
__global__ void advance_state_kernel(DevState *st, int n_eval) { st->n_past = st->n_past + n_eval; }

}  // namespace bgk
