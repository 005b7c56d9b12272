// Single-token decode at BioGPT-base shapes (d_model 1024, d_ff 4096, 16 heads of 64, block-quantized weights,
// contexts up to 256 keys): the five dependent launches per layer of the first chain (LN+QKV, attention, out_proj,
// LN+fc1, fc2) become THREE, and the embedding / arg-max launches disappear.
//
// Why this shape (measured on the MI355X, tools/microbench4.hip, profiles/microbench4_r2.txt):
//   * a dependent kernel boundary costs 1.58 us whatever the grid (16..512 workgroups, 256 or 1024 threads, LDS,
//     kernarg size); an in-launch last-arriver ticket (sc1 stores -> drain -> agent atomic -> sc1 loads) costs
//     1.9-2.5 us -- MORE than the boundary it would replace.  So no cross-workgroup hand-offs: every fusion below is
//     workgroup-local, and a token is 3 x 24 + 1 = 73 launches instead of 122;
//   * one 1024-thread workgroup pulls 128 KB issued at once in ~1.0 us beyond the boundary (256 KB: 2.2 us), so a
//     head's whole working set (q/k/v rows 110 KB, its K/V rows, its out_proj columns 36 KB) can go through ONE
//     compute unit;
//   * 16 waves per workgroup share the LayerNorm statistics (each wave 1/16 of the column + one LDS exchange)
//     instead of every wave re-reducing the whole column in double.
//
//   dec_attn_kernel   one workgroup per HEAD: [embedding (+ arg-max of the previous token's logits partials)] ->
//                     LayerNorm -> Q8 -> the head's 192 q/k/v rows -> KV append -> attention over the cache ->
//                     Q8 of the head's 64 outputs -> the head's two out_proj block terms for all 1024 rows
//                     biogpt.cpp:664-686, :691-764, :767 (the mat-mul part)
//   dec_fc1_kernel    x1 = x + b_o + sum of the 32 out_proj block terms IN BLOCK ORDER (the reference's scalar
//                     association) -> LayerNorm -> Q8 -> fc1 rows -> bias -> GELU table -> Q8 block(s) for fc2
//                     biogpt.cpp:767-787
//   dec_fc2_kernel    fc2 + bias + residual, one row per wave, 16 rows per workgroup        biogpt.cpp:790-795
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

#ifdef BIOGPT_HIP_PROFILE_HOOKS   // make EXTRA=-DBIOGPT_HIP_PROFILE_HOOKS: per-segment shader-clock stamps of workgroup 0, wave 0
#define DEC_STAMP(k) do { if (p.tstamp && blockIdx.x == 0 && threadIdx.x == 0) p.tstamp[(k)] = __builtin_readcyclecounter(); } while (0)
// dbg & 64: entry / exit of EVERY workgroup on the constant 100 MHz clock, [slot][1024 workgroups][2] after the segment stamps
#define DEC_WALL(which) do { if (p.wall && threadIdx.x == 0) \
        p.wall[((size_t)p.wall_slot * 1024 + blockIdx.x) * 2 + (which)] = wall_clock64(); } while (0)
#else
#define DEC_STAMP(k) do {} while (0)
#define DEC_WALL(which) do {} while (0)
#endif

// LayerNorm (ggml_norm + affine, double statistics) and Q8_0 / Q8_1 quantization of ONE 1024-element column by a
// 1024-thread workgroup, thread t holding element t.  Leaves the 32 activation blocks in LDS (s_xq / s_xd / s_xs).
// s_red: 32 doubles.  Ends with a workgroup barrier.
template <bool Q81>
__device__ __forceinline__ void coop_ln_q8_1024(float xv, float lnw, float lnb, float eps, double *s_red, uint32_t *s_xq, float *s_xd,
                                                uint32_t *s_xs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double inv_k = 1.0 / 1024.0;
    const double s1 = wave_sum_f64((double)xv);
    if (lane == 0) s_red[wave] = s1;
    __syncthreads();
    double t1 = 0.0;
#pragma unroll
    for (int w = 0; w < 16; w++) t1 += s_red[w];
    const float mean = (float)(t1 * inv_k);
    const float dv = __fsub_rn(xv, mean);
    const double s2 = wave_sum_f64((double)__fmul_rn(dv, dv));
    if (lane == 0) s_red[16 + wave] = s2;
    __syncthreads();
    double t2 = 0.0;
#pragma unroll
    for (int w = 0; w < 16; w++) t2 += s_red[16 + w];
    const float var = (float)(t2 * inv_k);
    const float scale = 1.0f / sqrtf(__fadd_rn(var, eps));
    const float y = __fadd_rn(__fmul_rn(lnw, __fmul_rn(dv, scale)), lnb);
    // quantize_row_q8_0 / q8_1: one block = 32 consecutive lanes
    const float amax = group32_max(fabsf(y));
    const float d = amax / 127.0f;
    const float id = (d != 0.0f) ? 1.0f / d : 0.0f;
    const int q = (int)roundf(__fmul_rn(y, id));
    const int isum = group32_sum(q);
    reinterpret_cast<int8_t *>(s_xq)[tid] = (int8_t)q;
    if ((lane & 31) == 0) {
        const int b = tid >> 5;
        if (Q81) { s_xd[b] = d; s_xs[b] = __float_as_uint(__fmul_rn((float)isum, d)); }
        else { s_xd[b] = h2f(f2h(d)); s_xs[b] = (uint32_t)isum; }
    }
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
__device__ __forceinline__ void q8_block32(float v, bool q81, int8_t &q_out, float &d_out, uint32_t &s_out) {
    const float amax = group32_max(fabsf(v));
    const float d = amax / 127.0f;
    const float id = (d != 0.0f) ? 1.0f / d : 0.0f;
    const int q = (int)roundf(__fmul_rn(v, id));
    const int isum = group32_sum(q);
    q_out = (int8_t)q;
    if (q81) { d_out = d; s_out = __float_as_uint(__fmul_rn((float)isum, d)); }
    else { d_out = h2f(f2h(d)); s_out = (uint32_t)isum; }
}

struct DecAttnParams {
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
    float *kcache, *vcache;    // layer slice, head-major [H][P][64]
    int32_t P, t_cap;          // t_cap: launch-time bound of the context (multiple of 64, <= 256) for the cache loads
    const uint16_t *exp_tab;
    DevMatrix Wo;              // [1024][1024]
    float *terms;              // out: out_proj block terms, block-major [32][1024]
    float *att_out;            // optional: F32 attention output [1024] (null: off)
    int32_t q81;
    unsigned long long *tstamp;
    unsigned long long *wall;
    int32_t wall_slot;
};

constexpr int DEC_PS = 36;     // floats between two rows of block terms in LDS (32 + 4: float4 aligned, skewed)
__host__ __device__ inline size_t dec_attn_smem_bytes() { return 3728 + (size_t)16 * 12 * DEC_PS * 4; }

template <int WT>
__global__ __launch_bounds__(1024) void dec_attn_kernel(const DecAttnParams p) {
    using TI = TypeInfo<WT>;
    constexpr int D = 1024, DK = 64, NST = 6, RPWV = 12;   // 192 rows / 16 waves, 2 rows per wave step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *const s_xq = reinterpret_cast<uint32_t *>(smem);
    float *const s_xd = reinterpret_cast<float *>(smem + 1024);
    uint32_t *const s_xs = reinterpret_cast<uint32_t *>(smem + 1152);
    double *const s_red = reinterpret_cast<double *>(smem + 1280);
    float *const s_q = reinterpret_cast<float *>(smem + 1536);
    float *const s_k = s_q + 64, *const s_v = s_q + 128;
    float *const s_S = reinterpret_cast<float *>(smem + 2304);
    float *const s_redf = reinterpret_cast<float *>(smem + 3328);
    double *const s_redd = reinterpret_cast<double *>(smem + 3392);
    uint32_t *const s_aq = reinterpret_cast<uint32_t *>(smem + 3520);
    float *const s_ad = reinterpret_cast<float *>(smem + 3584);
    uint32_t *const s_as = reinterpret_cast<uint32_t *>(smem + 3592);
    float *const s_amv = reinterpret_cast<float *>(smem + 3600);
    int *const s_ami = reinterpret_cast<int *>(smem + 3664);
    float *const s_part = reinterpret_cast<float *>(smem + 3728);
    double *const s_pv = reinterpret_cast<double *>(smem + 3728);   // reuses the block-term strips after the q/k/v finish

    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 31, rsub = lane >> 5;
    DEC_STAMP(0);
    DEC_WALL(0);

    // ---- t = 0: every load that does not depend on the token -------------------------------------------------
    Unit<WT> wq[NST];
#pragma unroll
    for (int s = 0; s < NST; s++) {
        const int R = wave * RPWV + s * 2 + rsub;                        // row of the head's [q; k; v] stack
        const int grow = (R >> 6) * D + h * DK + (R & 63);
        load_unit<WT>(wq[s], p.Wqkv, (int64_t)grow * 32 + sub);
    }
    float e_bias = 0.0f;
    if (lane < RPWV) {
        const int R = wave * RPWV + lane;
        e_bias = p.bqkv[(R >> 6) * D + h * DK + (R & 63)];
    }
    // this thread's two out_proj units (row tid, blocks 2h / 2h+1); Q8_0 units are twice as large and are fetched once
    // the q/k/v units have been consumed (register budget of a 1024-thread workgroup: 128)
    constexpr bool WO_EARLY = (WT != W_Q8_0);
    Unit<WT> wo[2];
    if (WO_EARLY) {
#pragma unroll
        for (int b = 0; b < 2; b++) load_unit<WT>(wo[b], p.Wo, (int64_t)tid * 32 + 2 * h + b);
    }
    const int t_cap = p.t_cap;
    const int ksub = tid & 3, kidx = tid >> 2;
    const int dd = tid & (DK - 1), sl = tid >> 6;
    float4 kr[4];
    {
        const float4 *kbase = reinterpret_cast<const float4 *>(p.kcache + (size_t)h * p.P * DK) + ksub;
        if (kidx < t_cap) {
#pragma unroll
            for (int m = 0; m < 4; m++) kr[m] = kbase[(size_t)kidx * (DK / 4) + 4 * m];
        }
    }
    float vr[16];
    {
        const float *vbase = p.vcache + (size_t)h * p.P * DK + dd;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int j = sl + 16 * k;
            if (j < t_cap) vr[k] = vbase[(size_t)j * DK];
        }
    }
    const float lnw = p.ln_w[tid], lnb = p.ln_b[tid];
    const int n_past = p.st->n_past;
    float xv;
    if (p.tok_src == 0) {
        xv = p.x[tid];
    } else {
        int tok;
        if (p.tok_src == 2) {
            // greedy sampler of the PREVIOUS token (main.cpp:109-128, top_k = 1): finish the arg-max over the lm_head
            // kernel's per-workgroup partials (lowest id wins ties); workgroup 0 records it
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int k = tid; k < p.nparts; k += 1024) {
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
            for (int w = 1; w < 16; w++)
                if (s_amv[w] > bv || (s_amv[w] == bv && s_ami[w] < bi)) { bv = s_amv[w]; bi = s_ami[w]; }
            tok = bi;
            if (tok < 0 || tok >= p.n_vocab) tok = 0;      // partials never written (first replay of a fresh context)
            if (h == 0 && tid == 0) {
                int32_t *tokens = state_tokens(p.st);
                const int g = p.st->n_gen;
                if (g < p.n_positions) tokens[p.n_positions + g] = tok;
                tokens[0] = tok;
            }
        } else {
            tok = state_tokens(p.st)[0];
        }
        // biogpt.cpp:664-686: embed_tokens[tok] * sqrt(D) + embed_positions[n_past + 2]
        const float te = __fmul_rn(dequant_elem(p.tok_emb, tok, tid), p.embed_scale);
        const float pe = dequant_elem(p.pos_emb, n_past + 2, tid);
        xv = __fadd_rn(te, pe);
        if (h == 0) p.x_out[tid] = xv;
    }
    DEC_STAMP(1);

    // ---- LayerNorm + Q8 of the column, shared by the 16 waves ---------------------------------------------------
    if (TI::q81) coop_ln_q8_1024<true>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
    else coop_ln_q8_1024<false>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
    DEC_STAMP(2);

    // ---- the head's 192 q/k/v rows: lane = block, 2 rows per wave step ----------------------------------------
    {
        uint32_t ax[8];
        const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
        ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
        const float axd = s_xd[sub];
        const uint32_t axs = s_xs[sub];
        float *const part = s_part + wave * RPWV * DEC_PS;
#pragma unroll
        for (int s = 0; s < NST; s++)
            part[(s * 2 + rsub) * DEC_PS + sub] = unit_dot_quant<WT>(wq[s], ax, axd, __uint_as_float(axs), (int)axs);
        if (!WO_EARLY) {
#pragma unroll
            for (int b = 0; b < 2; b++) load_unit<WT>(wo[b], p.Wo, (int64_t)tid * 32 + 2 * h + b);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < RPWV) {
            const float v = __fadd_rn(e_bias, sum32_in_order(part + lane * DEC_PS));
            const int R = wave * RPWV + lane, which = R >> 6, rr = R & 63;
            if (which == 0) {
                s_q[rr] = __fmul_rn(v, p.q_scale);                     // Q scaled AFTER the bias (biogpt.cpp:708-710)
            } else {
                (which == 1 ? s_k : s_v)[rr] = v;
                (which == 1 ? p.kcache : p.vcache)[((size_t)h * p.P + n_past) * DK + rr] = v;   // KV append (biogpt.cpp:721-727)
            }
        }
    }
    __syncthreads();
    DEC_STAMP(3);

    // ---- attention of this head over T = n_past + 1 keys (attn_fast_kernel<1, true> at 1024 threads) ----------
    const int T = n_past + 1;
    float sc;
    {
        if (kidx == n_past) {                                           // this token's key row is still only in LDS
#pragma unroll
            for (int m = 0; m < 4; m++) kr[m] = *reinterpret_cast<const float4 *>(s_k + 16 * m + 4 * ksub);
        }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const float4 qm = *reinterpret_cast<const float4 *>(s_q + 16 * m + 4 * ksub);
            a0 += (double)__fmul_rn(kr[m].x, qm.x); a1 += (double)__fmul_rn(kr[m].y, qm.y);
            a2 += (double)__fmul_rn(kr[m].z, qm.z); a3 += (double)__fmul_rn(kr[m].w, qm.w);
        }
        double acc = (a0 + a1) + (a2 + a3);
        acc += dpp_d<DPP_QUAD_XOR1>(acc);
        acc += dpp_d<DPP_QUAD_XOR2>(acc);
        sc = (kidx < T) ? (float)acc : -INFINITY;
    }
    float mx = wave_max_f32(sc);
    if (lane == 0) s_redf[wave] = mx;
    __syncthreads();
    mx = s_redf[0];
#pragma unroll
    for (int w = 1; w < 16; w++) mx = fmaxf(mx, s_redf[w]);
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
    for (int w = 0; w < 16; w++) sum += s_redd[w];
    const float inv = inv_sum_f32(sum);
    {
        double a0 = 0.0, a1 = 0.0;
        const float vnew = s_v[dd];
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            const int j0 = sl + 16 * k, j1 = j0 + 16;
            if (j0 < T) a0 += (double)__fmul_rn(j0 == n_past ? vnew : vr[k], __fmul_rn(s_S[j0], inv));
            if (j1 < T) a1 += (double)__fmul_rn(j1 == n_past ? vnew : vr[k + 1], __fmul_rn(s_S[j1], inv));
        }
        s_pv[tid] = a0 + a1;
    }
    __syncthreads();
    if (tid < DK) {
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < 16; s2 += 2) { t0 += s_pv[s2 * DK + tid]; t1 += s_pv[(s2 + 1) * DK + tid]; }
        const float o = (float)(t0 + t1);
        if (p.att_out) p.att_out[h * DK + tid] = o;
        int8_t q8; float d8; uint32_t s8;
        q8_block32(o, p.q81 != 0, q8, d8, s8);
        reinterpret_cast<int8_t *>(s_aq)[tid] = q8;
        if ((tid & 31) == 0) { s_ad[tid >> 5] = d8; s_as[tid >> 5] = s8; }
    }
    __syncthreads();
    DEC_STAMP(4);

    // ---- the head's share of out_proj: block terms 2h, 2h+1 of every row (summed in block order by dec_fc1_kernel) ----
#pragma unroll
    for (int b = 0; b < 2; b++) {
        uint32_t ax[8];
        const uint4 a = *reinterpret_cast<const uint4 *>(s_aq + b * 8), c = *reinterpret_cast<const uint4 *>(s_aq + b * 8 + 4);
        ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = c.x; ax[5] = c.y; ax[6] = c.z; ax[7] = c.w;
        const uint32_t xs = s_as[b];
        p.terms[(size_t)(2 * h + b) * D + tid] = unit_dot_quant<WT>(wo[b], ax, s_ad[b], __uint_as_float(xs), (int)xs);
    }
    DEC_STAMP(5);
    DEC_WALL(1);
}

struct DecFc1Params {
    const float *terms;        // [32][1024] out_proj block terms of this layer (dec_attn_kernel)
    const float *x;            // [1024] the layer's input (residual of out_proj)
    const float *bo;           // out_proj bias
    float *x1_out;             // [1024] x1 = x + out_proj(...) + b_o, written by workgroup 0 (fc2's residual)
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

template <int NB> __host__ __device__ inline size_t dec_fc1_smem_bytes() { return 1536 + 256 + (size_t)16 * 2 * NB * DEC_PS * 4; }

// NB = Q8 output blocks (of 32 rows) per workgroup: grid = 4096 / (32 * NB)
template <int WT, int NB>
__global__ __launch_bounds__(1024) void dec_fc1_kernel(const DecFc1Params p) {
    using TI = TypeInfo<WT>;
    static_assert(NB == 1 || NB == 2, "one wave quantizes the workgroup's output blocks");
    constexpr int D = 1024;
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

    Unit<WT> wq[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) load_unit<WT>(wq[s], p.W1, (int64_t)(row0 + s * 32 + wave * 2 + rsub) * 32 + sub);
    float e_bias = 0.0f;
    if (lane < 2 * NB) e_bias = p.b1[row0 + (lane >> 1) * 32 + wave * 2 + (lane & 1)];
    float t[32];
#pragma unroll
    for (int b = 0; b < 32; b++) t[b] = p.terms[(size_t)b * D + tid];
    const float xres = p.x[tid], bo = p.bo[tid];
    const float lnw = p.ln_w[tid], lnb = p.ln_b[tid];
    // out_proj: the row's 32 block terms in block order, then bias, then residual (biogpt.cpp:767-772)
    float sumf = 0.0f;
#pragma unroll
    for (int b = 0; b < 32; b++) sumf = __fadd_rn(sumf, t[b]);
    const float x1 = __fadd_rn(__fadd_rn(sumf, bo), xres);
    if (blockIdx.x == 0) p.x1_out[tid] = x1;
    DEC_STAMP(1);

    if (TI::q81) coop_ln_q8_1024<true>(x1, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
    else coop_ln_q8_1024<false>(x1, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
    DEC_STAMP(2);

    {
        uint32_t ax[8];
        const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
        ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
        const float axd = s_xd[sub];
        const uint32_t axs = s_xs[sub];
        float *const part = s_part + wave * 2 * NB * DEC_PS;
#pragma unroll
        for (int s = 0; s < NB; s++)
            part[(s * 2 + rsub) * DEC_PS + sub] = unit_dot_quant<WT>(wq[s], ax, axd, __uint_as_float(axs), (int)axs);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 2 * NB) {
            const float v = __fadd_rn(e_bias, sum32_in_order(part + lane * DEC_PS));
            s_g[(lane >> 1) * 32 + wave * 2 + (lane & 1)] = h2f(p.gelu_tab[f2h(v)]);   // ggml_gelu: fp16 table
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

struct DecFc2Params {
    DevMatrix W2;              // [1024][4096]
    const int8_t *aq_q; const float *aq_d; const uint32_t *aq_s;   // fc1 output as 128 Q8 blocks
    const float *bias;
    const float *resid;        // x1
    float *out;                // next layer's input x
    unsigned long long *tstamp;
    unsigned long long *wall;
    int32_t wall_slot;
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
        p.out[row] = __fadd_rn(__fadd_rn(sumf, e_bias), e_res);      // biogpt.cpp:790-795
    }
    DEC_STAMP(3);
    DEC_WALL(1);
}

// the device-side position moves on without a sampler launch: after a prompt pass, before the first fused step
__global__ void advance_state_kernel(DevState *st, int n_eval) { st->n_past = st->n_past + n_eval; }

}  // namespace bgk
