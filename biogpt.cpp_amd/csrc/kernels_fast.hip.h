// Shape-specialised single-token (N = 1) mat-vec kernel for the block-quantized weight types.
//
// Same arithmetic, same LDS hand-offs and same results as the generic matvec_kernel in kernels.hip.h
// (which stays the path for prefill chunks, float weight types and unusual shapes), but the row length
// K is a compile-time constant so that every loop is fully unrolled and un-predicated.  Why this exists:
// a BioGPT-base decode mat-vec is 0.6-2.4 MB, i.e. ONE wave per SIMD with nothing to overlap -- the
// kernel's duration is its dependent-instruction count times the ~4-8 cycle issue latency, not bytes
// (measured: the generic kernel executes ~1.2k instructions per wave = ~4 us; rocprof in profiles/).
//
// Per wave (4 waves per workgroup):
//   t=0   issue loads: this lane's weight units of the first row step, the activation column
//         (LayerNorm: all K/256 chunks per lane; plain: only the wave's share), LN gain/bias of the
//         share, bias/residual of the row this lane will finish
//   LN    mean / variance in double over the whole column, per wave, DPP reductions (no barrier)
//   Q8    quantize the wave's share (Q8_0 / Q8_1 exactly as quantize_row_q8_*), publish in LDS
//   ---   one workgroup barrier
//   dot   lane = block: v_dot4 int8 dot * d_w * d_x  -> block term c_b into the wave's LDS strip
//   sum   lane f < rows: sum_b c_b in block order (the reference's scalar association), epilogue
#pragma once

#include "kernels.hip.h"

namespace bgk {

// NC = activation columns per workgroup: 1 for decode; 8 for prefill chunks (PRO_Q8IN only: the columns
// were quantized once by lnq_kernel / the producer's epilogue, every workgroup just fetches them).
template <int WT, int PRO, int EPI, int K, int PF, int NC = 1>
__global__ __launch_bounds__(256) void matvec_fast_kernel(const MatvecParams p) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant, "fast path is for the block-quantized types");
    static_assert(NC == 1 || PRO == PRO_Q8IN, "multi-column fast path takes pre-quantized activations");
    if (EPI == EPI_LOGITS) seq_forward(p.lineage, SEQ_LM_HEAD);
    constexpr int BPR = K / QK;                      // blocks per row
    constexpr int LPR = BPR < 64 ? BPR : 64;         // lanes per row
    constexpr int NIT = BPR / LPR;                   // blocks per lane per row
    constexpr int RPS = 64 / LPR;                    // rows per wave step
    constexpr int NCHUNK = K / 4;                    // float4 chunks in the column
    constexpr int NJJ = NCHUNK / 64;                 // chunks per lane over the whole column
    constexpr int NSHARE = NJJ / 4;                  // chunks per lane of one wave's share (4 waves)
    constexpr int PSTRIDE = BPR + 4;                 // floats; keeps float4 alignment, skews banks
    constexpr int TAIL = 64 + (EPI == EPI_GELU_Q8 ? 32 * NC : 0);   // floats
    static_assert(BPR % LPR == 0 && NCHUNK % 256 == 0, "K must be a multiple of 1024");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *const s_xq = reinterpret_cast<uint32_t *>(smem_raw);            // [K/4] packed int8
    float *const s_xd = reinterpret_cast<float *>(smem_raw + K);              // [BPR]
    uint32_t *const s_xs = reinterpret_cast<uint32_t *>(s_xd + BPR);          // [BPR]
    float *const s_tail = reinterpret_cast<float *>(s_xs + BPR);              // [TAIL]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rpw = p.rpw;                                                    // rows per wave (multiple of RPS)
    float *const s_part = s_tail + TAIL + wave * rpw * NC * PSTRIDE;

    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    const int M = p.W.M;
    const int row_base = (blockIdx.x * 4 + wave) * rpw;
    const int nsteps = (PF == 1) ? 1 : rpw / RPS;
    const int col0 = blockIdx.y * NC;
    const int ncols = (NC == 1) ? 1 : min(NC, p.N - col0);

    // ---- t = 0: independent loads ---------------------------------------------------------------
    const float4 *xcol = reinterpret_cast<const float4 *>(p.x);
    float4 xr[PRO == PRO_LN ? NJJ : 1];
    float4 xs4[NSHARE], lw4[NSHARE], lb4[NSHARE];
    uint32_t ax[NC][NIT][8];   // this lane's activation blocks (constant over the row steps)
    float axd[NC][NIT];
    uint32_t axs[NC][NIT];
    if (PRO == PRO_Q8IN) {
        // the producer (lnq_kernel / attention / fc1) already quantized the activations: fetch this lane's blocks
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (c < ncols) {
                const uint4 *aq = reinterpret_cast<const uint4 *>(p.aq_q + (size_t)(col0 + c) * K);
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int u = sub + it * LPR;
                    const uint4 a = aq[u * 2], b = aq[u * 2 + 1];
                    ax[c][it][0] = a.x; ax[c][it][1] = a.y; ax[c][it][2] = a.z; ax[c][it][3] = a.w;
                    ax[c][it][4] = b.x; ax[c][it][5] = b.y; ax[c][it][6] = b.z; ax[c][it][7] = b.w;
                    axd[c][it] = p.aq_d[(size_t)(col0 + c) * BPR + u];
                    axs[c][it] = p.aq_s[(size_t)(col0 + c) * BPR + u];
                }
            }
        }
    } else {
        if (PRO == PRO_LN) {
#pragma unroll
            for (int i = 0; i < NJJ; i++) xr[i] = xcol[i * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < NSHARE; i++) {
            const int ch = (wave + 4 * i) * 64 + lane;
            xs4[i] = xcol[ch];
            if (PRO == PRO_LN) {
                lw4[i] = reinterpret_cast<const float4 *>(p.ln_w)[ch];
                lb4[i] = reinterpret_cast<const float4 *>(p.ln_b)[ch];
            }
        }
    }
    Unit<WT> wq[PF][NIT];
#pragma unroll
    for (int s = 0; s < PF; s++) {
        const int row = row_base + s * RPS + rsub;
        if (s < nsteps && row < M) {
#pragma unroll
            for (int it = 0; it < NIT; it++) load_unit<WT>(wq[s][it], p.W, (int64_t)row * BPR + sub + it * LPR);
        }
    }
    // finisher slot of this lane: output (row row_base + f % rpw, column col0 + f / rpw), f = lane
    const int f_r = (NC == 1) ? lane : lane % rpw, f_c = (NC == 1) ? 0 : lane / rpw;
    const int f_row = row_base + f_r;
    const bool finisher = lane < rpw * ncols && f_row < M;
    float e_bias = 0.0f, e_res = 0.0f;
    int e_npast = 0, e_seq = 0;
    if (finisher) {
        if (EPI != EPI_LOGITS) e_bias = p.bias[f_row];
        if (EPI == EPI_RESID) e_res = p.resid[(size_t)(col0 + f_c) * p.ldr + f_row];
        if (EPI == EPI_QKV) {   // cache row (and, for columns of several sequences, cache) of this column
            e_npast = p.seq ? p.seq[col0 + f_c].n_past : p.st->n_past + col0 + f_c;
            e_seq = (p.seq && p.col_mode) ? p.seq[col0 + f_c].seq_id : col0 + f_c;
        }
    }

    // ---- LayerNorm statistics (ggml_norm: double sums; per wave over the whole column) ----------
    float mean = 0.0f, scale = 1.0f;
    if (PRO == PRO_LN) {
        double s1 = 0.0;
#pragma unroll
        for (int i = 0; i < NJJ; i++) s1 += ((double)xr[i].x + (double)xr[i].y) + ((double)xr[i].z + (double)xr[i].w);
        s1 = wave_sum_f64(s1);
        mean = (float)(s1 * p.inv_k);  // K is a power of two: exact
        double s2 = 0.0;
#pragma unroll
        for (int i = 0; i < NJJ; i++) {
            const float a = __fsub_rn(xr[i].x, mean), b = __fsub_rn(xr[i].y, mean);
            const float c = __fsub_rn(xr[i].z, mean), d = __fsub_rn(xr[i].w, mean);
            s2 += ((double)__fmul_rn(a, a) + (double)__fmul_rn(b, b)) + ((double)__fmul_rn(c, c) + (double)__fmul_rn(d, d));
        }
        s2 = wave_sum_f64(s2);
        const float var = (float)(s2 * p.inv_k);
        scale = 1.0f / sqrtf(__fadd_rn(var, p.eps));
    }

    // ---- convert the wave's share: [normalise] -> Q8_0 / Q8_1 -> LDS ----------------------------
#pragma unroll
    for (int i = 0; i < (PRO == PRO_Q8IN ? 0 : NSHARE); i++) {
        const int ch = (wave + 4 * i) * 64 + lane;
        float4 v = xs4[i];
        if (PRO == PRO_LN) {
            v.x = __fadd_rn(__fmul_rn(lw4[i].x, __fmul_rn(__fsub_rn(v.x, mean), scale)), lb4[i].x);
            v.y = __fadd_rn(__fmul_rn(lw4[i].y, __fmul_rn(__fsub_rn(v.y, mean), scale)), lb4[i].y);
            v.z = __fadd_rn(__fmul_rn(lw4[i].z, __fmul_rn(__fsub_rn(v.z, mean), scale)), lb4[i].z);
            v.w = __fadd_rn(__fmul_rn(lw4[i].w, __fmul_rn(__fsub_rn(v.w, mean), scale)), lb4[i].w);
        }
        const float amax = group8_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        const float d = amax / 127.0f;
        const float id = (d != 0.0f) ? 1.0f / d : 0.0f;
        const int q0 = (int)roundf(__fmul_rn(v.x, id)), q1 = (int)roundf(__fmul_rn(v.y, id));
        const int q2 = (int)roundf(__fmul_rn(v.z, id)), q3 = (int)roundf(__fmul_rn(v.w, id));
        const int isum = group8_sum(q0 + q1 + q2 + q3);
        s_xq[ch] = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
        if ((lane & 7) == 0) {
            const int b = ch >> 3;
            if (TI::q81) {
                s_xd[b] = d;
                s_xs[b] = __float_as_uint(__fmul_rn((float)isum, d));
            } else {
                s_xd[b] = h2f(f2h(d));
                s_xs[b] = (uint32_t)isum;
            }
        }
    }
    if (PRO != PRO_Q8IN) __syncthreads();

    // ---- this lane's activation blocks from LDS ---------------------------------------------------
#pragma unroll
    for (int it = 0; it < (PRO == PRO_Q8IN ? 0 : NIT); it++) {
        const int u = sub + it * LPR;
        const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + u * 8);
        const uint4 b = *reinterpret_cast<const uint4 *>(s_xq + u * 8 + 4);
        ax[0][it][0] = a.x; ax[0][it][1] = a.y; ax[0][it][2] = a.z; ax[0][it][3] = a.w;
        ax[0][it][4] = b.x; ax[0][it][5] = b.y; ax[0][it][6] = b.z; ax[0][it][7] = b.w;
        axd[0][it] = s_xd[u];
        axs[0][it] = s_xs[u];
    }

    // ---- row steps: block terms -> the wave's LDS strip -----------------------------------------
    for (int s0 = 0; s0 < (PF == 1 ? 1 : nsteps); s0 += PF) {   // PF == 1: launched with exactly one row step per wave
#pragma unroll
        for (int s = 0; s < PF; s++) {
            const int stp = s0 + s;
            const int row = row_base + stp * RPS + rsub;
            if (stp < nsteps && row < M) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if (c < ncols) {
#pragma unroll
                        for (int it = 0; it < NIT; it++)
                            s_part[((stp * RPS + rsub) * NC + c) * PSTRIDE + sub + it * LPR] =
                                unit_dot_quant<WT>(wq[s][it], ax[c][it], axd[c][it], __uint_as_float(axs[c][it]), (int)axs[c][it]);
                    }
                }
            }
            const int nstp = stp + PF;  // refill this register slot with the row PF steps ahead
            const int nrow = row_base + nstp * RPS + rsub;
            if (PF > 1 && nstp < nsteps && nrow < M) {
#pragma unroll
                for (int it = 0; it < NIT; it++) load_unit<WT>(wq[s][it], p.W, (int64_t)nrow * BPR + sub + it * LPR);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- finish: lane f adds its (row, column)'s block terms in block order, then the epilogue ----
    float best_val = -INFINITY;
    int best_idx = 0x7fffffff;
    if (finisher) {
        const float4 *part = reinterpret_cast<const float4 *>(s_part + (f_r * NC + f_c) * PSTRIDE);
        float sumf = 0.0f;
#pragma unroll
        for (int b0 = 0; b0 < BPR / 4; b0 += 8) {  // 32 terms per batch: 8 LDS reads in flight
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; j++) t[j] = part[b0 + j];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                sumf = __fadd_rn(sumf, t[j].x); sumf = __fadd_rn(sumf, t[j].y);
                sumf = __fadd_rn(sumf, t[j].z); sumf = __fadd_rn(sumf, t[j].w);
            }
        }
        float v = sumf;
        const int r = f_row, col = col0 + f_c;
        if (EPI == EPI_QKV) {
            v = __fadd_rn(e_bias, v);
            const int which = r / K, rr = r - which * K;  // d_model == K for the q/k/v projection
            if (which == 0) {
                p.q_out[(size_t)col * K + rr] = __fmul_rn(v, p.q_scale);
            } else {
                float *cache = ((which == 1) ? p.kcache : p.vcache) + (p.seq ? (size_t)e_seq * p.kv_seq_stride : 0);
                const int hh = rr >> p.dk_log2, dd = rr & (p.dk - 1);  // head-major cache: [H][P][dk], dk = 2^k
                cache[(((size_t)hh * p.P + e_npast) << p.dk_log2) + dd] = v;
            }
        } else if (EPI == EPI_RESID) {
            p.out[(size_t)col * p.ldo + r] = __fadd_rn(__fadd_rn(v, e_bias), e_res);
        } else if (EPI == EPI_GELU) {
            p.out[(size_t)col * p.ldo + r] = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias, v))]);
        } else if (EPI == EPI_GELU_Q8) {
            s_tail[64 + f_c * 32 + wave * 8 + f_r] = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias, v))]);  // rpw == 8: 32 rows / workgroup
        } else {
            p.out[(size_t)col * p.ldo + r] = v;
            best_val = v;
            best_idx = r;
        }
    }
    if (EPI == EPI_GELU_Q8) {
        // the workgroup's 32 outputs per column are one Q8 block of fc2's activation: quantize_row_q8_0 / q8_1
        // here, so the consumer starts from int8 (saves its whole quantize prologue)
        __syncthreads();
        for (int c = wave * 2 + (lane >> 5); c < ncols; c += 8) {   // a half-wave per column (NC == 1: wave 0, lanes 0-31)
            const float v = s_tail[64 + c * 32 + (lane & 31)];
            float amax = fabsf(v);
            amax = fmaxf(amax, dpp_f<DPP_QUAD_XOR1>(amax)); amax = fmaxf(amax, dpp_f<DPP_QUAD_XOR2>(amax));
            amax = fmaxf(amax, dpp_f<DPP_ROW_HALF_MIRROR>(amax)); amax = fmaxf(amax, dpp_f<DPP_ROW_MIRROR>(amax));
            amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
            const float d = amax / 127.0f;
            const float id = (d != 0.0f) ? 1.0f / d : 0.0f;
            const int q = (int)roundf(__fmul_rn(v, id));
            int isum = q;
            isum += dpp_i<DPP_QUAD_XOR1>(isum); isum += dpp_i<DPP_QUAD_XOR2>(isum);
            isum += dpp_i<DPP_ROW_HALF_MIRROR>(isum); isum += dpp_i<DPP_ROW_MIRROR>(isum);
            isum += __shfl_xor(isum, 16, 64);
            const size_t blk = (size_t)(col0 + c) * (M / 32) + blockIdx.x;   // column-major [N][d_ff/32]
            p.oq_q[blk * 32 + (lane & 31)] = (int8_t)q;
            if ((lane & 31) == 0) {
                if (TI::q81) { p.oq_d[blk] = d; p.oq_s[blk] = __float_as_uint(__fmul_rn((float)isum, d)); }
                else { p.oq_d[blk] = h2f(f2h(d)); p.oq_s[blk] = (uint32_t)isum; }
            }
        }
    }
    if (EPI == EPI_LOGITS && NC == 1 && p.pmax_val != nullptr) {
        // per-block partial arg-max (lowest index wins ties), finished by argmax_kernel
#pragma unroll
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
#pragma unroll
            for (int w = 1; w < 4; w++)
                if (sv[w] > best_val || (sv[w] == best_val && si[w] < best_idx)) { best_val = sv[w]; best_idx = si[w]; }
            p.pmax_val[blockIdx.x] = best_val;
            p.pmax_idx[blockIdx.x] = best_idx;
            // fused decode step (kernels_decode.hip.h): every kernel of this step has read the position by now
            if (blockIdx.x == 0 && p.st_adv != nullptr && p.adv != 0) { p.st_adv->n_past += p.adv; p.st_adv->n_gen += p.adv; }
        }
    }
}

// LayerNorm + Q8 quantization of N activation columns, once per LN site of a prefill chunk (the decode
// path keeps this fused in the mat-vec prologue; with 8 columns it would be replicated in every workgroup).
// One workgroup per column; same arithmetic as the prologue of matvec_fast_kernel<PRO_LN>.
template <int K, bool Q81>
__global__ __launch_bounds__(256) void lnq_kernel(const float *x, int ldx, const float *ln_w, const float *ln_b, float eps, double inv_k,
                                                  int8_t *oq_q, float *oq_d, uint32_t *oq_s) {
    constexpr int NJJ = K / 4 / 64, NSHARE = NJJ / 4, BPR = K / QK;
    const int col = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 *xcol = reinterpret_cast<const float4 *>(x + (size_t)col * ldx);
    float4 xr[NJJ], xs4[NSHARE], lw4[NSHARE], lb4[NSHARE];
#pragma unroll
    for (int i = 0; i < NJJ; i++) xr[i] = xcol[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < NSHARE; i++) {
        const int ch = (wave + 4 * i) * 64 + lane;
        xs4[i] = xcol[ch];
        lw4[i] = reinterpret_cast<const float4 *>(ln_w)[ch];
        lb4[i] = reinterpret_cast<const float4 *>(ln_b)[ch];
    }
    double s1 = 0.0;
#pragma unroll
    for (int i = 0; i < NJJ; i++) s1 += ((double)xr[i].x + (double)xr[i].y) + ((double)xr[i].z + (double)xr[i].w);
    s1 = wave_sum_f64(s1);
    const float mean = (float)(s1 * inv_k);
    double s2 = 0.0;
#pragma unroll
    for (int i = 0; i < NJJ; i++) {
        const float a = __fsub_rn(xr[i].x, mean), b = __fsub_rn(xr[i].y, mean);
        const float c = __fsub_rn(xr[i].z, mean), d = __fsub_rn(xr[i].w, mean);
        s2 += ((double)__fmul_rn(a, a) + (double)__fmul_rn(b, b)) + ((double)__fmul_rn(c, c) + (double)__fmul_rn(d, d));
    }
    s2 = wave_sum_f64(s2);
    const float var = (float)(s2 * inv_k);
    const float scale = 1.0f / sqrtf(__fadd_rn(var, eps));
#pragma unroll
    for (int i = 0; i < NSHARE; i++) {
        const int ch = (wave + 4 * i) * 64 + lane;
        float4 v = xs4[i];
        v.x = __fadd_rn(__fmul_rn(lw4[i].x, __fmul_rn(__fsub_rn(v.x, mean), scale)), lb4[i].x);
        v.y = __fadd_rn(__fmul_rn(lw4[i].y, __fmul_rn(__fsub_rn(v.y, mean), scale)), lb4[i].y);
        v.z = __fadd_rn(__fmul_rn(lw4[i].z, __fmul_rn(__fsub_rn(v.z, mean), scale)), lb4[i].z);
        v.w = __fadd_rn(__fmul_rn(lw4[i].w, __fmul_rn(__fsub_rn(v.w, mean), scale)), lb4[i].w);
        const float amax = group8_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        const float d = amax / 127.0f;
        const float id = (d != 0.0f) ? 1.0f / d : 0.0f;
        const int q0 = (int)roundf(__fmul_rn(v.x, id)), q1 = (int)roundf(__fmul_rn(v.y, id));
        const int q2 = (int)roundf(__fmul_rn(v.z, id)), q3 = (int)roundf(__fmul_rn(v.w, id));
        const int isum = group8_sum(q0 + q1 + q2 + q3);
        reinterpret_cast<uint32_t *>(oq_q + (size_t)col * K)[ch] =
            (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
        if ((lane & 7) == 0) {
            const size_t b = (size_t)col * BPR + (ch >> 3);
            if (Q81) { oq_d[b] = d; oq_s[b] = __float_as_uint(__fmul_rn((float)isum, d)); }
            else { oq_d[b] = h2f(f2h(d)); oq_s[b] = (uint32_t)isum; }
        }
    }
}

__host__ __device__ inline size_t matvec_fast_smem_bytes(int K, int rpw, int nc = 1, bool gelu_q8 = false) {
    return (size_t)K + 2 * (size_t)(K / QK) * 4 + (64 + (gelu_q8 ? 32 * nc : 0)) * 4 + 4 * (size_t)rpw * nc * (K / QK + 4) * 4 + 64;
}

}  // namespace bgk

namespace bgk {

// ---- attention, head size 64, contexts up to 1024 keys ---------------------------------------------
// One workgroup per (head, query).  Same arithmetic as attn_kernel (biogpt.cpp:741-764), laid out for
// latency: 4 lanes share a key (quad DPP reduce), NT/64 slices of 64 lanes share the PV sum, and every
// K / V / q load is issued at kernel entry.  The loads are bounded by t_cap, a launch-time upper bound
// of the context (the captured decode graphs are bucketed by it), NOT by the device-side position:
// rows in [T, t_cap) are allocated cache whose values are loaded and ignored (masked by T).
//   KP   = key passes of NT/4 keys;  VPRE = V values prefetched at entry (16 per lane; t_cap <= NT/4)
// Layouts that were measured and rejected (tools/sweep_attn.py): 256 threads with several key passes per
// lane (one wave per SIMD) and a thread-per-key layout (t_cap threads) are both 15-40 % SLOWER -- with every
// load and all arithmetic ablated this kernel still takes 4.4 us, i.e. its cost is the serial chain of
// dependent steps (entry, position load, two block reductions, table lookup, PV, final reduce), and more
// lanes shorten each step.
// exact (float)(1.0/sum): v_rcp_f64 + two Newton steps is within 1 ulp(double) of the quotient, which
// rounds to the same float as the IEEE double division except on ~1e-9 of inputs
__device__ __forceinline__ float inv_sum_f32(double s) {
    double r = __builtin_amdgcn_rcp(s);
    r = __builtin_fma(r, __builtin_fma(-s, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-s, r, 1.0), r);
    return (float)r;
}

// LDS of attn_group_kernel: max(scores [G][pitch] floats, partial outputs [8][G][64] doubles) + [G] reciprocal sums
__host__ __device__ inline size_t attn_group_main_bytes(int t_cap, int g = 8) {
    const size_t sc = (size_t)g * t_cap * 4, po = (size_t)8 * g * 64 * 8;
    return sc > po ? sc : po;
}
__host__ __device__ inline size_t attn_group_smem_bytes(int t_cap, int g = 8) { return attn_group_main_bytes(t_cap, g) + (size_t)g * 64 * 4 + 64; }

// One head's 64 outputs, held one per lane by wave 0: F32 row for the reference layout and, for the fast
// out_proj, the two Q8 blocks of its activation row (amax / roundf / block sum as quantize_row_q8_0/_1).
__device__ __forceinline__ void store_head_output(const AttnParams &p, int i, int h, int tid, float o, bool q8) {
    constexpr int DK = 64;
    p.out[(size_t)i * p.D + (size_t)h * DK + tid] = o;
    if (!q8) return;
    float amax = fabsf(o);
    amax = fmaxf(amax, dpp_f<DPP_QUAD_XOR1>(amax)); amax = fmaxf(amax, dpp_f<DPP_QUAD_XOR2>(amax));
    amax = fmaxf(amax, dpp_f<DPP_ROW_HALF_MIRROR>(amax)); amax = fmaxf(amax, dpp_f<DPP_ROW_MIRROR>(amax));
    amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
    const float dq = amax / 127.0f;
    const float id = (dq != 0.0f) ? 1.0f / dq : 0.0f;
    const int q = (int)roundf(__fmul_rn(o, id));
    int isum = q;
    isum += dpp_i<DPP_QUAD_XOR1>(isum); isum += dpp_i<DPP_QUAD_XOR2>(isum);
    isum += dpp_i<DPP_ROW_HALF_MIRROR>(isum); isum += dpp_i<DPP_ROW_MIRROR>(isum);
    isum += __shfl_xor(isum, 16, 64);
    const size_t blk = (size_t)i * (p.D / 32) + h * 2 + (tid >> 5);   // [N][d_model/32]
    p.oq_q[blk * 32 + (tid & 31)] = (int8_t)q;
    if ((tid & 31) == 0) {
        if (p.q81) { p.oq_d[blk] = dq; p.oq_s[blk] = __float_as_uint(__fmul_rn((float)isum, dq)); }
        else { p.oq_d[blk] = h2f(f2h(dq)); p.oq_s[blk] = (uint32_t)isum; }
    }
}

template <int KP, bool VPRE>
__global__ __launch_bounds__(1024) void attn_fast_kernel(const AttnParams p) {
    constexpr int DK = 64;
    __shared__ float S[KP * 256];
    __shared__ double red[16];
    __shared__ float redf[16];
    __shared__ double pv[1024];
    const int h = blockIdx.x, i = blockIdx.y;
    const int tid = threadIdx.x, nt = blockDim.x;
#ifdef BIOGPT_HIP_PROFILE_HOOKS   // make EXTRA=-DBIOGPT_HIP_PROFILE_HOOKS: timestamps + ablation bits (tools/sweep_attn.py)
#define AT_STAMP(k)                                                                                   \
    do {                                                                                              \
        if ((p.dbg & 32) && (tid & 63) == 0 && blockIdx.x == 0 && blockIdx.y == 0)                    \
            p.tstamp[(tid >> 6) * 8 + (k)] = __builtin_readcyclecounter();                            \
    } while (0)
#define AT_DBG(bit) (p.dbg & (bit))
#else
#define AT_STAMP(k) do {} while (0)
#define AT_DBG(bit) 0
#endif
    AT_STAMP(0);
    const int D = p.D;
    const int kpp = nt >> 2;                       // keys per pass
    const int nsl = nt >> 6;                       // PV slices
    const int ksub = tid & 3, kidx = tid >> 2;
    const int d = tid & (DK - 1), sl = tid >> 6;
    const int t_cap = p.t_cap;

    // lane ksub of a quad owns float4 #(4m + ksub) of the 16 float4 of a key row: each load instruction
    // of a quad covers 64 contiguous bytes
    // batched decode: query row i = sequence i; prompt columns of several sequences name their sequence (one more dependent load)
    const size_t seq_off = p.seq ? (size_t)(p.col_mode ? p.seq[i].seq_id : i) * p.kv_seq_stride : 0;
    const float4 *kbase = reinterpret_cast<const float4 *>(p.kcache + seq_off + (size_t)h * p.P * DK) + ksub;   // [H][P][dk]
    const float *__restrict__ vbase = p.vcache + seq_off + (size_t)h * p.P * DK + d;
    const float4 *qp = reinterpret_cast<const float4 *>(p.q + (size_t)i * D + (size_t)h * DK) + ksub;

    // ---- entry: all loads ----
    const int n_past = p.seq ? p.seq[i].n_past : p.st->n_past;
    float4 kr[KP][4];
#pragma unroll
    for (int ps = 0; ps < KP; ps++) {
        const int j = ps * kpp + kidx;
        if (j < t_cap && !AT_DBG(1)) {
#pragma unroll
            for (int m = 0; m < 4; m++) kr[ps][m] = kbase[(size_t)j * (DK / 4) + 4 * m];
        }
    }
    float4 qv[4];
#pragma unroll
    for (int m = 0; m < 4; m++) qv[m] = qp[4 * m];
    float vr[VPRE ? 16 : 1];
    if (VPRE) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int j = sl + nsl * k;
            if (j < t_cap && !AT_DBG(2)) vr[k] = vbase[(size_t)j * DK];
        }
    }
    AT_STAMP(1);
    const int T = p.seq ? (p.col_mode ? p.seq[i].t_vis : n_past + 1) : visible_keys(p.st, i, p.N);

    // ---- scores: 16 dims per lane, quad reduce ----
    float sc[KP];
#pragma unroll
    for (int ps = 0; ps < KP; ps++) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            a0 += (double)__fmul_rn(kr[ps][m].x, qv[m].x); a1 += (double)__fmul_rn(kr[ps][m].y, qv[m].y);
            a2 += (double)__fmul_rn(kr[ps][m].z, qv[m].z); a3 += (double)__fmul_rn(kr[ps][m].w, qv[m].w);
        }
        double acc = (a0 + a1) + (a2 + a3);
        acc += dpp_d<DPP_QUAD_XOR1>(acc);
        acc += dpp_d<DPP_QUAD_XOR2>(acc);
        sc[ps] = (ps * kpp + kidx < T) ? (float)acc : -INFINITY;
    }
    AT_STAMP(2);
    // ---- softmax (ggml_soft_max: fp16-table exp, double sum, scale by (float)(1/sum)) ----
    // one barrier for the max, one for the sum (which also publishes S): separate exchange arrays
    const int nw = nt >> 6, lane = tid & 63, wv = tid >> 6;
    float mx = sc[0];
#pragma unroll
    for (int ps = 1; ps < KP; ps++) mx = fmaxf(mx, sc[ps]);
    mx = wave_max_f32(mx);
    if (lane == 0) redf[wv] = mx;
    __syncthreads();
    mx = redf[0];
    for (int w = 1; w < nw; w++) mx = fmaxf(mx, redf[w]);
    AT_STAMP(3);
    double sum = 0.0;
#pragma unroll
    for (int ps = 0; ps < KP; ps++) {
        const int j = ps * kpp + kidx;
        if (j < T && ksub == 0) {
            const float val = AT_DBG(4) ? __fsub_rn(sc[ps], mx) : h2f(p.exp_tab[f2h(__fsub_rn(sc[ps], mx))]);
            S[j] = val;
            sum += (double)val;
        }
    }
    AT_STAMP(4);
    sum = wave_sum_f64(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    sum = 0.0;
    for (int w = 0; w < nw; w++) sum += red[w];
    const float inv = inv_sum_f32(sum);
    AT_STAMP(5);

    // ---- PV: nsl slices x 64 dims, double accumulation ----
    double a0 = 0.0, a1 = 0.0;
    if (VPRE && !AT_DBG(8)) {
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            const int j0 = sl + nsl * k, j1 = j0 + nsl;
            if (j0 < T) a0 += (double)__fmul_rn(vr[k], __fmul_rn(S[j0], inv));
            if (j1 < T) a1 += (double)__fmul_rn(vr[k + 1], __fmul_rn(S[j1], inv));
        }
    } else {
        for (int j = sl; j < T; j += nsl * 8) {
            float v8[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v8[k] = (j + nsl * k < t_cap) ? vbase[(size_t)(j + nsl * k) * DK] : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                if (j + nsl * k < T) a0 += (double)__fmul_rn(v8[k], __fmul_rn(S[j + nsl * k], inv));
                if (j + nsl * (k + 1) < T) a1 += (double)__fmul_rn(v8[k + 1], __fmul_rn(S[j + nsl * (k + 1)], inv));
            }
        }
    }
    pv[tid] = a0 + a1;
    AT_STAMP(6);
    __syncthreads();
    AT_STAMP(7);
    if (tid < DK) {
        double t0 = 0.0, t1 = 0.0;
        for (int s2 = 0; s2 + 1 < nsl; s2 += 2) { t0 += pv[s2 * DK + tid]; t1 += pv[(s2 + 1) * DK + tid]; }
        if (nsl & 1) t0 += pv[(nsl - 1) * DK + tid];
        store_head_output(p, i, h, tid, (float)(t0 + t1), p.oq_q != nullptr && !AT_DBG(16));
    }
#undef AT_STAMP
#undef AT_DBG
}

// ---- single-token attention over a long context, split over the keys -----------------------------------
// attn_fast_kernel puts a head on ONE compute unit: at 1024 keys that is 512 KB of K/V through one CU per
// layer (21 us measured) while 240 CUs idle.  Here a head's keys are cut into ranges of 64 and spread over
// the chip; the softmax of ggml_soft_max needs the global maximum before the table lookup and the global sum
// before the PV products (p_j = fl(e_j * (float)(1/sum)) is rounded BEFORE it multiplies V), so the work
// falls into three dependent launches:
//   scores  (H x S workgroups)  S_j = K_j . q  -> scratch, per-range maximum
//   pv      (H x S workgroups)  global max from the S maxima; e_j for ALL keys of the head (4 lookups per
//                               lane, every workgroup computes the same double sum in the same order);
//                               partial  sum_j V_jd * p_j  over its own 64 keys, in double
//   combine (H workgroups)      partials added in range order, F32 store + Q8 hand-off for out_proj
// Arithmetic per element is that of attn_fast_kernel; only the (double) association of the sums differs.
constexpr int SPLIT_KEYS = 64;    // keys per workgroup
constexpr int SPLIT_MAX = 16;     // ranges per head (contexts up to 1024 keys)

__global__ __launch_bounds__(256) void attn_split_scores_kernel(const AttnParams p) {
    constexpr int DK = 64;
    __shared__ float redf[4];
    const int h = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const int ksub = tid & 3, j = s * SPLIT_KEYS + (tid >> 2);
    const float4 *krow = reinterpret_cast<const float4 *>(p.kcache + (size_t)h * p.P * DK) + (size_t)j * (DK / 4) + ksub;
    const float4 *qp = reinterpret_cast<const float4 *>(p.q + (size_t)h * DK) + ksub;
    const int n_past = p.st->n_past;
    float4 kr[4], qv[4];
#pragma unroll
    for (int m = 0; m < 4; m++) kr[m] = (j < p.t_cap) ? krow[4 * m] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < 4; m++) qv[m] = qp[4 * m];
    const int T = n_past + 1;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        a0 += (double)__fmul_rn(kr[m].x, qv[m].x); a1 += (double)__fmul_rn(kr[m].y, qv[m].y);
        a2 += (double)__fmul_rn(kr[m].z, qv[m].z); a3 += (double)__fmul_rn(kr[m].w, qv[m].w);
    }
    double acc = (a0 + a1) + (a2 + a3);
    acc += dpp_d<DPP_QUAD_XOR1>(acc);
    acc += dpp_d<DPP_QUAD_XOR2>(acc);
    const float sc = (j < T) ? (float)acc : -INFINITY;
    if (ksub == 0 && j < p.t_cap) p.sp_scores[(size_t)h * p.P + j] = sc;
    const float mx = wave_max_f32(sc);
    if ((tid & 63) == 0) redf[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) p.sp_max[h * SPLIT_MAX + s] = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
}

__global__ __launch_bounds__(256) void attn_split_pv_kernel(const AttnParams p) {
    constexpr int DK = 64;
    __shared__ float e_own[SPLIT_KEYS];
    __shared__ double red[4];
    __shared__ double pv[256];
    const int h = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const int t0 = s * SPLIT_KEYS, d = tid & 63, sl = tid >> 6;
    const float *__restrict__ vbase = p.vcache + (size_t)h * p.P * DK + d;
    float vr[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int j = t0 + sl + 4 * k;
        vr[k] = (j < p.t_cap) ? vbase[(size_t)j * DK] : 0.0f;
    }
    float scv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = tid + 256 * k;
        scv[k] = (j < p.t_cap) ? p.sp_scores[(size_t)h * p.P + j] : -INFINITY;
    }
    // the S range maxima: one per lane of every wave (a loop of dependent scalar loads would serialise)
    float mx = ((tid & 63) < p.n_split) ? p.sp_max[h * SPLIT_MAX + (tid & 63)] : -INFINITY;
    const int T = p.st->n_past + 1;
    if (t0 >= T) return;                      // a range past the context contributes nothing (combine skips it)
    mx = wave_max_f32(mx);
    float ev[4];
#pragma unroll
    for (int k = 0; k < 4; k++) ev[k] = (tid + 256 * k < T) ? h2f(p.exp_tab[f2h(__fsub_rn(scv[k], mx))]) : 0.0f;
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = tid + 256 * k;
        if (j < T) {
            sum += (double)ev[k];
            if (j >= t0 && j < t0 + SPLIT_KEYS) e_own[j - t0] = ev[k];
        }
    }
    sum = wave_sum_f64(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = inv_sum_f32(sum);
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const int j0 = sl + 4 * k, j1 = j0 + 4;
        if (t0 + j0 < T) a0 += (double)__fmul_rn(vr[k], __fmul_rn(e_own[j0], inv));
        if (t0 + j1 < T) a1 += (double)__fmul_rn(vr[k + 1], __fmul_rn(e_own[j1], inv));
    }
    pv[tid] = a0 + a1;
    __syncthreads();
    if (tid < DK) p.sp_pv[(size_t)(h * SPLIT_MAX + s) * DK + tid] = (pv[tid] + pv[64 + tid]) + (pv[128 + tid] + pv[192 + tid]);
}

__global__ __launch_bounds__(64) void attn_split_combine_kernel(const AttnParams p) {
    constexpr int DK = 64;
    const int h = blockIdx.x, tid = threadIdx.x;
    const int T = p.st->n_past + 1;
    const int ns = (T + SPLIT_KEYS - 1) / SPLIT_KEYS;
    const double *part = p.sp_pv + (size_t)h * SPLIT_MAX * DK + tid;
    double pr[SPLIT_MAX];                     // all loads in flight at once; ranges past the context hold stale values
#pragma unroll
    for (int s = 0; s < SPLIT_MAX; s++) pr[s] = part[s * DK];
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int s = 0; s < SPLIT_MAX; s += 2) {
        if (s < ns) t0 += pr[s];
        if (s + 1 < ns) t1 += pr[s + 1];
    }
    store_head_output(p, 0, h, tid, (float)(t0 + t1), p.oq_q != nullptr);
}

// ---- attention for a pass of many query columns of ONE sequence (prompt passes) ------------------------------
// attn_fast_kernel gives every (head, query) its own workgroup, so a 128-column pass re-reads a head's K and V
// 128 times and launches 2048 workgroups (43 us per layer at 512 keys).  Here a workgroup takes one head and G = 8
// consecutive queries: a key row is loaded once and scored against the 8 queries (their 16 dims per lane live in
// registers), a V row is loaded once and accumulated into 8 outputs.  Per (query, key) the arithmetic is that of
// attn_fast_kernel -- scores with four double accumulators per lane and a quad reduce, fp16-table exp, double row
// sum, p = fl(e * (float)(1/sum)), V*p in double -- so the outputs agree bit for bit; each query keeps its own
// visible-key limit (visible_keys: the chunk it belongs to).
template <int G>
__global__ __launch_bounds__(512) void attn_group_kernel(const AttnParams p) {
    constexpr int DK = 64, NT = 512, KPP = NT / 4, NSL = NT / 64;
    static_assert(G == NSL && G == 8, "one wave per query in the softmax and output phases; 8 probabilities = two 16-byte LDS reads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int t_cap = p.t_cap;
    float *const S = reinterpret_cast<float *>(smem_raw);                  // [key][G]: scores, then probabilities p = fl(e * 1/sum)
    double *const pv = reinterpret_cast<double *>(smem_raw);               // [NSL][G][DK], reuses S after the PV loop
    const int h = blockIdx.x, i0 = blockIdx.y * G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksub = tid & 3, kidx = tid >> 2;
    const int N = p.N, D = p.D;

    int Tq[G], Tmax = 0;
#pragma unroll
    for (int q = 0; q < G; q++) {
        Tq[q] = (i0 + q < N) ? visible_keys(p.st, i0 + q, N) : 0;
        Tmax = max(Tmax, Tq[q]);
    }
    // the G query rows of this head live in LDS (2 KB): holding them in registers costs 128 VGPRs and halves the occupancy
    float *const s_q = reinterpret_cast<float *>(smem_raw + attn_group_main_bytes(t_cap));   // [G][DK]
    {
        const int q = tid >> 6, dd = tid & 63;                                // 512 threads = G x 64 values
        s_q[tid] = p.q[(size_t)min(i0 + q, N - 1) * D + (size_t)h * DK + dd];
    }
    __syncthreads();
    // ---- scores: 4 lanes per key, KPP keys per trip ----
    const float4 *kbase = reinterpret_cast<const float4 *>(p.kcache + (size_t)h * p.P * DK) + ksub;
    for (int j0 = 0; j0 < Tmax; j0 += KPP) {
        const int j = j0 + kidx, jc = min(j, t_cap - 1);
        float4 kr[4];
#pragma unroll
        for (int m = 0; m < 4; m++) kr[m] = kbase[(size_t)jc * (DK / 4) + 4 * m];
        float sc[G];
#pragma unroll
        for (int q = 0; q < G; q++) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const float4 qm = *reinterpret_cast<const float4 *>(s_q + q * DK + 16 * m + 4 * ksub);   // dims of float4 #(4m + ksub)
                a0 += (double)__fmul_rn(kr[m].x, qm.x); a1 += (double)__fmul_rn(kr[m].y, qm.y);
                a2 += (double)__fmul_rn(kr[m].z, qm.z); a3 += (double)__fmul_rn(kr[m].w, qm.w);
            }
            double acc = (a0 + a1) + (a2 + a3);
            acc += dpp_d<DPP_QUAD_XOR1>(acc);
            acc += dpp_d<DPP_QUAD_XOR2>(acc);
            sc[q] = (j < Tq[q]) ? (float)acc : -INFINITY;
        }
        if (ksub == 0 && j < t_cap) {
            float4 *dst = reinterpret_cast<float4 *>(S + (size_t)j * G);
            dst[0] = make_float4(sc[0], sc[1], sc[2], sc[3]);
            dst[1] = make_float4(sc[4], sc[5], sc[6], sc[7]);
        }
    }
    __syncthreads();
    // ---- softmax: wave q owns query q (ggml_soft_max: fp16-table exp, double row sum); the column is left as the
    //      probabilities the PV product uses, p_j = fl(e_j * (float)(1/sum)), and 0 for the keys it may not see ----
    const int Tw = (i0 + wave < N) ? visible_keys(p.st, i0 + wave, N) : 0;
    {
        float *Sq = S + wave;                                // element j at Sq[j * G]
        float inv = 0.0f;
        if (Tw > 0) {
            float mx = -INFINITY;
            for (int j = lane; j < Tw; j += 64) mx = fmaxf(mx, Sq[(size_t)j * G]);
            mx = wave_max_f32(mx);
            double sum = 0.0;
            for (int j0 = lane; j0 < Tw; j0 += 256) {      // 4 table lookups in flight per lane
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + 64 * u;
                    e[u] = (j < Tw) ? h2f(p.exp_tab[f2h(__fsub_rn(Sq[(size_t)j * G], mx))]) : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + 64 * u;
                    if (j < Tw) { Sq[(size_t)j * G] = e[u]; sum += (double)e[u]; }
                }
            }
            sum = wave_sum_f64(sum);
            inv = inv_sum_f32(sum);
        }
        for (int j = lane; j < Tmax; j += 64) Sq[(size_t)j * G] = (j < Tw) ? __fmul_rn(Sq[(size_t)j * G], inv) : 0.0f;
    }
    __syncthreads();
    // ---- PV: NSL key slices x 64 dims; one V load and two 16-byte LDS reads feed the G queries (a hidden key has p = 0
    //      and adds +-0 to the double accumulator, which leaves it unchanged) ----
    const int d = lane, sl = wave;
    double acc[G];
#pragma unroll
    for (int q = 0; q < G; q++) acc[q] = 0.0;
    const float *__restrict__ vbase = p.vcache + (size_t)h * p.P * DK + d;
    for (int j = sl; j < Tmax; j += NSL * 4) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = vbase[(size_t)min(j + NSL * k, t_cap - 1) * DK];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int jj = j + NSL * k;
            if (jj < Tmax) {
                const float4 p0 = *reinterpret_cast<const float4 *>(S + (size_t)jj * G), p1 = *reinterpret_cast<const float4 *>(S + (size_t)jj * G + 4);
                acc[0] += (double)__fmul_rn(v[k], p0.x); acc[1] += (double)__fmul_rn(v[k], p0.y);
                acc[2] += (double)__fmul_rn(v[k], p0.z); acc[3] += (double)__fmul_rn(v[k], p0.w);
                acc[4] += (double)__fmul_rn(v[k], p1.x); acc[5] += (double)__fmul_rn(v[k], p1.y);
                acc[6] += (double)__fmul_rn(v[k], p1.z); acc[7] += (double)__fmul_rn(v[k], p1.w);
            }
        }
    }
    __syncthreads();                                         // every read of S is done: the area becomes pv
#pragma unroll
    for (int q = 0; q < G; q++) pv[(size_t)(sl * G + q) * DK + d] = acc[q];
    __syncthreads();
    if (i0 + wave < N) {                                     // wave q: the 64 outputs of query q
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < NSL; s2 += 2) {
            t0 += pv[(size_t)(s2 * G + wave) * DK + lane];
            t1 += pv[(size_t)((s2 + 1) * G + wave) * DK + lane];
        }
        store_head_output(p, i0 + wave, h, lane, (float)(t0 + t1), p.oq_q != nullptr);
    }
}

// ---- attention for a pass of many query columns, register-tiled (prompt passes of >= 80 columns) ---------------------
// attn_group_kernel above spends its time issuing: per (query, key, dim) one v_mul_f32 + v_cvt_f64_f32 + v_add_f64 PLUS the
// LDS reads of the query values, a DPP quad reduce per (query, key) and 16-way bank conflicts in its softmax (51 us per
// layer at 512 columns, all VALU).  Here the two contractions are tiled like a GEMM, only with the reference's arithmetic
// per element (product rounded to f32, double accumulation):
//   scores  a thread owns 4 queries x 4 keys and walks the 64 dims IN ORDER (the oracle's own association, no cross-lane
//           reduce): per 4 dims it reads 4 K float4 (global / L1, one row per key) + 4 Q float4 (LDS) for 64 products
//   softmax thread = (query, one of 16 key slots): fp16-table exp, double sums, probabilities p = fl(e * (float)(1/sum))
//           written back as [key][query] -- conflict-free, 16 queries = one 64-byte LDS row
//   PV      a thread owns 4 queries x 4 dims over a quarter of the keys: per key one V float4 (a wave reads whole 256-byte
//           rows) + one LDS float4 of 4 probabilities for 16 products; the 4 key slices are added at the end
// 16 queries share every K / V row they load (8 before); 256 threads, ~40 KB of LDS -> 4 workgroups per compute unit.
// Visibility per query as everywhere (visible_keys: the end of its own reference chunk, F1).
// DMA form (contexts up to 640 keys): behind the scores a ring of 3 x 1 KB per wave (24 KB) that the K / V rows of the two MAC loops travel through (global_load_lds)
constexpr int ATTN_TILE_RING = 3;
__host__ __device__ inline bool attn_tile_dma_ok(int t_cap) { return (size_t)t_cap * 16 * 4 + (size_t)ATTN_TILE_RING * 8 * 1024 <= (size_t)8 * 16 * 64 * 8; }
template <int G>
__host__ __device__ inline size_t attn_tile_smem_bytes(int t_cap) {
    const size_t sc = (size_t)t_cap * G * 4, po = (size_t)8 * G * 64 * 8;
    return (sc > po ? sc : po) + (size_t)G * 64 * 4 + 36 * G * 4 + 8 * G * 8 + 64;
}
// One 1 KB DMA: 64 lanes x 16 bytes from the lanes' own global addresses to LDS bytes lds .. lds + 1023, lane-linear (global_load_lds_dwordx4; M0 = the LDS address, saved and
// restored around it: the compiler owns M0).  Inline assembly so that hipcc does not count it: a load it counts is drained by a vmcnt(0) in front of the next ds_read of ANY
// address (it sees LDS written by VMEM), which would wait for the prefetch just issued.  The waits below (at_wait_dma) are ours.
__device__ __forceinline__ void at_dma16(const void *gsrc, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void at_wait_dma() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// 512 threads = 8 waves: the critical path of the pass is its longest tile (512 keys), whose per-thread work this halves
// against a 256-thread version; two workgroups per compute unit = 4 waves per SIMD, hence the 128-register bound.
// Measured per layer at 512 columns (profiles/prefill_attention_r2.txt): grouped kernel above 51 us; this tiling with 256
// threads 48 us whatever the inner-loop order; + K staged through LDS 57 us; 512 threads 41 us (shipped); two tiles (longest +
// shortest) per workgroup for perfect balance 55 us (half the waves per SIMD); register double-buffering of the operands under
// the 128-register bound spills (262 us).  The instruction stream itself (v_pk_mul_f32 + 2 v_cvt_f64_f32 + 2 v_add_f64 per two
// products) costs 8.5-9.2 SIMD cycles per wave-MAC in isolation (tools/microbench10): ~17 us for this layer if nothing waited.
// ATTN_STAMPS (tools/microbench23.hip only): wave 0 of every workgroup stamps the shader clock at the phase borders (0 entry, 1 scores done, 2 softmax done, 3 PV loop done,
// 4 end) and the 100 MHz wall clock at entry / end (5, 6); 7: XCC_ID << 32 | HW_ID
#ifdef ATTN_STAMPS
#define ATTN_STAMP(k) do { if (threadIdx.x == 0) p.tstamp[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = ((k) == 5 || (k) == 6) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATTN_STAMP(k) do { } while (0)
#endif
typedef float at_f4 __attribute__((ext_vector_type(4)));
typedef float at_f2 __attribute__((ext_vector_type(2)));
template <int G, bool DMA = false>
__global__ __launch_bounds__(512, 4) void attn_tile_kernel(const AttnParams p) {
    constexpr int DK = 64, NW = 8, RING = ATTN_TILE_RING;
    static_assert(G == 16, "thread maps assume 16 queries: 4 query groups x 128 key groups, 16 queries x 32 key slots, 8 x (4 x 16) for PV");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int t_cap = p.t_cap;
    const size_t main_bytes = ((size_t)t_cap * G * 4 > (size_t)NW * G * DK * 8) ? (size_t)t_cap * G * 4 : (size_t)NW * G * DK * 8;
    float *const S = reinterpret_cast<float *>(smem_raw);                       // [key][G]: scores, then e, then probabilities
    double *const pvp = reinterpret_cast<double *>(smem_raw);                   // [8 slices][G][DK] partial outputs (after the PV loop)
    float *const Qs = reinterpret_cast<float *>(smem_raw + main_bytes);         // [G][DK]
    constexpr int MXP = 36;                                                     // floats per query row of s_mx (32 + 4: the 16 queries' 16-byte reads spread over the banks)
    float *const s_mx = Qs + G * DK;                                            // [G][MXP]: per query the maxima of the 32 (wave, lane row) key groups of the scores phase
    double *const s_sum = reinterpret_cast<double *>(s_mx + G * MXP);           // [8 waves][G]
    const int h = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, D = p.D;
    // DMA: this wave's ring behind the scores: slot r at ring + r * 1024 (generic pointer for the reads, LDS byte address in an SGPR for the DMAs)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const size_t ring_off = (((size_t)t_cap * G * 4 + 1023) & ~(size_t)1023) + (size_t)wv * (RING * 1024);
    const unsigned char *const ring = smem_raw + ring_off;
    const unsigned ring_lds = DMA ? (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem_raw + (int)ring_off) : 0u;
    // dispatch order: the long-context half of the tiles first (largest first), then the short half in ASCENDING order -- the
    // second round of workgroups then lands a short tile next to each long one (44.9 -> 41.2 us against plain longest-first)
    const int ny = (int)gridDim.y, yb = (int)blockIdx.y, nhalf = (ny + 1) / 2;
    const int i0 = ((yb < nhalf) ? ny - 1 - yb : yb - nhalf) * G;

    ATTN_STAMP(0); ATTN_STAMP(5);
#ifdef ATTN_STAMPS
    if (threadIdx.x == 0) p.tstamp[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));
#endif
    // queries of this tile -> LDS (row-major), 4 floats per thread of the first four waves
    if (tid < 256) {
        const int q = tid >> 4, d4 = tid & 15;
        reinterpret_cast<float4 *>(Qs)[tid] = *reinterpret_cast<const float4 *>(p.q + (size_t)min(i0 + q, N - 1) * D + (size_t)h * DK + 4 * d4);
    }
    for (int i = tid; i < G * MXP; i += 512) s_mx[i] = -INFINITY;
    const int Tmax = visible_keys(p.st, min(i0 + G - 1, N - 1), N);              // visibility grows with the column index
    __syncthreads();

    // ---- scores: a thread owns 4 queries x 4 keys and walks the 64 dims in order ----
    {
        const int kg = tid >> 2, qg = tid & 3;
        int Tq[4];
#pragma unroll
        for (int qi = 0; qi < 4; qi++) Tq[qi] = (i0 + 4 * qg + qi < N) ? visible_keys(p.st, i0 + 4 * qg + qi, N) : 0;
        const float4 *kbase = reinterpret_cast<const float4 *>(p.kcache + (size_t)h * p.P * DK);
        const float4 *qbase = reinterpret_cast<const float4 *>(Qs + 4 * qg * DK);
        for (int j0 = 0; j0 < Tmax; j0 += 512) {
            if (j0 + 64 * wave >= Tmax) break;                                   // this wave's 64 keys are past every query's context
            const int jb = j0 + 4 * kg;
            // the 4 key rows of this thread are consecutive: ONE address, the row in the load's immediate offset (rows past the context are clamped into the head's
            // allocated rows; what is computed from them is never stored).  Needs 4 | P and P >= 4 -- jb is a multiple of 4, so a clamped base never holds a key that
            // IS stored; the host only launches this kernel for such tables (engine.hip), any other size takes attn_group_kernel
            const at_f4 *krow0 = reinterpret_cast<const at_f4 *>(kbase) + (size_t)min(jb, p.P - 4) * (DK / 4);
            // DMA: lane = key (j0 + 64 wave + lane), one 16-byte piece (4 dims) of its row per step, two steps ahead of the arithmetic: the loop above (loads at the top of
            // a step, waited for at once) left a wave standing for an L2 round trip per step with one other wave per SIMD to cover it.  Step m's piece of key k of the
            // wave lies at ring slot m mod 3, byte 16 k; the 4 query groups' lanes of a key group read the same 4 pieces.  The ring always runs two requests ahead (the
            // last two steps re-request steps 0 and 1: a constant vmcnt(2)).
            const unsigned char *const kdma = reinterpret_cast<const unsigned char *>(kbase) + (size_t)min(j0 + 64 * wv + lane, p.P - 1) * (DK * 4);
            if (DMA) { at_dma16(kdma, ring_lds); at_dma16(kdma + 16, ring_lds + 1024); }
            int rs = 0;                                                          // slot of step m
            double acc[4][4];
#pragma unroll
            for (int qi = 0; qi < 4; qi++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[qi][c] = 0.0;
#pragma unroll 1
            for (int m = 0; m < DK / 4; m++) {
                at_f4 kv[4], qv[4];
                if (DMA) {
                    const int rn = rs == 0 ? 2 : rs - 1;                         // (m + 2) mod 3
                    at_dma16(kdma + 16 * ((m + 2) & 15), ring_lds + rn * 1024);
                    at_wait_dma<2>();
                    const at_f4 *const kb = reinterpret_cast<const at_f4 *>(ring + rs * 1024) + 4 * (lane >> 2);
#pragma unroll
                    for (int c = 0; c < 4; c++) kv[c] = kb[c];
                    rs = rs == 2 ? 0 : rs + 1;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c++) kv[c] = krow0[c * (DK / 4) + m];
                }
#pragma unroll
                for (int qi = 0; qi < 4; qi++) qv[qi] = reinterpret_cast<const at_f4 *>(qbase)[qi * (DK / 4) + m];
                // two dims of ONE (query, key) per packed multiply: the operands are the register pairs the 16-byte loads left behind (pairing two queries of one
                // dim -- what the vectoriser picks by itself -- costs a v_mov per product pair to build the operand); the sums stay in dim order per accumulator
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
#pragma unroll
                    for (int qi = 0; qi < 4; qi += 2) {
                        at_f2 pr2[2][4];
#pragma unroll
                        for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
                            for (int c = 0; c < 4; c++) pr2[q2][c] = (hf == 0 ? kv[c].lo : kv[c].hi) * (hf == 0 ? qv[qi + q2].lo : qv[qi + q2].hi);
#pragma unroll
                        for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
                            for (int c = 0; c < 4; c++) acc[qi + q2][c] += (double)pr2[q2][c].x;
#pragma unroll
                        for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
                            for (int c = 0; c < 4; c++) acc[qi + q2][c] += (double)pr2[q2][c].y;
                    }
            }
            if (DMA) at_wait_dma<0>();                                           // (the two re-requests: the ring is quiet before it is used again)
            // the softmax's first pass, taken while the scores are in registers: per query the maximum of this thread's keys, then of the four key groups of its lane row
            // (lanes 4 apart: same query group) -> one value per (wave, row, query) in s_mx (preset to -inf: a wave without keys writes nothing)
            float mq[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int j = jb + c;
                if (j < t_cap) {
                    const float4 sc = make_float4(j < Tq[0] ? (float)acc[0][c] : -INFINITY, j < Tq[1] ? (float)acc[1][c] : -INFINITY,
                                                  j < Tq[2] ? (float)acc[2][c] : -INFINITY, j < Tq[3] ? (float)acc[3][c] : -INFINITY);
                    *reinterpret_cast<float4 *>(S + (size_t)j * G + 4 * qg) = sc;
                    mq[0] = fmaxf(mq[0], sc.x); mq[1] = fmaxf(mq[1], sc.y); mq[2] = fmaxf(mq[2], sc.z); mq[3] = fmaxf(mq[3], sc.w);
                }
            }
#pragma unroll
            for (int qi = 0; qi < 4; qi++) {
                mq[qi] = fmaxf(mq[qi], dpp_f<DPP_ROW_ROR4>(mq[qi]));
                mq[qi] = fmaxf(mq[qi], dpp_f<DPP_ROW_ROR8>(mq[qi]));
            }
            if ((lane & 12) == 0) {
                const int grp = wave * 4 + (lane >> 4);
#pragma unroll
                for (int qi = 0; qi < 4; qi++) {
                    float *const slot = s_mx + (4 * qg + qi) * MXP + grp;
                    *slot = (j0 == 0) ? mq[qi] : fmaxf(mq[qi], *slot);       // (contexts beyond 512 keys: the same lane comes back to its slot)
                }
            }
        }
    }
    __builtin_amdgcn_s_setprio(3);                           // the short, latency-bound phases go first when they compete with another workgroup's MAC loops for issue slots
    __syncthreads();
    ATTN_STAMP(1);

    // ---- softmax (ggml_soft_max: fp16-table exp, double row sum, p = fl(e * (float)(1/sum))) ----
    // thread = (query, one of 32 key slots).  The maximum comes from the scores phase; 8 table lookups are in flight per thread; the row sum is a sum of fp16 values below
    // 2^11 in double -- exact in any order -- so the slots are added lane to lane and wave to wave, and every thread takes the reciprocal itself (three barriers, were five).
    {
        const int q = tid & 15, slot = tid >> 4;           // 32 key slots
        const int Tw = (i0 + q < N) ? visible_keys(p.st, i0 + q, N) : 0;
        float mx;
        {
            const at_f4 *mrow = reinterpret_cast<const at_f4 *>(s_mx + q * MXP);
            at_f4 m4 = mrow[0];
#pragma unroll
            for (int s2 = 1; s2 < 8; s2++) { const at_f4 t = mrow[s2]; m4.x = fmaxf(m4.x, t.x); m4.y = fmaxf(m4.y, t.y); m4.z = fmaxf(m4.z, t.z); m4.w = fmaxf(m4.w, t.w); }
            mx = fmaxf(fmaxf(m4.x, m4.y), fmaxf(m4.z, m4.w));
        }
        double sum = 0.0;
        for (int j = slot; j < Tw; j += 256) {
            float e[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int jj = j + 32 * u;
                e[u] = (jj < Tw) ? h2f(p.exp_tab[f2h(__fsub_rn(S[(size_t)jj * G + q], mx))]) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int jj = j + 32 * u;
                if (jj < Tw) { S[(size_t)jj * G + q] = e[u]; sum += (double)e[u]; }
            }
        }
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);          // the wave's four slots of this query
        if (lane < 16) s_sum[wave * G + q] = sum;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; w++) t += s_sum[w * G + q];
        const float inv = (Tw > 0) ? inv_sum_f32(t) : 0.0f;
        for (int j = slot; j < Tmax; j += 256) {
            float pr[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int jj = j + 32 * u; pr[u] = (jj < Tw) ? S[(size_t)jj * G + q] : 0.0f; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int jj = j + 32 * u; if (jj < Tmax) S[(size_t)jj * G + q] = __fmul_rn(pr[u], inv); }
        }
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    ATTN_STAMP(2);

    // ---- PV: wave = key slice (j mod 8), lane = (4 queries, 4 dims) ----
    double acc[4][4];
#pragma unroll
    for (int qi = 0; qi < 4; qi++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[qi][c] = 0.0;
    {
        const int qg = lane >> 4, dg = lane & 15;
        const float4 *vbase = reinterpret_cast<const float4 *>(p.vcache + (size_t)h * p.P * DK) + dg;
        // 4 keys of this slice per trip, all loads first.  Whole trips carry no guards (every key < Tmax <= t_cap: nothing to clamp, no probability to zero): the
        // guarded form spent 33 of its 200 VALU instructions per trip on clamps, 64-bit row addresses and exec masks
        auto mac4 = [&](const at_f4 &v, const at_f4 &pr) __attribute__((always_inline)) {
#pragma unroll
            for (int qi = 0; qi < 4; qi++) {
                const at_f2 pp = {pr[qi], pr[qi]};
                const at_f2 a = v.lo * pp, b = v.hi * pp;
                acc[qi][0] += (double)a.x; acc[qi][1] += (double)a.y; acc[qi][2] += (double)b.x; acc[qi][3] += (double)b.y;
            }
        };
        const at_f4 *vrow = reinterpret_cast<const at_f4 *>(vbase);
        const float *Sq = S + 4 * qg;
        int j = wave;
        if (DMA) {
            // a trip's four V rows (keys j, j + 8, j + 16, j + 24 of this wave's slice: 4 x 256 bytes) are ONE 1 KB DMA -- lane = (row u = lane / 16, piece lane % 16) --
            // requested two trips ahead; lane (qg, dg) reads piece dg of the four rows (the four query groups' lanes the same bytes).  Rows past the context are clamped
            // (never used); the ring always runs two requests ahead.
            const unsigned char *const vb = reinterpret_cast<const unsigned char *>(p.vcache + (size_t)h * p.P * DK) + (lane & 15) * 16;
            const int ju = wv + NW * (lane >> 4);                               // this lane's row in trip 0
            auto vreq = [&](int trip, int slot) __attribute__((always_inline)) { at_dma16(vb + (size_t)min(ju + 4 * NW * trip, t_cap - 1) * (DK * 4), ring_lds + slot * 1024); };
            vreq(0, 0); vreq(1, 1);
            int rs = 0, trip = 0;
            const at_f4 *const rb = reinterpret_cast<const at_f4 *>(ring) + (lane & 15);
            for (; j + 3 * NW < Tmax; j += 4 * NW, trip++) {
                at_f4 v[4], pr[4];
                vreq(trip + 2, rs == 0 ? 2 : rs - 1);
                at_wait_dma<2>();
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    v[u] = rb[rs * 64 + u * 16];
                    pr[u] = *reinterpret_cast<const at_f4 *>(Sq + (size_t)(j + NW * u) * G);
                }
                rs = rs == 2 ? 0 : rs + 1;
#pragma unroll
                for (int u = 0; u < 4; u++) mac4(v[u], pr[u]);
            }
            at_wait_dma<0>();                                                    // (also: the ring is quiet before the area becomes pvp)
            if (j < Tmax) {                               // the last, partial trip: requested already (slot rs)
                at_f4 v[4], pr[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    v[u] = rb[rs * 64 + u * 16];
                    pr[u] = *reinterpret_cast<const at_f4 *>(Sq + (size_t)min(j + NW * u, t_cap - 1) * G);
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (j + NW * u < Tmax) mac4(v[u], pr[u]);   // (a key past Tmax is not touched: its V row may never have been written)
            }
        } else {
        for (; j + 3 * NW < Tmax; j += 4 * NW) {
            at_f4 v[4], pr[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                v[u] = vrow[(size_t)(j + NW * u) * (DK / 4)];
                pr[u] = *reinterpret_cast<const at_f4 *>(Sq + (size_t)(j + NW * u) * G);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) mac4(v[u], pr[u]);
        }
        if (j < Tmax) {                                   // the last, partial trip
            at_f4 v[4], pr[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int jj = j + NW * u;
                v[u] = vrow[(size_t)min(jj, t_cap - 1) * (DK / 4)];
                pr[u] = *reinterpret_cast<const at_f4 *>(Sq + (size_t)min(jj, t_cap - 1) * G);
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (j + NW * u < Tmax) mac4(v[u], pr[u]);   // (a key past Tmax is not touched: its V row may never have been written)
        }
        }
    }
    ATTN_STAMP(3);
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();                                      // every read of S is done: the area becomes pvp
    {
        const int qg = lane >> 4, dg = lane & 15;
#pragma unroll
        for (int qi = 0; qi < 4; qi++)
#pragma unroll
            for (int c = 0; c < 4; c++) pvp[((size_t)wave * G + 4 * qg + qi) * DK + 4 * dg + c] = acc[qi][c];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; r++) {                         // wave w finishes queries w and w + 8: 64 outputs each = two Q8 blocks
        const int q = wave + NW * r;
        if (i0 + q < N) {
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < NW; s2 += 2) {
                t0 += pvp[((size_t)s2 * G + q) * DK + lane];
                t1 += pvp[((size_t)(s2 + 1) * G + q) * DK + lane];
            }
            store_head_output(p, i0 + q, h, lane, (float)(t0 + t1), p.oq_q != nullptr);
        }
    }
    ATTN_STAMP(4); ATTN_STAMP(6);
}

// (A matrix-core version of this contraction -- v_mfma_f32_16x16x4_f32 for QK^T and PV -- was built in round 1 and removed in
// round 2: f32 MFMA is an f32 fma chain, the reference sums the f32-rounded products in double, and the ~1e-7 difference flips
// int8 codes of the next Q8 activation block often enough that its logits only held 5e-2, 50x outside the contract; it was also
// slower than the VALU kernel at 8-16 query rows.  The matrix cores are used where the reference's arithmetic is integer and
// therefore exact: the int8 block dots of the projections, kernels_mfma.hip.h.  DESIGN.md section 4.5.)

}  // namespace bgk
