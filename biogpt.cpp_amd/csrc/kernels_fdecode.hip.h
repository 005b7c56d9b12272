// Single-token decode mat-vecs for FLOAT weight files (F32 / F16, biogpt.cpp:160-165 accepts both) at BioGPT-base shapes -- the one
// decode regime of this model that IS bandwidth-bound: 50 MB (F32) of weights per layer against 7 MB for Q4_0.
//
// The generic kernel (kernels.hip.h matvec_kernel) gives every wave ONE row: 1024 workgroups of 16 KB each come and go, every one of
// them repeats the LayerNorm prologue, and a compute unit never has more than a few rows in flight: 8.4 us for fc1's 16 MB
// (2.0 TB/s, rocprofv3).  Here the whole matrix is requested at t = 0: 256 workgroups (one per compute unit) x 4 waves, wave w of
// workgroup b owns R = M / 1024 consecutive rows and issues ALL of their 16-byte loads (up to 16 per lane, 64 KB per workgroup)
// before anything else; the activation column is prepared meanwhile (LayerNorm once per workgroup by its 4 waves, or the plain column
// straight into registers), then per row: f32 products, double accumulation (ggml_vec_dot_f32 / _f16: `sumf += (ggml_float)(x*y)`),
// one DPP wave reduction, epilogue.  The double sums are order-insensitive at f32 output precision (as in the generic kernel, whose
// results these equal bit for bit in every test); F16 weights see the activation rounded through fp16 (ggml_fp32_to_fp16_row).
//
// Measured and NOT kept: 256 extra workgroups per launch that touch the NEXT mat-vec's weights (same 1/256 partition, same XCD) so that its
// ramp overlaps this launch's tail: F32 569 -> 714 us per token -- the extra stream competes with the launch's own, and the lines are gone (or in
// another XCD's L2) when the next launch asks for them.
#pragma once

#include "kernels_decode.hip.h"

namespace bgk {

typedef unsigned int fd_u4 __attribute__((ext_vector_type(4)));

template <int WT> __device__ __forceinline__ void fdec_dot16(const uint4 w, const float *xr, double &a0, double &a1);
// one 16-byte weight chunk against its activation values: F32 = 4 elements, F16 = 8
template <> __device__ __forceinline__ void fdec_dot16<W_F32>(const uint4 w, const float *xr, double &a0, double &a1) {
    a0 += (double)__fmul_rn(__uint_as_float(w.x), xr[0]); a1 += (double)__fmul_rn(__uint_as_float(w.y), xr[1]);
    a0 += (double)__fmul_rn(__uint_as_float(w.z), xr[2]); a1 += (double)__fmul_rn(__uint_as_float(w.w), xr[3]);
}
template <> __device__ __forceinline__ void fdec_dot16<W_F16>(const uint4 w, const float *xr, double &a0, double &a1) {
    const uint32_t p[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a0 += (double)__fmul_rn(h2f((uint16_t)(p[i] & 0xFFFFu)), xr[2 * i]);
        a1 += (double)__fmul_rn(h2f((uint16_t)(p[i] >> 16)), xr[2 * i + 1]);
    }
}

// K = row length (1024 or 4096), R = rows per wave; grid = M / (4 R) workgroups of 256 threads
template <int WT, int PRO, int EPI, int K, int R>
__global__ __launch_bounds__(256) void fdec_kernel(const MatvecParams p) {
    static_assert(WT == W_F32 || WT == W_F16, "float weights");
    static_assert(PRO == PRO_LN || PRO == PRO_PLAIN, "prologue");
    static_assert(PRO != PRO_LN || K == 1024, "LayerNorm prologue: d_model columns");
    constexpr int EPC = (WT == W_F32) ? 4 : 8;             // elements per 16-byte chunk
    constexpr int NI = K / (64 * EPC);                     // chunks per lane per row
    __shared__ __attribute__((aligned(16))) float s_x[PRO == PRO_LN ? 1024 : 4];
    __shared__ double s_red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * R;
    // ---- t = 0: every weight byte of this wave's rows ----
    uint4 w[R][NI];
    {
        const uint4 *base = reinterpret_cast<const uint4 *>(p.W.qs) + (size_t)row0 * (K / EPC) + lane;
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int i = 0; i < NI; i++) {      // streamed once: non-temporal
                const fd_u4 t = __builtin_nontemporal_load(reinterpret_cast<const fd_u4 *>(base + (size_t)r * (K / EPC) + 64 * i));
                w[r][i] = make_uint4(t.x, t.y, t.z, t.w);
            }
    }
    // epilogue operands of the rows this lane finishes (lane r < R finishes row row0 + r)
    float e_bias = 0.0f, e_res = 0.0f;
    int e_npast = 0;
    if (lane < R) {
        e_bias = p.bias[row0 + lane];
        if (EPI == EPI_RESID) e_res = p.resid[row0 + lane];
        if (EPI == EPI_QKV) e_npast = p.st->n_past;
    }
    // ---- the activation column: xr[i][j] = element EPC (lane + 64 i) + j ----
    float xr[NI][EPC];
    if constexpr (PRO == PRO_LN) {
        // ggml_norm + affine (biogpt.cpp:691-701), double statistics: thread t holds elements 4t .. 4t+3 (the arithmetic of ln4_q8_1024 without the Q8 step)
        const float4 v = reinterpret_cast<const float4 *>(p.x)[tid];
        const float4 lw = reinterpret_cast<const float4 *>(p.ln_w)[tid], lb = reinterpret_cast<const float4 *>(p.ln_b)[tid];
        const double s1 = wave_sum_f64(((double)v.x + (double)v.y) + ((double)v.z + (double)v.w));
        if (lane == 0) s_red[wave] = s1;
        __syncthreads();
        const float mean = (float)(((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * (1.0 / 1024.0));
        float a = __fsub_rn(v.x, mean), b = __fsub_rn(v.y, mean), c = __fsub_rn(v.z, mean), d = __fsub_rn(v.w, mean);
        const double s2 = wave_sum_f64(((double)__fmul_rn(a, a) + (double)__fmul_rn(b, b)) + ((double)__fmul_rn(c, c) + (double)__fmul_rn(d, d)));
        if (lane == 0) s_red[4 + wave] = s2;
        __syncthreads();
        const float var = (float)(((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])) * (1.0 / 1024.0));
        const float scale = 1.0f / sqrtf(__fadd_rn(var, p.eps));
        a = __fadd_rn(__fmul_rn(lw.x, __fmul_rn(a, scale)), lb.x);
        b = __fadd_rn(__fmul_rn(lw.y, __fmul_rn(b, scale)), lb.y);
        c = __fadd_rn(__fmul_rn(lw.z, __fmul_rn(c, scale)), lb.z);
        d = __fadd_rn(__fmul_rn(lw.w, __fmul_rn(d, scale)), lb.w);
        if (WT == W_F16) { a = h2f(f2h(a)); b = h2f(f2h(b)); c = h2f(f2h(c)); d = h2f(f2h(d)); }
        reinterpret_cast<float4 *>(s_x)[tid] = make_float4(a, b, c, d);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < EPC; j += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(s_x + EPC * (lane + 64 * i) + j);
                xr[i][j] = t.x; xr[i][j + 1] = t.y; xr[i][j + 2] = t.z; xr[i][j + 3] = t.w;
            }
    } else {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < EPC; j += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(p.x + EPC * (lane + 64 * i) + j);
                xr[i][j] = t.x; xr[i][j + 1] = t.y; xr[i][j + 2] = t.z; xr[i][j + 3] = t.w;
            }
        if (WT == W_F16) {
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < EPC; j++) xr[i][j] = h2f(f2h(xr[i][j]));
        }
    }
    // ---- rows: products in f32, sums in double ----
    float mine = 0.0f;
#pragma unroll
    for (int r = 0; r < R; r++) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < NI; i++) fdec_dot16<WT>(w[r][i], xr[i], a0, a1);
        const float v = (float)wave_sum_f64(a0 + a1);
        if (lane == r) mine = v;
    }
    if (lane < R) {
        const int r = row0 + lane;
        float v = mine;
        if (EPI == EPI_QKV) {       // biogpt.cpp:705-727: bias, Q scaled after the bias, K / V appended to the head-major cache
            v = __fadd_rn(e_bias, v);
            const int which = r / p.D, rr = r - which * p.D;
            if (which == 0) {
                p.q_out[rr] = __fmul_rn(v, p.q_scale);
            } else {
                float *cache = (which == 1) ? p.kcache : p.vcache;
                const int hh = rr / p.dk, dd = rr - hh * p.dk;
                cache[((size_t)hh * p.P + e_npast) * p.dk + dd] = v;
            }
        } else if (EPI == EPI_RESID) {      // biogpt.cpp:767-772, :790-795
            p.out[r] = __fadd_rn(__fadd_rn(v, e_bias), e_res);
        } else {                            // EPI_GELU: biogpt.cpp:777-787, ggml_gelu's fp16 table
            p.out[r] = h2f(p.gelu_tab[f2h(__fadd_rn(e_bias, v))]);
        }
    }
}

}  // namespace bgk
