// Single-token decode of FLOAT weight files (F32 / F16; biogpt.cpp:160-165 accepts both, README.md:24,45 times them) as ONE persistent launch for all layers.
// biogpt.cpp:691-795 per layer; the arithmetic per element is that of kernels_fdecode.hip.h / attn_fast_kernel (f32 products, double sums, fp16-table exp and GELU).
//
// Why: this is the one decode regime of the model that IS bandwidth-bound (50 MB of F32 weights per layer against 7 MB of Q4_0) -- and on five launches per layer it ran at
// 27.5 % of the HBM peak: 120 launches per token of 4 - 16 MB each are launch boundaries, not streams (5.4 us per launch for 0.5 - 2 us of bytes).  The register-stationary
// XCD pipeline of kernels_xpipe.hip.h does not carry over (a layer is 50 MB, not 7), but its hand-offs do: tagged 8-byte {value, tag} granules, no flags, no fences.
//
// Shape: 256 workgroups (one per compute unit) x 6 waves (4 computing, 2 polling).  Workgroup b owns 1/256 of the rows of EVERY matrix (q/k/v 12, out_proj 4, fc1 16, fc2 4 rows;
// a computing wave a quarter of them) and keeps the layer's whole share -- 196 KB of F32 -- in registers: each matrix's rows are re-requested for the NEXT layer at the top of
// the stage behind their use, so the weight stream of layer l + 1 runs under the dependent stages of layer l and the launch is one continuous stream.  Stages per layer, each
// consuming the previous one's output of ALL workgroups: A LayerNorm + q/k/v rows (+ KV append), B attention (workgroups 0 .. 15: one head each, its old K / V rows brought
// into LDS by DMA a layer ahead), C out_proj + residual, D LayerNorm + fc1 + GELU, E fc2 + residual.  What the stage-border timeline (BIOGPT_HIP_FPIPE_STAMPS,
// tools/fpipe_timeline.py) showed, in the order it was found (per layer, F32 / F16: 33 / 32 us on the first build -> 19 / 16; five launches: 27 / 23):
//   * A wave's vector-memory operations return in order: a poll issued behind a weight request waits for the weights.  So waves 0 .. 3 stream and compute and NEVER poll;
//     waves 4 and 5 poll and never stream: they collect a stage's input granules (1024 - 4096) into LDS -- running the LayerNorm on them where the stage has one: the column is
//     in a wave's lanes anyway -- and every wave meets at ONE s_barrier per stage.
//   * For the same reason the computing waves issue NO other vector load: a bias read or a table look-up would be answered only when every weight request in front of it has
//     come back.  Biases, LayerNorm weights and both table look-ups (GELU: on the producer's side, sixteen values per workgroup; the softmax's exp: inside the attention
//     workgroups, score - max handed over in LDS) are the polling waves'; the biases reach the computing waves through LDS.
//   * ... and hipcc must be able to COUNT: weight pointers read from the layer table are generic (flat loads: every wait becomes vmcnt(0)) unless cast to the global address
//     space; a request behind a branch, or a prologue in another order than the loop's, prices every wait for the path without it; a per-lane choice between two table fields
//     is a vector load of a pointer + vmcnt(0).  All of these drained the stream once per stage.
//   * Everything a polling wave reads cold (LayerNorm weights, biases) is asked for a stage or a layer ahead -- a cold read is 2 - 3 us under the weight stream; in front of a
//     sweep it delays the sweep, behind it it stands in the CU's memory queue behind the weight requests.
//   * 256 waves sweeping 8 - 32 KB each, pass after pass, are megabytes per microsecond next to the weight stream.  A sweep starts when the workgroup's OWN rows of the stage
//     are published (LDS flag from computing wave 0) plus FpParams::lead x 64 clocks -- a store's way to the memory side -- and is then one pass that usually finds everything:
//     1.2 - 1.6 us per hand-over instead of 1.8 - 2.7 for "one sample granule, a round trip, then the sweep" (kept where a workgroup has no rows of its own in the stage: the
//     attention output).  Two passes in flight: slower (the pass left over stands in front of the next stage's sweep).
//   * A stage's buffer is reused by the next layer with the next tag; the full dependency chain (every stage needs every workgroup's output of the one before) makes that safe.
// What bounds it now: five all-to-all hand-overs per layer at 1.2 - 1.6 us (store to the memory side + one read round trip), 3 us of row dots (ONE wave per SIMD issues an
// instruction every ~5 clocks), 1.8 us of LayerNorm on single waves, 3.4 us inside the attention stage -- 16 of 19 us are dependent latency, not bytes (F32: 2.6 TB/s).
// Contexts up to FP_TMAX keys (a head's K / V rows must fit LDS beside the stage inputs); beyond, and for every other shape, the five-launch layer stays.
#pragma once

#include "kernels_xpipe.hip.h"
#include "kernels_fdecode.hip.h"

namespace bgk {

struct FpLayer {
    const float *ln0_w, *ln0_b, *ln1_w, *ln1_b;
    const float *bqkv, *bo, *b1, *b2;
    const uint8_t *Wqkv, *Wo, *W1, *W2;      // row-major float / half rows: [3072][1024], [1024][1024], [4096][1024], [1024][4096]
    float *kcache, *vcache;                  // layer slice, head-major [16][P][64]
};

typedef const FpLayer __attribute__((address_space(4))) FpLayerK;
typedef const unsigned char __attribute__((address_space(1))) *fp_gptr;      // weights / cache rows: GLOBAL loads and stores (a pointer read from the table is generic: flat
                                                                             // operations, behind which hipcc makes every wait a vmcnt(0))

struct FpParams {
    const FpLayer *layers;
    int32_t n_layer;
    xp_u64 *g_x, *g_qkv, *g_att, *g_x1, *g_h;      // hand-off granules: 1024 / 3072 / 1024 / 1024 / 4096
    uint32_t *ctl;                                 // [0] the tag of this launch's layer 0 (moved on by n_layer at its end), [1] error word
    uint32_t *err_host;                            // pinned mirror of the error word
    const DevState *st;                            // n_past
    const float *x_in;                             // [1024] layer 0's input (the embedding launch in front)
    float *x_out;                                  // [1024] the last layer's output (input of the final LayerNorm + lm_head launch)
    float eps, q_scale;
    int32_t P;
    const uint16_t *exp_tab, *gelu_tab;
    int32_t fault;                                 // test hook (BIOGPT_HIP_FPIPE_FAULT): workgroup 255 withholds its out_proj rows of layer 0 -- the launch must drain on its bounded spins
    int32_t lead;                                  // s_sleep units (64 clocks) between "this workgroup's rows of the stage are published" and the first sweep for everybody's
    unsigned long long *stamps;                    // diagnostics (BIOGPT_HIP_FPIPE_STAMPS=1; nullptr otherwise): s_memrealtime (100 MHz) of three workgroups at every stage border, [3][32 layers][32]
};

constexpr int FP_TMAX = 224;

// LDS (bytes)
constexpr int FP_S_X0 = 0;              // [1024] f32 the layer's input (residual of out_proj)
constexpr int FP_S_XN = 4096;           // [1024] LayerNorm 0 of it (F16 files: rounded through fp16, as ggml converts the activation row)
constexpr int FP_S_ATT = 8192;          // [1024] attention output
constexpr int FP_S_X1 = 12288;          // [1024] out_proj's output (residual of fc2)
constexpr int FP_S_X1N = 16384;         // [1024] LayerNorm 1 of it
constexpr int FP_S_H = 20480;           // [4096] GELU(fc1)
constexpr int FP_S_Q = 36864;           // [64] q, [64] new k, [64] new v of the head (attention workgroups)
constexpr int FP_S_S = 37632;           // [256] softmax numerators
constexpr int FP_S_REDF = 38656;        // [16] f32
constexpr int FP_S_REDD = 38720;        // [8] double
constexpr int FP_S_PV = 38784;          // [4][64] double
constexpr int FP_S_BIAS = 40832;        // [2][48] f32 the layer's biases of this workgroup's rows: q/k/v 12, out_proj 4, fc1 16, fc2 4 -- written by the polling wave while the
                                        // computing waves may still be in stage E of the layer before: two buffers, by the layer's parity
constexpr int FP_S_K = 41216;           // [FP_TMAX][16 pieces of 16 bytes], piece c of key k at k * 16 + (c ^ (k & 15)): a wave's lanes (one key each) read one piece index without bank conflicts
constexpr int FP_S_V = FP_S_K + FP_TMAX * 256;      // [FP_TMAX][64] f32
constexpr int FP_S_TOTAL = FP_S_V + FP_TMAX * 256;
__host__ __device__ inline size_t fpipe_smem_bytes() { return (size_t)FP_S_TOTAL; }
// workgroup barrier for LDS hand-overs: this wave's LDS operations are done (lgkmcnt) -- its vector-memory queue is NOT drained (__syncthreads() would wait for the weight
// requests of the next layer); the "memory" clobber keeps the compiler from moving LDS accesses across it
#define FP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ void fp_fail(const FpParams &p, uint32_t code) {
    __hip_atomic_store(p.ctl + 1, code, XP_RLX);
    __hip_atomic_store(p.err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// every lane polls its N granules (stride S) until all of them carry `tag`; wave-uniform exit, bounded; false: the launch is failing (time-out here or the error word set elsewhere)
__device__ __forceinline__ void fp_lead(int n) { for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(1); }

template <int N, int S, int SAMPLE = N - 1>      // SAMPLE: the granule (index, in units of S) polled first; -1: none (a second sweep behind one that had it)
__device__ __forceinline__ bool fp_sweep(const xp_u64 *gb, uint32_t i0, uint32_t tag, float (&v)[N], const FpParams &p) {      // granules gb[i0 + k S]: uniform base + 32-bit index = one offset register per 4 KB of span
    // first ONE granule per lane (a single 512-byte request) with a pause between passes, the full sweep only once that sample carries the tag: 256 polling waves sweeping
    // 8 - 32 KB each, pass after pass, are megabytes per microsecond on the memory side -- next to the weight stream, and in front of this CU's own requests
    for (uint32_t spins = 0; SAMPLE >= 0; spins++) {
        const xp_u64 a = __hip_atomic_load(gb + (i0 + (uint32_t)(SAMPLE * S)), XP_RLX);
        if (__all((uint32_t)(a >> 32) == tag)) break;
        if (spins >= XP_SPIN_MAX) { if ((threadIdx.x & 63) == 0) fp_fail(p, 1u); return false; }
        if ((spins & 1023u) == 1023u && __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    for (uint32_t spins = 0;; spins++) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const xp_u64 a = __hip_atomic_load(gb + (i0 + (uint32_t)(k * S)), XP_RLX);
            v[k] = __uint_as_float((uint32_t)a);
            ok &= (uint32_t)(a >> 32) == tag;
        }
        if (__all(ok)) return true;
        if (spins >= XP_SPIN_MAX) { if ((threadIdx.x & 63) == 0) fp_fail(p, 1u); return false; }
        if ((spins & 1023u) == 1023u && __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) return false;
    }
}

// a row of K elements (this lane's chunks in w) against the activation column in LDS: f32 products, double sums, one wave reduction (fdec_kernel's arithmetic)
template <int WT, int NI>
__device__ __forceinline__ float fp_row_dot(const uint4 (&w)[NI], const float *s_act, int lane) {
    constexpr int EPC = (WT == W_F32) ? 4 : 8;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int i = 0; i < NI; i++) {
        float xr[EPC];
#pragma unroll
        for (int j = 0; j < EPC; j += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(s_act + EPC * (lane + 64 * i) + j);
            xr[j] = t.x; xr[j + 1] = t.y; xr[j + 2] = t.z; xr[j + 3] = t.w;
        }
        fdec_dot16<WT>(w[i], xr, a0, a1);
    }
    return (float)wave_sum_f64(a0 + a1);
}

template <int WT, bool ST>      // ST: the diagnostics build (stage-border stamps)
__global__ __launch_bounds__(384) void fpipe_kernel(const FpParams p) {
    static_assert(WT == W_F32 || WT == W_F16, "float weights");
    constexpr int EPC = (WT == W_F32) ? 4 : 8;
    constexpr int NI1 = 1024 / (64 * EPC), NI4 = 4096 / (64 * EPC);      // 16-byte chunks per lane of a 1024- / 4096-element row
    constexpr int RB1 = 1024 / EPC * 16, RB4 = 4096 / EPC * 16;          // bytes per row
    constexpr bool H16 = WT == W_F16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *const s_x0 = reinterpret_cast<float *>(smem + FP_S_X0);
    float *const s_xn = reinterpret_cast<float *>(smem + FP_S_XN);
    float *const s_att = reinterpret_cast<float *>(smem + FP_S_ATT);
    float *const s_x1 = reinterpret_cast<float *>(smem + FP_S_X1);
    float *const s_x1n = reinterpret_cast<float *>(smem + FP_S_X1N);
    float *const s_h = reinterpret_cast<float *>(smem + FP_S_H);
    float *const s_q = reinterpret_cast<float *>(smem + FP_S_Q);
    float *const s_S = reinterpret_cast<float *>(smem + FP_S_S);
    float *const s_redf = reinterpret_cast<float *>(smem + FP_S_REDF);
    double *const s_redd = reinterpret_cast<double *>(smem + FP_S_REDD);
    double *const s_pv = reinterpret_cast<double *>(smem + FP_S_PV);
    float *const s_bias = reinterpret_cast<float *>(smem + FP_S_BIAS);
    // "this workgroup's rows of stage s of layer L are published" (1 + 4 L + s; s = 0 q/k/v, 1 out_proj, 3 fc2), written by computing wave 0: the polling wave starts its
    // two-deep sweep for the stage's output of ALL workgroups from there (fp_sweep_piped)
    // (LDS address space spelled out: through a generic pointer the flag is a FLAT access, and behind a flat access hipcc makes every wait of the wave a vmcnt(0))
    volatile __attribute__((address_space(3))) uint32_t *const s_flag = (volatile __attribute__((address_space(3))) uint32_t *)((__attribute__((address_space(3))) unsigned char *)smem + (FP_S_REDD + 32));
    auto own_rows_out = [&](uint32_t want) __attribute__((always_inline)) {
        for (uint32_t n = 0; n < (1u << 22) && *s_flag < want; n++) __builtin_amdgcn_s_sleep(1);
    };
    float4 *const s_K = reinterpret_cast<float4 *>(smem + FP_S_K);
    float *const s_V = reinterpret_cast<float *>(smem + FP_S_V);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const uint32_t epoch0 = __hip_atomic_load(p.ctl, XP_RLX);
    const int n_past = p.st->n_past, T = n_past + 1;
    const bool attn_wg = b < 16;
    const int nl = p.n_layer;
    unsigned long long *stp = nullptr;
    if (ST && p.stamps && (b == 0 || b == 128 || b == 255) && (wave == 0 || wave >= 4)) stp = p.stamps + (b == 0 ? 0 : b == 128 ? 1 : 2) * 1024 + (wave >= 4 ? 16 : 0);
#define FP_STAMP(L_, i_) do { if (ST && stp && (L_) < 32) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (lane == 0) stp[(L_) * 32 + (i_)] = t_; } } while (0)

    if (wave >= 4) {
        // ======================= the two polling waves: stage inputs -> LDS (LayerNorm where the stage has one); every wave of the workgroup meets at the same barriers =======================
        // wave 4: the layer's input + LayerNorm 0 + the biases, the attention's inputs and table look-ups, the attention output, GELU + publication of fc1's rows, the first half of
        // GELU(fc1); wave 5: out_proj's output + LayerNorm 1, the second half of GELU(fc1).  (One wave for all of it holds 64 registers of LayerNorm weights a stage ahead and
        // cannot keep two passes over 32 granules per lane in flight beside them.)
        const bool w5 = wave == 5;
        bool alive = true;
        if (!w5 && lane == 0) *s_flag = 0u;
        // wave 5: LayerNorm weights, asked for a stage or a layer ahead in its idle stretch behind stage A's barrier -- a cold read is 2 - 3 us under the weight stream and a wave's
        // requests return in order: in front of a sweep they delay it, behind it they stand in the CU's queue behind the weight requests
        float lw0[16], lb0[16], lw1[16], lb1[16];
        // wave 4: the biases of this workgroup's 36 rows, one per lane, a layer ahead (a scalar load by the computing waves is a cold miss -- 1 us -- at its use, and cannot be asked
        // for earlier: every LDS barrier waits for it)
        auto bias_of = [&](const FpLayerK &Yx) __attribute__((always_inline)) -> float {
            const int l = lane < 36 ? lane : 35;
            const float *q = l < 12 ? Yx.bqkv + b * 12 + l : l < 16 ? Yx.bo + b * 4 + (l - 12) : l < 32 ? Yx.b1 + b * 16 + (l - 16) : Yx.b2 + b * 4 + (l - 32);
            return *(const __attribute__((address_space(1))) float *)q;
        };
        float bias_next = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; k++) { lw0[k] = 0.0f; lb0[k] = 0.0f; lw1[k] = 0.0f; lb1[k] = 0.0f; }
        if (!w5) {
            bias_next = bias_of(((const FpLayerK *)p.layers)[0]);
        } else {
            const FpLayerK &Y0 = ((const FpLayerK *)p.layers)[0];
#pragma unroll
            for (int k = 0; k < 16; k++) { lw0[k] = Y0.ln0_w[lane + 64 * k]; lb0[k] = Y0.ln0_b[lane + 64 * k]; }
        }
        for (int L = 0; L < nl; L++) {
            const uint32_t tag = epoch0 + (uint32_t)L;
            const FpLayerK &Y = ((const FpLayerK *)p.layers)[L];      // (constant address space: the table's pointers arrive by scalar loads)
            int tidp = threadIdx.x;
            asm volatile("" : "+v"(tidp));      // (per layer, as in the computing waves' loop: hoisted granule addresses are spilled at the kernel's 256 registers)
            const int lane = tidp & 63;
            // the column in a wave: element lane + 64 k, k = 0 .. 15 -- ALL of a stage's granules of a lane are requested in one poll pass (a pass is a round trip to the memory side)
            auto layer_norm = [&](const float (&x)[16], const float (&lw)[16], const float (&lb)[16], float *raw, float *out) __attribute__((always_inline)) {
                double s1 = 0.0;
#pragma unroll
                for (int k = 0; k < 16; k += 4) s1 += ((double)x[k] + (double)x[k + 1]) + ((double)x[k + 2] + (double)x[k + 3]);
                s1 = wave_sum_f64(s1);
                const float mean = (float)(s1 * (1.0 / 1024.0));
                float a[16];
                double s2 = 0.0;
#pragma unroll
                for (int k = 0; k < 16; k += 4) {
#pragma unroll
                    for (int j = 0; j < 4; j++) a[k + j] = __fsub_rn(x[k + j], mean);
                    s2 += ((double)__fmul_rn(a[k], a[k]) + (double)__fmul_rn(a[k + 1], a[k + 1])) + ((double)__fmul_rn(a[k + 2], a[k + 2]) + (double)__fmul_rn(a[k + 3], a[k + 3]));
                }
                s2 = wave_sum_f64(s2);
                const float var = (float)(s2 * (1.0 / 1024.0));
                const float scale = 1.0f / sqrtf(__fadd_rn(var, p.eps));
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    float y = __fadd_rn(__fmul_rn(lw[k], __fmul_rn(a[k], scale)), lb[k]);
                    if (H16) y = h2f(f2h(y));
                    raw[lane + 64 * k] = x[k];
                    out[lane + 64 * k] = y;
                }
            };
            // ---- A: the layer's input, LayerNorm 0 (wave 5); this layer's biases -> LDS (wave 4) ----
            if (w5) {
                float x[16];
                if (L == 0) {
#pragma unroll
                    for (int k = 0; k < 16; k++) x[k] = p.x_in[lane + 64 * k];
                } else {
#pragma unroll
                    for (int k = 0; k < 16; k++) x[k] = 0.0f;
                    own_rows_out(1u + 4u * (uint32_t)(L - 1) + 3u);
                    fp_lead(p.lead);
                    if (alive) alive = fp_sweep<16, 64, -1>(p.g_x, (uint32_t)lane, tag - 1u, x, p);
                }
                FP_STAMP(L, 0);
                layer_norm(x, lw0, lb0, s_x0, s_xn);
                FP_STAMP(L, 6);
            } else if (lane < 36) s_bias[(L & 1) * 48 + lane] = bias_next;
            FP_BARRIER();
            const FpLayerK &Yn = ((const FpLayerK *)p.layers)[L + 1 < nl ? L + 1 : L];
            if (w5) {      // behind the barrier, with two idle stages in front of this wave: nothing waits for these reads
#pragma unroll
                for (int k = 0; k < 16; k++) { lw1[k] = Y.ln1_w[lane + 64 * k]; lb1[k] = Y.ln1_b[lane + 64 * k]; lw0[k] = Yn.ln0_w[lane + 64 * k]; lb0[k] = Yn.ln0_b[lane + 64 * k]; }
            }
            // ---- B: the head's q row and the token's new k / v rows (attention workgroups, wave 4) ----
            if (attn_wg && !w5) {
                float v[3] = {0.f, 0.f, 0.f};
                own_rows_out(1u + 4u * (uint32_t)L);
                fp_lead(p.lead);
                if (alive) alive = fp_sweep<3, 1024, -1>(p.g_qkv, (uint32_t)(b * 64 + lane), tag, v, p);
                s_q[lane] = v[0]; s_q[64 + lane] = v[1]; s_q[128 + lane] = v[2];
                FP_STAMP(L, 1);
            }
            FP_BARRIER();
            if (attn_wg) {      // the attention's own seven barriers; between the fourth and the fifth wave 4 looks the softmax numerators up (ggml_soft_max: fp16 exp table)
                FP_BARRIER(); FP_BARRIER(); FP_BARRIER(); FP_BARRIER();
                if (!w5) {
                    float e[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) { const int key = lane + 64 * k; e[k] = key < T ? h2f(p.exp_tab[f2h(s_S[key])]) : 0.0f; }
#pragma unroll
                    for (int k = 0; k < 4; k++) s_S[lane + 64 * k] = e[k];
                }
                FP_BARRIER(); FP_BARRIER(); FP_BARRIER();
            }
            // ---- C: the attention output of all heads (wave 4) ----
            if (!w5) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = 0.0f;
                if (alive) alive = fp_sweep<16, 64>(p.g_att, (uint32_t)lane, tag, v, p);
                FP_STAMP(L, 2);
#pragma unroll
                for (int k = 0; k < 16; k++) s_att[lane + 64 * k] = H16 ? h2f(f2h(v[k])) : v[k];
            }
            FP_BARRIER();
            if (!w5) bias_next = bias_of(Yn);      // (wave 4's idle stretch: its next poll is two stages on)
            // ---- D: out_proj's output, LayerNorm 1 (wave 5) ----
            if (w5) {
                float x[16];
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = 0.0f;
                own_rows_out(1u + 4u * (uint32_t)L + 1u);
                fp_lead(p.lead);
                if (alive) alive = fp_sweep<16, 64, -1>(p.g_x1, (uint32_t)lane, tag, x, p);
                FP_STAMP(L, 3);
                layer_norm(x, lw1, lb1, s_x1, s_x1n);
                FP_STAMP(L, 7);
            }
            FP_BARRIER();
            // fc1's sixteen rows of this workgroup (bias + dot, handed over in LDS): ggml_gelu's fp16 table look-up and the publication are wave 4's -- a look-up by a computing
            // wave would come back behind its weight requests; on the consumers' side it was 2 x 32 look-ups per lane, two table latencies per layer instead of one
            FP_BARRIER();
            if (!w5) {
                if (lane < 16) xp_put(p.g_h + b * 16 + lane, tag, __float_as_uint(h2f(p.gelu_tab[f2h(s_redf[lane])])));
                FP_STAMP(L, 5);
            }
            // ---- E: GELU(fc1), 4096 values = 64 granules per lane: wave 4 the first half, wave 5 the second (no sample granule: the workgroup's own sixteen are being published) ----
            {
                float v[32];
#pragma unroll
                for (int k = 0; k < 32; k++) v[k] = 0.0f;
                const uint32_t h0 = w5 ? 2048u : 0u;
                fp_lead(p.lead);
                if (alive) alive = fp_sweep<32, 64, -1>(p.g_h, h0 + (uint32_t)lane, tag, v, p);
#pragma unroll
                for (int k = 0; k < 32; k++) s_h[h0 + lane + 64 * k] = v[k];
                FP_STAMP(L, w5 ? 8 : 4);
            }
            FP_BARRIER();
        }
        return;
    }

    // ======================= waves 0 .. 3: stream and compute =======================
    uint4 wq[3][NI1], wo[NI1], w1[4][NI1], w2[NI4];
    auto ld = [&](fp_gptr ptr) __attribute__((always_inline)) -> uint4 {
        const fd_u4 t = __builtin_nontemporal_load((const __attribute__((address_space(1))) fd_u4 *)ptr);      // streamed once
        return make_uint4(t.x, t.y, t.z, t.w);
    };
    auto req_qkv = [&](int L) __attribute__((always_inline)) {
        fp_gptr base = (fp_gptr)((const FpLayerK *)p.layers)[L].Wqkv + (size_t)(b * 12 + wave * 3) * RB1 + lane * 16;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int i = 0; i < NI1; i++) wq[r][i] = ld(base + (size_t)r * RB1 + 1024 * i);
    };
    auto req_wo = [&](int L) __attribute__((always_inline)) {
        fp_gptr base = (fp_gptr)((const FpLayerK *)p.layers)[L].Wo + (size_t)(b * 4 + wave) * RB1 + lane * 16;
#pragma unroll
        for (int i = 0; i < NI1; i++) wo[i] = ld(base + 1024 * i);
    };
    auto req_w1 = [&](int L, int r0) __attribute__((always_inline)) {      // rows r0, r0 + 1 of this wave's four
        fp_gptr base = (fp_gptr)((const FpLayerK *)p.layers)[L].W1 + (size_t)(b * 16 + wave * 4) * RB1 + lane * 16;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < NI1; i++) w1[r0 + r][i] = ld(base + (size_t)(r0 + r) * RB1 + 1024 * i);
    };
    auto req_w2 = [&](int L, int i0) __attribute__((always_inline)) {      // chunks i0 .. i0 + NI4 / 2 - 1 of this wave's row
        fp_gptr base = (fp_gptr)((const FpLayerK *)p.layers)[L].W2 + (size_t)(b * 4 + wave) * RB4 + lane * 16;
#pragma unroll
        for (int i = 0; i < NI4 / 2; i++) w2[i0 + i] = ld(base + 1024 * (i0 + i));
    };
    // the head's old K / V rows -> LDS by DMA (asm: hipcc must not count it, or it drains every load in front of every ds_read).  K: 64 pieces of 16 bytes per instruction,
    // LDS piece q = 64 n + lane holds piece (q & 15) ^ (key & 15) of key q >> 4; V: natural order.  Keys past the cache slice are clamped (never used).
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
    auto req_kv = [&](int L) __attribute__((always_inline)) {
        const unsigned char *kb = reinterpret_cast<const unsigned char *>(((const FpLayerK *)p.layers)[L].kcache + (size_t)b * p.P * 64);
        const unsigned char *vb = reinterpret_cast<const unsigned char *>(((const FpLayerK *)p.layers)[L].vcache + (size_t)b * p.P * 64);
        const int ninstr = (T + 3) >> 2;
        for (int n = wave; n < ninstr; n += 4) {
            const int q = 64 * n + lane, key = min(q >> 4, p.P - 1), c = q & 15;
            at_dma16(kb + (size_t)key * 256 + ((c ^ (key & 15)) << 4), lds0 + FP_S_K + 1024 * n);
            at_dma16(vb + (size_t)key * 256 + (c << 4), lds0 + FP_S_V + 1024 * n);
        }
    };
    // Requests: a matrix's rows for the next layer at a stage top behind their use (the whole layer's share, 192 registers of F32, is always held or on its way), spread
    // over the five stage tops -- F32: 8 / 20 / 8 / 4 / 8 requests of 1 KB per wave at A / B / C / D / E.  (fc1's 16 and fc2's 16 each in one piece at E and A: 46 MB for the
    // chip within 7 us, the requests at the next top then took 1.4 - 2.2 us to ISSUE.)
    if (attn_wg) req_kv(0);
    // (in the loop's order, pinned: hipcc prices a wait for the worst path into the loop, and it interleaves unpinned requests as it likes)
    req_qkv(0); asm volatile("" ::: "memory"); req_wo(0); asm volatile("" ::: "memory"); req_w1(0, 0); asm volatile("" ::: "memory");

    for (int L = 0; L < nl; L++) {
        const uint32_t tag = epoch0 + (uint32_t)L;
        const FpLayerK &Y = ((const FpLayerK *)p.layers)[L];      // (constant address space: the table's pointers arrive by scalar loads)
        const bool more = L + 1 < nl;
        const float *const s_bl = s_bias + (L & 1) * 48;
        const int Ln = more ? L + 1 : L;      // the requests are UNCONDITIONAL (the last layer asks for its own rows again, 1/24 of the traffic, nobody waits for it): behind a branch
                                              // hipcc prices every wait for the path WITHOUT the requests -- vmcnt(11) in front of stage A instead of vmcnt(43), a drained queue per stage
        // (the thread index goes through an empty asm in every iteration: without it the attention's sixteen swizzled LDS addresses per thread and a dozen more are hoisted
        //  out of the loop and spilled -- and a scratch reload stands in the wave's vector-memory queue behind the weight requests)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        // ================= A: q / k / v rows (biogpt.cpp:705-727) =================
        FP_BARRIER();
        FP_STAMP(L, 0);
        req_w2(L, 0);      // (fc2's registers were used up in the stage before; the other half follows at stage C)
        FP_STAMP(L, 12);
        {
            float v[3];
#pragma unroll
            for (int r = 0; r < 3; r++) v[r] = fp_row_dot<WT, NI1>(wq[r], s_xn, lane);
            const int rowA = b * 12 + wave * 3;
            float *kc = Y.kcache, *vc = Y.vcache;
            asm volatile("" : "+s"(kc), "+s"(vc));      // both by scalar loads: a per-lane choice of the table's FIELD is a vector load of the pointer and a vmcnt(0) behind it
            if (lane < 3) {
                const int row = rowA + lane;
                float o = __fadd_rn(s_bl[wave * 3 + lane], lane == 0 ? v[0] : lane == 1 ? v[1] : v[2]);
                const int which = row >> 10, rr = row & 1023;
                if (which == 0) o = __fmul_rn(o, p.q_scale);                 // Q scaled AFTER the bias (biogpt.cpp:708-710)
                else ((__attribute__((address_space(1))) float *)((which == 1) ? kc : vc))[((size_t)(rr >> 6) * p.P + n_past) * 64 + (rr & 63)] = o;      // KV append (biogpt.cpp:721-727): for later launches
                xp_put(p.g_qkv + row, tag, __float_as_uint(o));
            }
            if (wave == 0 && lane == 0) *s_flag = 1u + 4u * (uint32_t)L;
            FP_STAMP(L, 1);
        }
        // ================= B: attention of head b (biogpt.cpp:729-764) =================
        FP_BARRIER();
        FP_STAMP(L, 2);
        req_qkv(Ln);
        req_w1(L, 2);      // (the second half of THIS layer's fc1 rows: two stages ahead of their use; at stage C's top, one stage ahead: 4 % slower)
        FP_STAMP(L, 10);
        if (attn_wg) {
            // this wave's share of the head's old rows (asked for behind out_proj's dot of the layer before) has landed: every request since -- out_proj, half of fc1, half of
            // fc2 of this layer, q/k/v of the next, the other half of fc1: 8 NI1 + NI4 / 2 -- may still be on its way (the handful of stores in between only make the wait a
            // little stricter)
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * NI1 + NI4 / 2) : "memory");
            FP_BARRIER();                          // ... and everyone's
            if (tid < 16) s_K[n_past * 16 + (tid ^ (n_past & 15))] = *reinterpret_cast<const float4 *>(s_q + 64 + 4 * tid);
            else if (tid < 32) *reinterpret_cast<float4 *>(s_V + n_past * 64 + 4 * (tid - 16)) = *reinterpret_cast<const float4 *>(s_q + 128 + 4 * (tid - 16));
            FP_BARRIER();
            FP_STAMP(L, 13);
            float sc = -INFINITY;
            if (tid < T) {
                double acc = 0.0;
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    const float4 k4 = s_K[tid * 16 + (c ^ (tid & 15))], q4 = *reinterpret_cast<const float4 *>(s_q + 4 * c);
                    acc += (double)__fmul_rn(k4.x, q4.x); acc += (double)__fmul_rn(k4.y, q4.y); acc += (double)__fmul_rn(k4.z, q4.z); acc += (double)__fmul_rn(k4.w, q4.w);
                }
                sc = (float)acc;
            }
            float mx = wave_max_f32(sc);
            if (lane == 0) s_redf[wave] = mx;
            FP_BARRIER();
            mx = fmaxf(fmaxf(s_redf[0], s_redf[1]), fmaxf(s_redf[2], s_redf[3]));
            s_S[tid] = __fsub_rn(sc, mx);      // the table look-up is the polling wave's (no vector load of this wave may stand behind its weight requests)
            FP_BARRIER();
            FP_BARRIER();
            FP_STAMP(L, 14);
            const float e = s_S[tid];
            double sum = wave_sum_f64((double)e);
            if (lane == 0) s_redd[wave] = sum;
            FP_BARRIER();
            sum = (s_redd[0] + s_redd[1]) + (s_redd[2] + s_redd[3]);      // (fp16 values below 2^11: exact in any order)
            const float inv = inv_sum_f32(sum);
            {
                // wave w: keys w, w + 4, w + 8, ... -- alternately into a0 / a1; eight keys per trip, their LDS reads in flight together (a trip per key pair with a branch in it
                // was 1.4 us of the stage at 100 keys)
                double a0 = 0.0, a1 = 0.0;
                for (int j0 = wave; j0 < T; j0 += 32) {
                    float vv[8], ss[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int j = min(j0 + 4 * i, T - 1);
                        vv[i] = s_V[j * 64 + lane]; ss[i] = s_S[j];
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const double t = (double)__fmul_rn(vv[i], __fmul_rn(ss[i], inv));
                        const bool live = j0 + 4 * i < T;
                        if (i & 1) a1 = live ? a1 + t : a1; else a0 = live ? a0 + t : a0;
                    }
                }
                s_pv[wave * 64 + lane] = a0 + a1;
            }
            FP_BARRIER();
            FP_STAMP(L, 15);
            if (wave == 0) {
                const double t0 = s_pv[lane] + s_pv[128 + lane], t1 = s_pv[64 + lane] + s_pv[192 + lane];
                xp_put(p.g_att + b * 64 + lane, tag, __float_as_uint((float)(t0 + t1)));
            }
        }
        // ================= C: out_proj + bias + residual (biogpt.cpp:767-772) =================
        FP_BARRIER();
        FP_STAMP(L, 3);
        req_w2(L, NI4 / 2);
        {
            const float v = fp_row_dot<WT, NI1>(wo, s_att, lane);
            FP_STAMP(L, 4);
            if (lane == 0) {
                const int row = b * 4 + wave;
                if (!(p.fault && b == 255 && L == 0)) xp_put(p.g_x1 + row, tag, __float_as_uint(__fadd_rn(__fadd_rn(v, s_bl[12 + wave]), s_x0[row])));
                if (wave == 0) *s_flag = 1u + 4u * (uint32_t)L + 1u;
            }
            // the head's rows of the next layer (the LDS they land in is free once the attention stage is over).  (At the top of fc1's stage instead -- further from this
            // workgroup's next poll: measured 3 - 4 % slower, fc1's and fc2's stages each 0.6 - 0.9 us longer)
            if (attn_wg && more) req_kv(L + 1);
        }
        // ================= D: fc1 + bias + GELU (biogpt.cpp:777-787) =================
        FP_BARRIER();
        FP_STAMP(L, 5);
        req_wo(Ln);
        FP_STAMP(L, 9);
        {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = fp_row_dot<WT, NI1>(w1[r], s_x1n, lane);
            const int rowD = b * 16 + wave * 4;
            if (lane < 4) s_redf[wave * 4 + lane] = __fadd_rn(s_bl[16 + wave * 4 + lane], lane == 0 ? v[0] : lane == 1 ? v[1] : lane == 2 ? v[2] : v[3]);
            FP_STAMP(L, 6);
        }
        FP_BARRIER();      // bias + dot of the sixteen rows -> the polling wave (GELU look-up, publication)
        // ================= E: fc2 + bias + residual (biogpt.cpp:790-795) =================
        FP_BARRIER();
        FP_STAMP(L, 7);
        req_w1(Ln, 0);
        FP_STAMP(L, 11);
        {
            const float v = fp_row_dot<WT, NI4>(w2, s_h, lane);
            FP_STAMP(L, 8);
            if (lane == 0) {
                const int row = b * 4 + wave;
                const float o = __fadd_rn(__fadd_rn(v, s_bl[32 + wave]), s_x1[row]);
                if (more) xp_put(p.g_x + row, tag, __float_as_uint(o));
                else p.x_out[row] = o;
                if (wave == 0) *s_flag = 1u + 4u * (uint32_t)L + 3u;
            }
        }
    }
    // the next launch's tags: every workgroup read ctl[0] before it could contribute to ANY stage, and this workgroup's last row needed all of them
    if (b == 0 && tid == 0) {
        if (epoch0 + (uint32_t)nl > 0xF0000000u) fp_fail(p, 5u);
        __hip_atomic_store(p.ctl, epoch0 + (uint32_t)nl, XP_RLX);
    }
}

}  // namespace bgk
