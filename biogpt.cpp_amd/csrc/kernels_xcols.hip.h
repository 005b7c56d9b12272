// A reference prompt chunk -- biogpt_eval with N = 2 .. 8 tokens (main.cpp:129-137, n_batch = 8) -- as ONE persistent launch: ONE COLUMN PER XCD.
// biogpt.cpp:664-795 for all layers; the arithmetic per element is that of kernels_xpipe.hip.h / kernels_decode.hip.h.
//
// Why: through the launch chain of kernels_fast.hip.h such an eval costs 7 launches per layer of 4 .. 10 us each (0.86 ms per eval at 24 layers, 9.3 k prompt
// tok/s through the drop-in API: profiles/rocprofv3_kernel_stats_r3_per_eval_chunks.csv).  The single-token pipeline of kernels_xpipe.hip.h cannot take the
// chunk's columns one behind the other: inside an eval there is no mask (F1), so column i attends to the K / V rows of columns i + 1 .. N - 1 of the SAME layer,
// and every layer needs all columns of the layer before.  What the chip offers instead: 8 XCDs and <= 8 columns.  Column c lives on XCD c for the whole pass:
//   * every hand-off between the five stages of a layer stays inside the XCD's L2 (plain stores, 0.4 us: the hand-offs of kernels_xpipe.hip.h before the layers
//     were split over XCD pairs), the layer's output too;
//   * the ONLY exchange between columns is the layer's N new K / V rows: workgroup 16 + h publishes head h's rows write-through, the attention workgroups of the
//     other XCDs poll them at the memory side -- one cross-XCD hop per layer, which is also the only synchronisation between the columns;
//   * every XCD streams ALL weights (7 MB per layer through its own L2: the eight XCDs read the same lines within microseconds of each other, the Infinity
//     Cache serves seven of the eight reads).  Units stay packed (5 .. 6 registers per unit, 30 units per lane in the q / k / v workgroups; Q8_0: 9) and are
//     requested in ONE burst per layer and role, in front of the one poll that waits for microseconds anyway (see xc_run: a wave's loads return in order, a poll
//     behind a weight request waits for the fabric) -- 16.7 us per layer against 9.8 us of pure weight stream (tools/microbench20.hip) and ~ 11 us of chain.
//
// Shape: grid = 256 workgroups x 512 threads; XCD and rank inside the XCD as in kernels_xpipe.hip.h (HW_REG_XCC_ID + per-XCD ticket, same control words, same
// hand-off tag counter: a chunk launch is one more launch of the context's pipeline and holds the device's pipeline slot like one).  XCDs >= N take their
// tickets and leave.  Workgroups 16 .. 31 of an XCD: LayerNorm -> Q8 -> the 192 q / k / v rows of head slot - 16 (+ KV append at row n_past + column);
// workgroups 0 .. 15: attention of head `slot` over the n_past old keys (registers, requested a layer ahead) and the N new ones (LDS); all 32: out_proj rows,
// LayerNorm + fc1 + GELU, fc2 rows.  The final LayerNorm + lm_head of the LAST column (F8) is the ordinary stand-alone launch behind this one.
// Contexts up to 512 keys (n_past + N <= 512; beyond 256 the second half of a head's old rows is requested inside the attention stage: xc_run, SEG2), all five block formats (Q8_0 with its q / k / v units requested late: 9 registers per unit).
#pragma once

#include "kernels_xpipe.hip.h"

namespace bgk {

struct XcParams {
    const XpLayer *layers;
    int32_t n_layer;
    xp_u64 *gran;              // [8 columns][n_layer][XP_G_LAYER], zeroed once at allocation
    uint32_t *ctl;             // the pipeline's control words (XpParams::ctl)
    uint32_t *err_host;
    const DevState *st;        // n_past and the chunk's tokens
    DevMatrix tok_emb, pos_emb;
    float embed_scale;
    int32_t n_positions, n_vocab;
    float eps, q_scale;
    int32_t P, t_cap;
    const uint16_t *exp_tab, *gelu_tab;
    int32_t gelu_p, gelu_n, gelu_z;
    int32_t n_cols;            // 2 .. 8
    // streams mode (seq != null; biogpt_hip_generate_greedy_batch with 2 .. 8 sequences): column c is the next token of sequence c -- its own position and token
    // (SeqState), its own K / V cache (kroot / vroot + c * seq_stride), attention over its own n_past + 1 keys only: no exchange between the XCDs at all
    const SeqState *seq;
    float *kroot, *vroot;
    int64_t seq_stride;        // floats between two sequences' caches ([n_layer][P][1024] each)
    float *x_out;              // [n_cols][1024] the last layer's output (input of the final LayerNorm + lm_head launch)
    unsigned long long *wall;  // profiling (BIOGPT_HIP_PROFILE_HOOKS): [n_layer][16] wall clock of workgroups 0 and 16 of column 0
};

#ifdef BIOGPT_HIP_PROFILE_HOOKS
#define XC_WALL(k) do { if (p.wall && tid == 0 && col == 0 && (slot & 15) == 0) p.wall[L * 16 + (k)] = wall_clock64(); } while (0)
#else
#define XC_WALL(k) do {} while (0)
#endif

__device__ __forceinline__ void xc_fail(const XcParams &p, uint32_t code) {
    __hip_atomic_store(p.ctl + 1, code, XP_RLX);
    __hip_atomic_store(p.err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// every ACTIVE lane polls its N granules (stride S) until all their tags carry this launch's counter; wave-uniform exit, bounded
template <int N, int S>
__device__ __forceinline__ void xc_sweep(const xp_u64 *g, bool active, uint32_t epoch, uint32_t (&v)[N], const XcParams &p) {
    for (uint32_t spins = 0;; spins++) {
        bool ok = true;
        if (active) {
#pragma unroll
            for (int k = 0; k < N; k++) {
                const xp_u64 a = __hip_atomic_load((xp_gq)g + k * S, XP_RLX);
                v[k] = (uint32_t)a;
                ok &= (uint32_t)(a >> 32) == epoch;
            }
        }
        if (__all(ok)) return;
        if (spins >= XP_SPIN_MAX) { if ((threadIdx.x & 63) == 0) xc_fail(p, 1u); return; }
        if ((spins & 1023u) == 1023u && __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) return;
    }
}

template <int WT>
__device__ __forceinline__ void xc_load_unit(Unit<WT> &u, const DevMatrix &W, int64_t idx) {
    load_unit<WT>(u, W, idx);
}
// (Measured and gone, round 4 -- table in HISTORY.md, "prompt chunk as one persistent launch": one polling wave per workgroup with the others streaming freely (0.73 against
// 0.50 ms per 8-token eval), waves 4 .. 7 re-requesting their units right behind the use (20 % slower), the streaming hint on the weight units (no change).  Waves 0 .. 3
// poll, every wave keeps the burst discipline described in xc_run.)
template <int WT, int LPK, int KCAP, int ROLE>
__device__ __forceinline__ void xc_run(const XcParams &p, unsigned char *smem, const int col, const int slot, const uint32_t epoch) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant, "block-quantized weights");
    // Q8_0 units are 9 registers: the q / k / v workgroups' 30 units do not fit at once -- their 12 q / k / v units of the NEXT layer are requested at the end of the layer
    // (when out_proj / fc1 / fc2 units are dead) instead of in the burst; that request sits in front of the next layer's input poll (exposed: ~ 3 us per layer)
    constexpr bool QKV_LATE = WT == W_Q8_0;
    static_assert(ROLE == 0 || ROLE == 1, "0 attention head, 1 q/k/v rows");
    static_assert(LPK == 1 || LPK == 2 || LPK == 4 || LPK == 8, "lanes per key");
    constexpr bool ATTN = ROLE == 0;
    constexpr int NW = 8, NT = 512, DK = 64;
    constexpr int QS = 96 / NW, OS = 16 / NW, FS = 64 / NW, F2R = 32 / NW;
    // 512-key variant (SEG2; LPK = 2): the lane pair of key k also takes key k + 256.  Keys 0 .. 255 wait in registers as in the 256-key variant (requested a layer
    // ahead); the rows of keys 256 .. 511 are requested at the START of the stage, in front of the poll for the layer's new rows (which waits for the other workgroups' stage A
    // anyway): 64 registers for the length of the stage instead of 128 more held through the layer (the one-lane-per-key form of round 4: slower than the launch chain).
    // The chain of kernels_fast.hip.h costs 0.92 ms per 8-token eval there.
    constexpr bool SEG2 = KCAP > 256;
    constexpr int KREG = SEG2 ? 256 : KCAP;      // keys whose rows are register-resident
    static_assert(KCAP % NW == 0 && KREG <= NW * 64 / LPK && (!SEG2 || (LPK == 2 && KCAP == 512)), "key capacity of the launch");
    constexpr int NF4 = 16 / LPK, NV = KREG / NW;
    // 256-key variant (Q8_0, 9 registers per unit: beyond 64 keys), attention workgroups: the head's K / V rows + 18 units + the dots' temporaries do not fit; there the fc1 /
    // fc2 units are requested when the attention is done (one / two stages ahead of their use: the poll in between waits for them) and never wait beside the K / V rows.
    // UNCOND (Q8_0): the end-of-layer requests are unconditional (the last layer asks for its own units once more): a request under `if (more)` keeps the OLD registers alive
    // through the whole layer -- 151 spilled VGPRs in the q / k / v workgroups.  For the nibble formats it is the other way round: measured, everything unconditional and no
    // late requests at all (161 - 252 VGPRs, no spills either) is 7 % slower (0.540 against 0.505 ms per 8-token eval): bigger bursts in front of the long poll.
    constexpr bool UNCOND = WT == W_Q8_0 || KCAP > 256;      // (512-key variant: 128 K / V registers must not stay alive through the layer)
    constexpr bool LATE_W2 = ATTN && (KCAP > 128 || (WT == W_Q8_0 && KCAP > 64));
    constexpr int PW = 4;      // polling waves: waves 0 .. 3 (each LayerNorm worker takes its own 4 elements in)
    float *const s_x = reinterpret_cast<float *>(smem + XP_S_X);
    float *const s_x1 = reinterpret_cast<float *>(smem + XP_S_X1);
    uint32_t *const s_xq = reinterpret_cast<uint32_t *>(smem + XP_S_XQ);
    float *const s_xd = reinterpret_cast<float *>(smem + XP_S_XD);
    uint32_t *const s_xs = reinterpret_cast<uint32_t *>(smem + XP_S_XS);
    double *const s_red = reinterpret_cast<double *>(smem + XP_S_RED);
    uint32_t *const s_hq = reinterpret_cast<uint32_t *>(smem + XP_S_HQ);
    float *const s_hd = reinterpret_cast<float *>(smem + XP_S_HD);
    uint32_t *const s_hs = reinterpret_cast<uint32_t *>(smem + XP_S_HS);
    float *const s_part = reinterpret_cast<float *>(smem + XP_S_PART);
    float *const s_new = s_part;        // [8][128] the chunk's new k (0 .. 63) / v (64 .. 127) rows of this head: only the attention stage reads them, no dot stage runs then
    float *const s_g = reinterpret_cast<float *>(smem + XP_S_G);
    float *const s_ln = reinterpret_cast<float *>(smem + XP_S_LN);
    float *const s_bias = reinterpret_cast<float *>(smem + XP_S_BIAS);
    float *const s_cur = reinterpret_cast<float *>(smem + XP_S_CUR);
    // [KCAP] softmax numerators; the 512-key variant's do not fit XP_S_S (256 floats): behind s_new in the block-term region, which no dot stage uses during the attention
    float *const s_S = KCAP > 256 ? reinterpret_cast<float *>(smem + XP_S_PART) + 1024 : reinterpret_cast<float *>(smem + XP_S_S);
    float *const s_redf = reinterpret_cast<float *>(smem + XP_S_REDF);
    double *const s_redd = reinterpret_cast<double *>(smem + XP_S_REDD);
    double *const s_pv = reinterpret_cast<double *>(smem + XP_S_PV);
    const uint16_t *const s_gelu = reinterpret_cast<const uint16_t *>(smem + XP_S_TOTAL);
    const bool streams = p.seq != nullptr;
    // chunk: keys before this eval, keys of the eval (F1: every column sees all N new ones), this column's row; streams: the sequence's own position, one new key
    const int n_old = streams ? p.seq[col].n_past : p.st->n_past;
    const int n_new = streams ? 1 : p.n_cols, c_first = streams ? col : 0;      // whose new K / V rows this column attends to: columns c_first .. c_first + n_new - 1
    const int T = n_old + n_new, pos = streams ? n_old : n_old + col;
    const size_t kv_off = streams ? (size_t)col * (size_t)p.seq_stride : 0;
    const int t_cap = p.t_cap;
    const int head = slot & 15;
    xp_u64 *const Gcol = p.gran + (size_t)col * p.n_layer * XP_G_LAYER;

    // ---- what waits in registers for its stage: the layer's weight units (packed) and, in the attention workgroups, the head's old keys / values ----
    Unit<WT> wqkv[QS], wo[OS], w1[FS], w2[F2R][2];
    float4 kr[NF4];
    float vr[NV];
    float4 l0 = make_float4(0.f, 0.f, 0.f, 0.f), l1 = l0, l2 = l0, l3 = l0;      // LayerNorm weights of the NEXT layer (workers), its bias entry
    float bnext = 0.0f;
    auto request_small = [&](int L, int tid) __attribute__((always_inline)) {
        const XpLayerK &Y = XPL(L);
        if (tid < 256) {
            l0 = xp_ldg4(Y.ln0_w, tid); l1 = xp_ldg4(Y.ln0_b, tid);
            l2 = xp_ldg4(Y.ln1_w, tid); l3 = xp_ldg4(Y.ln1_b, tid);
        }
        if (tid < 192) bnext = ((xp_gf)Y.bqkv)[(tid >> 6) * 1024 + head * 64 + (tid & 63)];
        else if (tid < 224) bnext = ((xp_gf)Y.bo)[slot * 32 + tid - 192];
        else if (tid < 352) bnext = ((xp_gf)Y.b1)[slot * 128 + tid - 224];
        else if (tid < 384) bnext = ((xp_gf)Y.b2)[slot * 32 + tid - 352];
    };
    auto request_qkv = [&](int L, int tid) __attribute__((always_inline)) {
        const int lane = tid & 63, wave = tid >> 6, sub = lane & 31, rsub = lane >> 5;
#pragma unroll
        for (int s = 0; s < QS; s++) {
            const int jj = s * 2 * NW + wave * 2 + rsub;
            xc_load_unit<WT>(wqkv[s], XPL_MATRIX(XPL(L).Wqkv), (int64_t)((jj >> 6) * 1024 + head * 64 + (jj & 63)) * 32 + sub);
        }
    };
    auto request_wo = [&](int L, int tid) __attribute__((always_inline)) {
        const int lane = tid & 63, wave = tid >> 6, sub = lane & 31, rsub = lane >> 5;
#pragma unroll
        for (int s = 0; s < OS; s++) xc_load_unit<WT>(wo[s], XPL_MATRIX(XPL(L).Wo), (int64_t)(slot * 32 + s * 2 * NW + wave * 2 + rsub) * 32 + sub);
    };
    auto request_w1 = [&](int L, int tid) __attribute__((always_inline)) {
        const int lane = tid & 63, wave = tid >> 6, sub = lane & 31, rsub = lane >> 5;
#pragma unroll
        for (int s = 0; s < FS; s++) xc_load_unit<WT>(w1[s], XPL_MATRIX(XPL(L).W1), (int64_t)(slot * 128 + s * 2 * NW + wave * 2 + rsub) * 32 + sub);
    };
    auto request_w2 = [&](int L, int tid) __attribute__((always_inline)) {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int r = 0; r < F2R; r++)
#pragma unroll
            for (int it = 0; it < 2; it++) xc_load_unit<WT>(w2[r][it], XPL_MATRIX(XPL(L).W2), (int64_t)(slot * 32 + wave * F2R + r) * 128 + lane + 64 * it);
    };
    // rows written by EARLIER evals (other launches); rows >= n_old are replaced from LDS before they are used.  Raw buffer loads: ONE offset register per lane for
    // the 8 + 32 loads of the 256-key variant (the per-load part sits in the scalar offset; 64-bit addresses per load cost ~ 30 VGPRs here), rows beyond the cache
    // slice read as 0 (range-checked), streaming hint (nt)
    auto request_kv = [&](int L, int tid) __attribute__((always_inline)) {
        const int ksub = tid & (LPK - 1), kidx = tid / LPK, dd = tid & (DK - 1), sl = tid >> 6;
        const float *kb = (streams ? p.kroot + kv_off + (size_t)L * p.P * 1024 : XPL(L).kcache) + (size_t)head * p.P * DK;
        const float *vb = (streams ? p.vroot + kv_off + (size_t)L * p.P * 1024 : XPL(L).vcache) + (size_t)head * p.P * DK;
        const __amdgpu_buffer_rsrc_t krs = xp_kv_rsrc(kb, p.P * DK * 4), vrs = xp_kv_rsrc(vb, p.P * DK * 4);
        constexpr int CPOL_NT = 2;
        const int ko = (kidx * DK + 4 * ksub) * 4, vo = (sl * DK + dd) * 4;
#pragma unroll
        for (int m = 0; m < NF4; m++) {
            const xp_v4u t = __builtin_amdgcn_raw_buffer_load_b128(krs, ko, LPK * m * 16, CPOL_NT);
            kr[m] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
        }
#pragma unroll
        for (int k = 0; k < NV; k++) vr[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vrs, vo, NW * k * DK * 4, CPOL_NT));
    };
    // layer 0's input (biogpt.cpp:664-686) is requested FIRST: loads return in order, so behind the weight requests below the embedding -- and with it layer 0's
    // LayerNorm -- would wait for them (layer 0 took 24 us against 16.7 for the others)
    const bool emb_fast = p.tok_emb.type == WT && p.pos_emb.type == WT;
    uint32_t e_pq = 0u, e_psc = 0u, e_pqh = 0u, e_tq = 0u, e_tsc = 0u, e_tqh = 0u;
    int tok0 = 0;
    {
        const int tid = threadIdx.x;
        if (tid < 256) {
            tok0 = streams ? p.seq[col].token : state_tokens(p.st)[col];
            if (tok0 < 0 || tok0 >= p.n_vocab) tok0 = 0;
            if (emb_fast) {
                xp_row4_request<WT>(p.pos_emb, pos + 2, tid, e_pq, e_psc, e_pqh);
                xp_row4_request<WT>(p.tok_emb, tok0, tid, e_tq, e_tsc, e_tqh);
            }
        }
        request_small(0, tid);
        if constexpr (!ATTN) {
            request_qkv(0, tid);      // (waves 0 .. 3: the other units in layer 0's burst)
        } else {
            request_wo(0, tid);
            if constexpr (!LATE_W2) { request_w1(0, tid); request_w2(0, tid); }
            request_kv(0, tid);
        }
        if (tid < 256) {      // biogpt.cpp:664-686: embed_tokens[tok] * sqrt(D) + embed_positions[position + 2]; parked in LDS (no register stays live through the layer loop)
            const int tok = tok0;
            float e[4];
            if (emb_fast) {
                float te[4], pe[4];
                xp_row4_values<WT>(e_pq, e_psc, e_pqh, tid, pe);
                xp_row4_values<WT>(e_tq, e_tsc, e_tqh, tid, te);
#pragma unroll
                for (int j = 0; j < 4; j++) e[j] = __fadd_rn(__fmul_rn(te[j], p.embed_scale), pe[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) e[j] = __fadd_rn(__fmul_rn(dequant_elem(p.tok_emb, tok, 4 * tid + j), p.embed_scale), dequant_elem(p.pos_emb, pos + 2, 4 * tid + j));
            }
            reinterpret_cast<float4 *>(s_x)[tid] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }

    for (int L = 0; L < p.n_layer; L++) {
        // (the thread index goes through an empty asm in every iteration: see kernels_xpipe.hip.h -- without it a few hundred per-thread addresses are hoisted out of the loop and spilled)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const int sub = lane & 31, rsub = lane >> 5;
        const bool worker = tid < 256;
        const bool more = L + 1 < p.n_layer;
        xp_u64 *const G = Gcol + (size_t)L * XP_G_LAYER;
        // this layer's small vectors: registers -> LDS (the previous layer's readers are behind the barrier at the end of the loop), the next layer's requested
        if (worker) {
            reinterpret_cast<float4 *>(s_ln)[tid] = l0; reinterpret_cast<float4 *>(s_ln + 1024)[tid] = l1;
            reinterpret_cast<float4 *>(s_ln + 2048)[tid] = l2; reinterpret_cast<float4 *>(s_ln + 3072)[tid] = l3;
        }
        if (tid < 384) s_bias[tid] = bnext;
        // WHEN a wave asks for weights decides what its polls cost: vector memory operations of a wave return in order, so a poll issued behind a burst of weight
        // loads is answered only when those have come back from the fabric (2.5 us under this launch's load: measured, profiles/xcols_timeline_r4.txt, as +2.5 us on
        // every hand-off when each stage re-requested its units right behind their use).  Every workgroup therefore requests its units in ONE burst per layer, in
        // front of the one poll that has to wait for microseconds anyway: the q / k / v workgroups behind their rows (they wait for the attention), the attention
        // workgroups at the end of the layer (they wait for the next layer's q / k / v rows, and do not need the layer input before their out_proj rows).
        // ---- the layer input of this column, 4 elements per LayerNorm worker: the embedding (layer 0) or the previous layer's granules (same XCD) ----
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        auto layer_input = [&]() __attribute__((always_inline)) {
        if (L == 0) {
            if (worker) xv = reinterpret_cast<const float4 *>(s_x)[tid];      // the embedding: computed in front of the loop (this thread's own LDS write)
        } else if (wave < 4) {
            uint32_t v[4];
            xc_sweep<4, 256>(G - XP_G_LAYER + XP_G_X + tid, true, epoch, v, p);
            xv = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
        if (worker) reinterpret_cast<float4 *>(s_x)[tid] = xv;      // the residual of stage C
        };
        if (!ATTN || L == 0) { layer_input(); XC_WALL(0); }
        if constexpr (!ATTN) {
            // ================= stage A (workgroups 16-31): LayerNorm -> Q8 -> the 192 q / k / v rows of head `head` (biogpt.cpp:691-727) =================
            float4 lnw = xv, lnb = xv;
            if (worker) { lnw = reinterpret_cast<const float4 *>(s_ln)[tid]; lnb = reinterpret_cast<const float4 *>(s_ln + 1024)[tid]; }
            ln4_q8_1024<TI::q81, true>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
            XC_WALL(6);
            uint32_t ax[8];
            const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
            ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
            const float axd = s_xd[sub];
            const uint32_t axs = s_xs[sub];
            float *const part = s_part + wave * 2 * QS * DEC_PS;
#pragma unroll
            for (int s = 0; s < QS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = unit_dot_quant<WT>(wqkv[s], ax, axd, __uint_as_float(axs), (int)axs);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 2 * QS) {
                const int jj = (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);
                float v = __fadd_rn(s_bias[jj], sum32_in_order(part + lane * DEC_PS));
                const int which = jj >> 6, d = jj & 63;
                if (which == 0) {
                    v = __fmul_rn(v, p.q_scale);                                   // Q scaled AFTER the bias (biogpt.cpp:708-710)
                    xp_put_local(G + XP_G_QKV + head * 64 + d, epoch, __float_as_uint(v));
                } else {
                    xp_put(G + XP_G_QKV + which * 1024 + head * 64 + d, epoch, __float_as_uint(v));      // write-through: the other columns' XCDs poll these
                    float *kc_ = XPL(L).kcache, *vc_ = XPL(L).vcache;
                    asm volatile("" : "+s"(kc_), "+s"(vc_));      // both by scalar loads (a per-lane choice of the table's FIELD is a vector load of the pointer)
                    float *cache = streams ? ((which == 1) ? p.kroot : p.vroot) + kv_off + (size_t)L * p.P * 1024
                                           : ((which == 1) ? kc_ : vc_);    // KV append (biogpt.cpp:721-727), head-major cache: for later evals
                    ((__attribute__((address_space(1))) float *)cache)[((size_t)head * p.P + pos) * DK + d] = v;
                }
            }
            XC_WALL(1);
            // the burst: this layer's other units in the order of their use, then the next layer's q / k / v units and small vectors -- the attention takes microseconds
            {
                request_wo(L, tid); request_w1(L, tid); request_w2(L, tid);
                if (more && !QKV_LATE) request_qkv(L + 1, tid);
            }
            if (more) request_small(L + 1, tid);
        } else {
            // ================= stage B (workgroups 0-15): attention of head `head`, query = this column (biogpt.cpp:729-764, no mask inside the eval: F1) =================
            const int ksub = tid & (LPK - 1), kidx = tid / LPK;
            const int dd = tid & (DK - 1), sl = tid >> 6;
            // SEG2: the rows of keys 256 .. 511 (written by earlier evals; where a key is one of THIS eval's its pieces are replaced from LDS below) -- the K pieces of key
            // kidx + 256 and the V values of keys 256 + sl + 8 k, requested HERE, in front of the poll for the layer's new q / k / v rows: that poll waits for stage A of the
            // other workgroups (microseconds) anyway, and a wave's loads return in order -- the round trip to the memory side hides behind it instead of standing in the stage
            float4 kr2[SEG2 ? NF4 : 1];
            float vr2[SEG2 ? NV : 1];
            if constexpr (SEG2) {
                const float *kb = XPL(L).kcache + (size_t)head * p.P * DK, *vb = XPL(L).vcache + (size_t)head * p.P * DK;
                const __amdgpu_buffer_rsrc_t krs = xp_kv_rsrc(kb, p.P * DK * 4), vrs = xp_kv_rsrc(vb, p.P * DK * 4);
                const int ko = ((kidx + 256) * DK + 4 * ksub) * 4, vo = ((256 + sl) * DK + dd) * 4;
#pragma unroll
                for (int m = 0; m < NF4; m++) {
                    const xp_v4u t = __builtin_amdgcn_raw_buffer_load_b128(krs, ko, LPK * m * 16, 2);
                    kr2[m] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
                }
#pragma unroll
                for (int k = 0; k < NV; k++) vr2[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vrs, vo, NW * k * DK * 4, 2));
            }
            if (wave < PW) {   // this column's q row (own XCD) and the k / v rows of ALL columns (their XCDs): up to 64 + 1024 granules, at most five per lane, in ONE poll loop
                const xp_u64 *gq = G + XP_G_QKV + head * 64 + (tid & 63);
                const bool aq = tid < 64;
                constexpr int NKV = 16 / PW;
                const xp_u64 *gk[NKV];
                bool ak[NKV];
#pragma unroll
                for (int h2 = 0; h2 < NKV; h2++) {
                    const int idx = tid + 64 * PW * h2, c = idx >> 7, e = idx & 127;
                    ak[h2] = c < n_new;
                    gk[h2] = p.gran + ((size_t)(ak[h2] ? c_first + c : col) * p.n_layer + L) * XP_G_LAYER + XP_G_QKV + 1024 * (1 + (e >> 6)) + head * 64 + (e & 63);
                }
                uint32_t vq = 0u, vk[NKV];
#pragma unroll
                for (int h2 = 0; h2 < NKV; h2++) vk[h2] = 0u;
                for (uint32_t spins = 0;; spins++) {
                    bool ok = true;
                    if (aq) { const xp_u64 x = __hip_atomic_load(gq, XP_RLX); vq = (uint32_t)x; ok &= (uint32_t)(x >> 32) == epoch; }
#pragma unroll
                    for (int h2 = 0; h2 < NKV; h2++)
                        if (ak[h2]) { const xp_u64 x = __hip_atomic_load(gk[h2], XP_RLX); vk[h2] = (uint32_t)x; ok &= (uint32_t)(x >> 32) == epoch; }
                    if (__all(ok)) break;
                    if (spins >= XP_SPIN_MAX) { if (lane == 0) xc_fail(p, 1u); break; }
                    if ((spins & 1023u) == 1023u && __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) break;
                }
                if (aq) s_cur[tid] = __uint_as_float(vq);
#pragma unroll
                for (int h2 = 0; h2 < NKV; h2++) if (ak[h2]) s_new[tid + 64 * PW * h2] = __uint_as_float(vk[h2]);
            }
            __syncthreads();
            XC_WALL(7);
            float sc = -INFINITY, sc2 = -INFINITY;
            if ((tid & ~63) < LPK * T) {        // whole waves past the context skip the double-precision work
                if (kidx >= n_old && kidx < T) {
#pragma unroll
                    for (int m = 0; m < NF4; m++) kr[m] = *reinterpret_cast<const float4 *>(s_new + (kidx - n_old) * 128 + 4 * (LPK * m + ksub));
                }
                if (kidx < T) {
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                    for (int m = 0; m < NF4; m++) {
                        const float4 qm = *reinterpret_cast<const float4 *>(s_cur + 4 * (LPK * m + ksub));
                        a0 += (double)__fmul_rn(kr[m].x, qm.x); a1 += (double)__fmul_rn(kr[m].y, qm.y);
                        a2 += (double)__fmul_rn(kr[m].z, qm.z); a3 += (double)__fmul_rn(kr[m].w, qm.w);
                    }
                    double acc = (a0 + a1) + (a2 + a3);
                    if (LPK >= 2) acc += dpp_d<DPP_QUAD_XOR1>(acc);
                    if (LPK >= 4) acc += dpp_d<DPP_QUAD_XOR2>(acc);
                    if (LPK >= 8) acc += dpp_d<DPP_ROW_HALF_MIRROR>(acc);
                    sc = (float)acc;
                }
            }
            if constexpr (SEG2) {
                const int k2 = kidx + 256;
                if ((tid & ~63) < LPK * (T - 256)) {
                    if (k2 >= n_old && k2 < T) {
#pragma unroll
                        for (int m = 0; m < NF4; m++) kr2[m] = *reinterpret_cast<const float4 *>(s_new + (k2 - n_old) * 128 + 4 * (LPK * m + ksub));
                    }
                    if (k2 < T) {
                        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                        for (int m = 0; m < NF4; m++) {
                            const float4 qm = *reinterpret_cast<const float4 *>(s_cur + 4 * (LPK * m + ksub));
                            a0 += (double)__fmul_rn(kr2[m].x, qm.x); a1 += (double)__fmul_rn(kr2[m].y, qm.y);
                            a2 += (double)__fmul_rn(kr2[m].z, qm.z); a3 += (double)__fmul_rn(kr2[m].w, qm.w);
                        }
                        double acc = (a0 + a1) + (a2 + a3);
                        acc += dpp_d<DPP_QUAD_XOR1>(acc);
                        sc2 = (float)acc;
                    }
                }
            }
            const bool valid = kidx < T && ksub == 0;      // this lane holds a score of the head
            const bool valid2 = SEG2 && kidx + 256 < T && ksub == 0;
            float mx = wave_max_f32(fmaxf(sc, sc2));
            if (lane == 0) s_redf[wave] = mx;
            __syncthreads();
            mx = s_redf[0];
#pragma unroll
            for (int w = 1; w < NW; w++) mx = fmaxf(mx, s_redf[w]);
            XC_WALL(13);
            double sum = 0.0;
            if (valid) {
                const float val = h2f(p.exp_tab[f2h(__fsub_rn(sc, mx))]);      // ggml_soft_max: fp16 exp table (every workgroup's LDS slice is GELU's here)
                s_S[kidx] = val;
                sum = (double)val;
            }
            if (valid2) {
                const float val = h2f(p.exp_tab[f2h(__fsub_rn(sc2, mx))]);
                s_S[kidx + 256] = val;
                sum += (double)val;      // (a sum of fp16 values in double: exact in any order)
            }
            sum = wave_sum_f64(sum);
            if (lane == 0) s_redd[wave] = sum;
            __syncthreads();
            sum = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w++) sum += s_redd[w];
            const float inv = inv_sum_f32(sum);
            XC_WALL(14);
            {
                // all LDS reads first, no branches in the loop: a key past the context adds +0.0 (exact), never its stale weight
                constexpr int CH = NV < 8 ? NV : 8;      // softmax weights and new-row values fetched 8 at a time (register budget: 47 spilled VGPRs with 16 in the 256-key variant)
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int k0 = 0; k0 < NV; k0 += CH) {
                    float pj[CH], vn[CH];
#pragma unroll
                    for (int k = 0; k < CH; k++) {
                        const int j = sl + NW * (k0 + k), nj = j - n_old;
                        pj[k] = s_S[j];
                        vn[k] = s_new[((nj < 0 || nj > 7) ? 0 : nj) * 128 + 64 + dd];
                    }
#pragma unroll
                    for (int k = 0; k < CH; k += 2) {
                        const int j0 = sl + NW * (k0 + k), j1 = j0 + NW;
                        const double c0 = (double)__fmul_rn(j0 >= n_old ? vn[k] : vr[k0 + k], __fmul_rn(pj[k], inv));
                        const double c1 = (double)__fmul_rn(j1 >= n_old ? vn[k + 1] : vr[k0 + k + 1], __fmul_rn(pj[k + 1], inv));
                        a0 += (j0 < T) ? c0 : 0.0;
                        a1 += (j1 < T) ? c1 : 0.0;
                    }
                }
                if constexpr (SEG2) {      // the same for keys 256 + sl + 8 k
#pragma unroll
                    for (int k0 = 0; k0 < NV; k0 += CH) {
                        float pj[CH], vn[CH];
#pragma unroll
                        for (int k = 0; k < CH; k++) {
                            const int j = 256 + sl + NW * (k0 + k), nj = j - n_old;
                            pj[k] = s_S[j];
                            vn[k] = s_new[((nj < 0 || nj > 7) ? 0 : nj) * 128 + 64 + dd];
                        }
#pragma unroll
                        for (int k = 0; k < CH; k += 2) {
                            const int j0 = 256 + sl + NW * (k0 + k), j1 = j0 + NW;
                            const double c0 = (double)__fmul_rn(j0 >= n_old ? vn[k] : vr2[k0 + k], __fmul_rn(pj[k], inv));
                            const double c1 = (double)__fmul_rn(j1 >= n_old ? vn[k + 1] : vr2[k0 + k + 1], __fmul_rn(pj[k + 1], inv));
                            a0 += (j0 < T) ? c0 : 0.0;
                            a1 += (j1 < T) ? c1 : 0.0;
                        }
                    }
                }
                s_pv[tid] = a0 + a1;
            }
            __syncthreads();
            XC_WALL(15);
            if (tid < DK) {
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int s2 = 0; s2 < NW; s2 += 2) { t0 += s_pv[s2 * DK + tid]; t1 += s_pv[(s2 + 1) * DK + tid]; }
                const float o = (float)(t0 + t1);
                int8_t q8; float d8; uint32_t s8;
                q8_block32(o, TI::q81, q8, d8, s8, true);
                const uint32_t packed = xp_pack4(q8);
                const int blk = head * 2 + (tid >> 5);
                if ((tid & 3) == 0) xp_put_local(G + XP_G_ATT + head * 16 + (tid >> 2), epoch, packed);
                if ((tid & 31) == 0) { xp_put_local(G + XP_G_ATT + 256 + blk, epoch, __float_as_uint(d8)); xp_put_local(G + XP_G_ATT + 288 + blk, epoch, s8); }
            }
            if (L != 0) { layer_input(); XC_WALL(0); }      // the residual of stage C: there since the layer began, no load of this workgroup is in flight now
        }
        XC_WALL(2);
        // ================= stage C: out_proj + bias + residual (biogpt.cpp:767-772) =================
        if constexpr (LATE_W2) { request_w1(L, tid); request_w2(L, tid); }
        if (wave < 4) {      // 256 + 32 + 32 granules: two per lane in wave 0, one elsewhere
            uint32_t v[1], w[1] = {0u};
            if (wave == 0) {
                const xp_u64 *g2 = G + XP_G_ATT + 256 + tid;
                for (uint32_t spins = 0;; spins++) {
                    const xp_u64 a = __hip_atomic_load(G + XP_G_ATT + tid, XP_RLX), b = __hip_atomic_load(g2, XP_RLX);
                    v[0] = (uint32_t)a; w[0] = (uint32_t)b;
                    if (__all((uint32_t)(a >> 32) == epoch && (uint32_t)(b >> 32) == epoch)) break;
                    if (spins >= XP_SPIN_MAX) { if (lane == 0) xc_fail(p, 1u); break; }
                    if ((spins & 1023u) == 1023u && __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) break;
                }
                if (tid < 32) s_xd[tid] = __uint_as_float(w[0]);
                else s_xs[tid - 32] = w[0];
            } else {
                xc_sweep<1, 1>(G + XP_G_ATT + tid, true, epoch, v, p);
            }
            s_xq[tid] = v[0];
        }
        __syncthreads();
        XC_WALL(8);
        {
            uint32_t ax[8];
            const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
            ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
            const float axd = s_xd[sub];
            const uint32_t axs = s_xs[sub];
            float *const part = s_part + wave * 2 * OS * DEC_PS;
#pragma unroll
            for (int s = 0; s < OS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = unit_dot_quant<WT>(wo[s], ax, axd, __uint_as_float(axs), (int)axs);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 2 * OS) {
                const int lr = (lane >> 1) * 2 * NW + wave * 2 + (lane & 1), row = slot * 32 + lr;
                const float v = __fadd_rn(__fadd_rn(sum32_in_order(part + lane * DEC_PS), s_bias[192 + lr]), s_x[row]);
                xp_put_local(G + XP_G_X1 + xp_col_slot(row), epoch, __float_as_uint(v));
            }
        }
        XC_WALL(3);
        // ================= stage D: LayerNorm -> Q8 -> fc1 -> GELU -> Q8 (biogpt.cpp:777-787) =================
        float4 x1v = make_float4(0.f, 0.f, 0.f, 0.f), lnw = x1v, lnb = x1v;
        if (wave < 4) {
            uint32_t v[4];
            xc_sweep<4, 256>(G + XP_G_X1 + tid, true, epoch, v, p);
            x1v = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            reinterpret_cast<float4 *>(s_x1)[tid] = x1v;
            XC_WALL(9);
            lnw = reinterpret_cast<const float4 *>(s_ln + 2048)[tid]; lnb = reinterpret_cast<const float4 *>(s_ln + 3072)[tid];
        }
        ln4_q8_1024<TI::q81, true>(x1v, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
        XC_WALL(10);
        {
            uint32_t ax[8];
            const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
            ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
            const float axd = s_xd[sub];
            const uint32_t axs = s_xs[sub];
            float *const part = s_part + wave * 2 * FS * DEC_PS;
#pragma unroll
            for (int s = 0; s < FS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = unit_dot_quant<WT>(w1[s], ax, axd, __uint_as_float(axs), (int)axs);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 2 * FS) {
                const int jr = (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);
                const float v = __fadd_rn(s_bias[224 + jr], sum32_in_order(part + lane * DEC_PS));
                const uint32_t ix = f2h(v), neg = ix - 0x8000u;               // ggml_gelu: fp16 table
                uint16_t g16;
                if (ix < (uint32_t)p.gelu_p) g16 = s_gelu[ix];
                else if (ix <= 0x7C00u) g16 = (p.gelu_p > 0) ? (uint16_t)ix : p.gelu_tab[ix];
                else if (neg < (uint32_t)p.gelu_n) g16 = s_gelu[p.gelu_p + neg];
                else if (neg < 0x7C00u && p.gelu_n > 0) g16 = (uint16_t)p.gelu_z;
                else g16 = p.gelu_tab[ix];                                     // -inf, NaN (or no slice in LDS)
                s_g[jr] = h2f(g16);
            }
        }
        __syncthreads();
        XC_WALL(11);
        if (tid < 128) {
            int8_t q8; float d8; uint32_t s8;
            q8_block32(s_g[tid], TI::q81, q8, d8, s8, true);
            const uint32_t packed = xp_pack4(q8);
            const int blk = slot * 4 + (tid >> 5);
            if ((tid & 3) == 0) xp_put_local(G + XP_G_H + slot * 32 + (tid >> 2), epoch, packed);
            if ((tid & 31) == 0) { xp_put_local(G + XP_G_H + 1024 + blk, epoch, __float_as_uint(d8)); xp_put_local(G + XP_G_H + 1152 + blk, epoch, s8); }
        }
        XC_WALL(4);
        // ================= stage E: fc2 + bias + residual (biogpt.cpp:790-795) =================
        if (wave < 4) {   // 1024 + 128 + 128 granules in ONE poll loop, five per lane: every pass has all of a lane's loads in flight together
            uint32_t v[5];
            const xp_u64 *g = G + XP_G_H + tid;
            xc_sweep<5, 256>(g, true, epoch, v, p);
#pragma unroll
            for (int k = 0; k < 4; k++) s_hq[tid + k * 256] = v[k];
            if (tid < 128) s_hd[tid] = __uint_as_float(v[4]);
            else s_hs[tid - 128] = v[4];
        }
        __syncthreads();
        XC_WALL(12);
        {
            float *const part = s_part + wave * F2R * DEC_PS2;
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int u = lane + 64 * it;
                uint32_t ax[8];
                const uint4 a = *reinterpret_cast<const uint4 *>(s_hq + u * 8), b = *reinterpret_cast<const uint4 *>(s_hq + u * 8 + 4);
                ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
                const float axd = s_hd[u];
                const uint32_t axs = s_hs[u];
#pragma unroll
                for (int r = 0; r < F2R; r++) part[r * DEC_PS2 + u] = unit_dot_quant<WT>(w2[r][it], ax, axd, __uint_as_float(axs), (int)axs);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < F2R) {
                const float4 *p4 = reinterpret_cast<const float4 *>(part + lane * DEC_PS2);
                float sumf = 0.0f;
#pragma unroll
                for (int b0 = 0; b0 < 32; b0 += 8) {
                    float4 t[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) t[j] = p4[b0 + j];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        sumf = __fadd_rn(sumf, t[j].x); sumf = __fadd_rn(sumf, t[j].y);
                        sumf = __fadd_rn(sumf, t[j].z); sumf = __fadd_rn(sumf, t[j].w);
                    }
                }
                const int lr = wave * F2R + lane, row = slot * 32 + lr;
                const float v = __fadd_rn(__fadd_rn(sumf, s_bias[352 + lr]), s_x1[row]);
                if (more) xp_put_local(G + XP_G_X + xp_col_slot(row), epoch, __float_as_uint(v));
                else p.x_out[(size_t)col * 1024 + row] = v;
            }
        }
        XC_WALL(5);
        if constexpr (!ATTN && QKV_LATE) request_qkv(more ? L + 1 : L, tid);      // (unconditional: a conditional request would keep the old units alive through the whole layer)
        if constexpr (ATTN) {      // the burst of the attention workgroups: the next layer's units in the order of their use, the head's old keys / values, the small vectors
            if (more || UNCOND) {
                const int Ln = more ? L + 1 : L;
                request_wo(Ln, tid);
                if constexpr (!LATE_W2) { request_w1(Ln, tid); request_w2(Ln, tid); }
                request_kv(Ln, tid);
                if (more) request_small(L + 1, tid);
            }
        }
        __syncthreads();       // s_ln / s_bias / s_x / s_x1 / s_part are rewritten by the next layer
    }
}

template <int WT, int LPK, int KCAP>
__global__ __launch_bounds__(512) void dec_xcols_kernel(const XcParams p) {
    constexpr int NT = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *const s_redi = reinterpret_cast<int *>(smem + XP_S_REDF + 256);
    uint16_t *const s_gelu = reinterpret_cast<uint16_t *>(smem + XP_S_TOTAL);
    // XCD and rank inside the XCD: as in dec_xpipe_kernel (same control words: a chunk launch is one more launch of the context's pipeline)
    if (threadIdx.x == 0) {
        const uint32_t e0 = __hip_atomic_load(p.ctl, XP_RLX), launch = __hip_atomic_load(p.ctl + 2, XP_RLX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // both words are read BEFORE the ticket is taken: the launch's last act waits for all 256 tickets, then moves them on
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;     // HW_REG_XCC_ID
        const uint32_t t = __hip_atomic_fetch_add(p.ctl + 8 + xcc, 1u, XP_RLX);
        s_redi[0] = (int)xcc;
        s_redi[1] = (int)(t - 32u * (launch - 1u));
        s_redi[2] = (int)e0;
        s_redi[3] = (int)launch;
    }
    __syncthreads();
    const int xcd = __builtin_amdgcn_readfirstlane(s_redi[0]), slot = __builtin_amdgcn_readfirstlane(s_redi[1]);
    const uint32_t epoch = (uint32_t)__builtin_amdgcn_readfirstlane(s_redi[2]), launch = (uint32_t)__builtin_amdgcn_readfirstlane(s_redi[3]);
    __syncthreads();
    if ((unsigned)slot >= 32u) { if (threadIdx.x == 0) xc_fail(p, 2u); return; }
    if (xcd >= p.n_cols) return;       // fewer columns than XCDs: nothing to do here (column 0's workgroup 0 waits for this ticket too)
    {   // ggml_gelu's fp16 table (biogpt.cpp:784): the slices that are neither the identity nor a constant, in LDS for the whole launch
        const uint4 *src = reinterpret_cast<const uint4 *>(p.gelu_tab);
        const int np8 = p.gelu_p / 8, nn8 = p.gelu_n / 8;
        for (int i = threadIdx.x; i < np8; i += NT) reinterpret_cast<uint4 *>(s_gelu)[i] = src[i];
        for (int i = threadIdx.x; i < nn8; i += NT) reinterpret_cast<uint4 *>(s_gelu + p.gelu_p)[i] = src[0x8000 / 8 + i];
    }
#ifdef XC_ONLY_ROLE      // (register census of one role: hipcc -Rpass-analysis=kernel-resource-usage -DXC_ONLY_ROLE=r; not a working kernel)
    xc_run<WT, LPK, KCAP, XC_ONLY_ROLE>(p, smem, xcd, slot, epoch);
#else
    if (slot < 16) xc_run<WT, LPK, KCAP, 0>(p, smem, xcd, slot, epoch);
    else xc_run<WT, LPK, KCAP, 1>(p, smem, xcd, slot, epoch);
#endif
    if (xcd == 0 && slot == 0 && threadIdx.x == 0) {
        // the next launch hands out tickets 32 launch .. and uses tag epoch + 1: only once every workgroup of THIS launch has taken its ticket (and read both words before)
        bool all = false;
        for (uint32_t spins = 0; !all && spins < XP_SPIN_MAX; spins++) {
            all = true;
            for (int x = 0; x < 8; x++) all &= __hip_atomic_load(p.ctl + 8 + x, XP_RLX) >= 32u * launch;
        }
        if (!all) xc_fail(p, 2u);
        if (epoch + 1u > 0xF0000000u) xc_fail(p, 5u);
        __hip_atomic_store(p.ctl, epoch + 1u, XP_RLX);
        __hip_atomic_store(p.ctl + 2, launch + 1u, XP_RLX);
    }
}

}  // namespace bgk
