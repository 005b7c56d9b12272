// The XCD-pipelined decode step (kernels_xpipe.hip.h) for contexts BEYOND 256 keys: 257 .. 512 keys (KR = 32) and 513 .. 1024 keys
// (KR = 64).  biogpt.cpp:664-811 for all layers + the output projection; the attention of biogpt.cpp:729-764 is spread over the chip.
//
// Why another shape: up to 256 keys a head's workgroup keeps the head's old K / V rows in registers (<= 64 KB).  At 1024 keys a
// layer's K / V are 8 MB -- 512 KB per head -- which no single compute unit, and not even one XCD's L2 (4 MB), can hold; the
// five-launch layer with its three key-split attention launches (attn_split_*_kernel) costs 25 us per layer there (608 us per token).
// Here the weights stay exactly where kernels_xpipe.hip.h puts them (unit u = half a layer on XCD u % 8, stationary in registers,
// loaded 7 units ahead), and ONLY the attention is redistributed:
//
//   * head h belongs to XCD h / 2 -- for EVERY layer.  Workgroup (xcd, slot) is the "helper" of head 2 xcd + slot / 16 and key
//     range r = slot % 16: keys [r KR, (r + 1) KR).  It holds its range's K and V rows of the NEXT layer in 8-16 registers per
//     lane (32 KB per workgroup per layer, loaded one layer ahead): all 256 compute units stream the cache, 8 MB per layer, while
//     they wait for their own unit's turn anyway.
//   * the layer's own (even) XCD computes LayerNorm + the q / k / v rows as before and publishes them as granules (one cross-XCD hop);
//     the helper of the range that holds position n_past appends the new K / V rows to the cache and uses them from LDS.
//   * helper: 64 (32) scores -> granules inside ITS XCD -> all 16 helpers of the head read the head's T scores (one in-XCD hop),
//     each computes the global maximum, every e_j = exp_table(S_j - max) and the double sum itself (identical in all 16: sums of
//     <= 1024 fp16-valued terms in double are exact whatever the order), then p_j = fl(e_j * (float)(1 / sum)) and its partial
//     sum_j V_jd p_j over its own keys in double (the arithmetic of attn_split_pv_kernel) -> 64 doubles as granules (in-XCD hop)
//     -> the head's first helper adds the partials in range order (attn_split_combine_kernel's association), quantizes the 64
//     outputs to two Q8 blocks and publishes them for the layer's XCD (one cross-XCD hop), where out_proj runs as before.
//   * K / V rows appended during a multi-token launch are written and later re-read by the SAME workgroup (plain stores, L1-bypassing
//     loads): no cross-XCD visibility question arises for the cache.
//
// Four hops (two cross-XCD, two in-XCD) instead of the <= 256-key variants' two in-XCD ones: about +2.6 us per layer, against
// +13 us for three launches.  Everything else -- stages A, C, D, E, the lm_head on the XCDs that are done, the sampler of the next
// token on XCD 0, tags, bounded spins, error word -- is kernels_xpipe.hip.h's, restructured so that every workgroup walks ALL
// layers in order (helper duty for each, its own unit's stages in between).
#pragma once

#include "kernels_xpipe.hip.h"

namespace bgk {

// (granules of one layer in the long-context buffer, XpParams::gran_l: XL_G_SC / XL_G_PV / XL_G_LAYER in kernels_xpipe.hip.h)

#ifdef BIOGPT_HIP_PROFILE_HOOKS
// wall clock of workgroups 0 and 16 of the layer's own XCD ([n_layer][32] slots; workgroup 0 is also the first helper of head 2 xcd)
#define XL_WALL(k) do { if (p.wall && tid == 0 && (slot & 15) == 0 && own_first) p.wall[L * 32 + (k)] = wall_clock64(); } while (0)
#define XL_WALL2(k) do { if (p.wall && tid == 0 && slot == 0) p.wall[L * 32 + (k)] = wall_clock64(); } while (0)      // the MLP half's workgroup 0
#else
#define XL_WALL(k) do {} while (0)
#define XL_WALL2(k) do {} while (0)
#endif

// RES (the resident instantiation, biogpt_hip_eval's loop: kernels_xpipe.hip.h, XpParams::resident): a launch that is asked to leave -- or gives up waiting, or
// fails -- must not hand anything downstream that looks like a result, append a K / V row or write a logits row.  Here a wave whose poll fails simply ENDS
// (s_endpgm) after raising a flag in LDS; a barrier releases the surviving waves of the workgroup (tools/microbench15.hip), and they look at the flag behind
// every barrier that stands between a poll and something they publish (xl_live).  Nothing is ever published from data that did not arrive.
__device__ __forceinline__ void xl_die(uint32_t *s_dead) {
    __hip_atomic_store(s_dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (an atomic, not a volatile access: the LDS address space is inferred through it)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_endpgm();
}
// the launch's error / quit word, as a wave-uniform value (one scalar branch, no lane masks)
__device__ __forceinline__ bool xl_err(const XpParams &p) { return __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p.ctl + 1, XP_RLX)) != 0; }
template <bool RES>
__device__ __forceinline__ void xl_live(uint32_t *s_dead) {
    if constexpr (RES) {
        if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(s_dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0) __builtin_amdgcn_endpgm();
    }
}
// CROSS: a hand-off that crosses XCDs -- two passes in flight (kernels_xpipe.hip.h, xp_sweep_pipelined)
template <bool RES, int N, int S = 1, bool CROSS = false>
__device__ __forceinline__ void xl_sweep(const xp_u64 *g, bool active, uint32_t epoch, uint32_t (&v)[N], const XpParams &p, uint32_t *s_dead) {
    if constexpr (!RES) {
        if constexpr (CROSS) xp_sweep_pipelined<N, S>(g, active, epoch, v, p);
        else xp_sweep<N, S>(g, active, epoch, v, p);
        return;
    }
    if constexpr (CROSS) {
        xp_u64 cur[N];
#pragma unroll
        for (int k = 0; k < N; k++) cur[k] = active ? __hip_atomic_load((xp_gq)g + k * S, XP_RLX) : ((xp_u64)epoch << 32);
        for (uint32_t spins = 0;; spins++) {
            xp_u64 nxt[N];
#pragma unroll
            for (int k = 0; k < N; k++) nxt[k] = active ? __hip_atomic_load((xp_gq)g + k * S, XP_RLX) : ((xp_u64)epoch << 32);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < N; k++) { v[k] = (uint32_t)cur[k]; ok &= (uint32_t)(cur[k] >> 32) == epoch; }
            if (__all(ok)) return;
            if (spins >= XP_SPIN_MAX) { if ((threadIdx.x & 63) == 0) xp_fail(p, 1u); xl_die(s_dead); }
            if ((spins & 255u) == 255u && xl_err(p)) xl_die(s_dead);
#pragma unroll
            for (int k = 0; k < N; k++) cur[k] = nxt[k];
        }
    }
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
        if (spins >= XP_SPIN_MAX) { if ((threadIdx.x & 63) == 0) xp_fail(p, 1u); xl_die(s_dead); }
        if ((spins & 255u) == 255u && xl_err(p)) xl_die(s_dead);
    }
}

// ROLE 0: workgroups 0-15 of an even XCD (LayerNorm + the 192 q / k / v rows of head `slot`, out_proj rows); 1: workgroups 16-31 of an even
// XCD (out_proj rows only); 2: the 32 workgroups of an odd XCD (LayerNorm, fc1, fc2).  Every role is a helper for every layer.
template <int WT, int KR, int ROLE, bool RES>
__device__ __forceinline__ void xl_run(const XpParams &p, unsigned char *smem, const int xcd, const int slot, const uint32_t epoch0, const int n_past0,
                                       const int n_gen0, const uint32_t launch0) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant, "block-quantized weights");
    static_assert(KR == 32 || KR == 64, "keys per helper: 16 ranges cover 512 / 1024 keys");
    constexpr bool FIRST = ROLE != 2, SECOND = ROLE == 2;
    constexpr int NW = 8, NT = 512, DK = 64;
    constexpr int QS = 96 / NW, OS = 16 / NW, FS = 64 / NW, F2R = 32 / NW, LMS = 128 / NW;
    constexpr int LPK = NT / KR, NF4 = 16 / LPK, NV = KR / NW, NSC = 16 * KR / NT;      // lanes per key; float4 of a key row per lane; values per lane; scores polled per lane
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
    float *const s_g = reinterpret_cast<float *>(smem + XP_S_G);
    float *const s_ln = reinterpret_cast<float *>(smem + XP_S_LN);
    float *const s_bias = reinterpret_cast<float *>(smem + XP_S_BIAS);
    float *const s_cur = reinterpret_cast<float *>(smem + XP_S_CUR);
    float *const s_S = reinterpret_cast<float *>(smem + XP_S_S);
    float *const s_redf = reinterpret_cast<float *>(smem + XP_S_REDF);
    int *const s_redi = reinterpret_cast<int *>(smem + XP_S_REDF + 256);
    double *const s_redd = reinterpret_cast<double *>(smem + XP_S_REDD);
    double *const s_pv = reinterpret_cast<double *>(smem + XP_S_PV);
    uint16_t *const s_gelu = reinterpret_cast<uint16_t *>(smem + XP_S_TOTAL);
    uint32_t *const s_dead = reinterpret_cast<uint32_t *>(smem + XP_S_REDD + 96);      // RES: a wave of this workgroup has ended (xl_die); zeroed by the kernel
    uint32_t *const s_spec = reinterpret_cast<uint32_t *>(smem + XP_S_REDD + 112);     // RES, workgroup 0 of XCD 0: kernels_xpipe.hip.h's s_spec
    const int n_layer = p.n_layer, n_units = 2 * n_layer, last_xcd = (n_units - 1) & 7;
    const int P = p.P;
    // helper duty: head and key range of this workgroup, the same for every layer and token
    const int hx_head = 2 * xcd + (slot >> 4), hx_r = slot & 15, hx_j0 = hx_r * KR;
    // own pipeline units: layers L with L % 4 == my_l0 (their first half on an even XCD, their second half on the next, odd one)
    const int my_l0 = xcd >> 1;
    const int my_last = (my_l0 < n_layer) ? my_l0 + ((n_layer - 1 - my_l0) & ~3) : -1;     // last own layer (-1: none)
    // lm_head: XCD 0 (the next token's layer 0) and the last unit's XCD take no part
    const int lm_xr = xcd - (xcd > 0 ? 1 : 0) - ((last_xcd != 0 && xcd > last_xcd) ? 1 : 0);
    const int lm_rank = slot + 32 * lm_xr;
    const bool lm_mine = p.lm != 0 && xcd != 0 && xcd != last_xcd && lm_rank * 4 < XPK(lm_blocks);

    // this workgroup's share of the NEXT layer's old keys / values (rows of positions it has appended itself come back through its own L2)
    float4 kr[NF4];
    float vr[NV];
    auto fetch_kv = [&](const int L, const int n_past) __attribute__((always_inline)) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (hx_j0 <= n_past) {          // the range is, or becomes, active with this token
            const XpLayerK &Y = XPL(L);
            const int ksub = tid & (LPK - 1), kidx = tid / LPK, dd = tid & (DK - 1), sl = tid >> 6;
            // streaming loads (kernels_xpipe.hip.h, xp_kv_load*<false>): the rows of this range that were appended during the launch were appended by THIS workgroup
            const float *kb = Y.kcache + (size_t)hx_head * P * DK, *vb = Y.vcache + (size_t)hx_head * P * DK;
            const __amdgpu_buffer_rsrc_t krs = xp_kv_rsrc(kb, P * DK * 4), vrs = xp_kv_rsrc(vb, P * DK * 4);
            if (hx_j0 + kidx < P) {
#pragma unroll
                for (int m = 0; m < NF4; m++) kr[m] = xp_kv_load4<false>(krs, kb, ((hx_j0 + kidx) * (DK / 4) + ksub + LPK * m) * 4);
            }
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const int j = hx_j0 + sl + NW * k;
                if (j < P) vr[k] = xp_kv_load1<false>(vrs, vb, j * DK + dd);
            }
        }
    };

    // ---- the partial sums of one head added in range order (attn_split_combine_kernel's association), Q8, published for out_proj.  Round 3: by the head's first
    //      helper on XCD head / 2 (partials in-XCD, the result across: two hops).  Round 4: by workgroup `head` of the layer's OWN XCD -- the one
    //      that computed the head's q / k / v rows and waits for the attention output anyway: the partials cross XCDs (one hop, 2048 granules per head), the result
    //      stays inside the XCD (plain stores): T = 1024 14.25 -> ~13.2 us per layer (profiles/xlong_timeline_r4.txt) ----
    auto combine = [&](const int L, const uint32_t epoch, const int T, const int head) __attribute__((always_inline)) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const bool own_first = true;      // (profiling hooks: the workgroup whose stamps are kept)
        {
            __syncthreads();                      // s_pv: the slice sums of the helper duty have been read
            const int nr = (T + KR - 1) / KR;     // active ranges
            const xp_u64 *const G0 = p.gran_l + (size_t)L * XL_G_LAYER + XL_G_PV + head * 16 * 128;
            {
                const int d = tid & (DK - 1), part = tid >> 6;          // ranges part and part + 8
                uint32_t v[4] = {0u, 0u, 0u, 0u};
                const bool a0 = part < nr, a1 = part + 8 < nr;
                const xp_u64 *g0 = G0 + part * 128 + d, *g1 = G0 + (part + 8) * 128 + d;
                // (one pass at a time: two in flight were measured here -- T = 1024 358 -> 365 us per token, profiles/xpipe_ab_r4.txt)
                for (uint32_t spins = 0;; spins++) {
                    bool ok = true;
                    if (a0) {
                        const xp_u64 x0 = __hip_atomic_load(g0, XP_RLX), x1 = __hip_atomic_load(g0 + 64, XP_RLX);
                        v[0] = (uint32_t)x0; v[1] = (uint32_t)x1;
                        ok &= (uint32_t)(x0 >> 32) == epoch && (uint32_t)(x1 >> 32) == epoch;
                    }
                    if (a1) {
                        const xp_u64 x0 = __hip_atomic_load(g1, XP_RLX), x1 = __hip_atomic_load(g1 + 64, XP_RLX);
                        v[2] = (uint32_t)x0; v[3] = (uint32_t)x1;
                        ok &= (uint32_t)(x0 >> 32) == epoch && (uint32_t)(x1 >> 32) == epoch;
                    }
                    if (__all(ok)) break;
                    if (spins >= XP_SPIN_MAX) { if ((tid & 63) == 0) xp_fail(p, 7u); if (RES) xl_die(s_dead); break; }
                    if ((spins & (RES ? 255u : 1023u)) == (RES ? 255u : 1023u) && (RES ? xl_err(p) : __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u)) { if (RES) xl_die(s_dead); break; }
                }
                s_pv[part * DK + d] = a0 ? __hiloint2double((int)v[1], (int)v[0]) : 0.0;
                s_pv[(part + 8) * DK + d] = a1 ? __hiloint2double((int)v[3], (int)v[2]) : 0.0;
            }
            __syncthreads();
            xl_live<RES>(s_dead);
            XL_WALL(21);
            if (tid < DK) {
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int s = 0; s < 16; s += 2) {
                    if (s < nr) t0 += s_pv[s * DK + tid];
                    if (s + 1 < nr) t1 += s_pv[(s + 1) * DK + tid];
                }
                const float o = (float)(t0 + t1);
                int8_t q8; float d8; uint32_t s8;
                q8_block32(o, TI::q81, q8, d8, s8, TI::q81);
                const uint32_t packed = xp_pack4(q8);
                xp_u64 *const G = p.gran + (size_t)L * XP_G_LAYER;
                const int blk = head * 2 + (tid >> 5);
                if ((tid & 3) == 0) xp_put_local(G + XP_G_ATT + head * 16 + (tid >> 2), epoch, packed);
                if ((tid & 31) == 0) { xp_put_local(G + XP_G_ATT + 256 + blk, epoch, __float_as_uint(d8)); if (TI::q81) xp_put_local(G + XP_G_ATT + 288 + blk, epoch, s8); }
            }
            XL_WALL(22);
            __syncthreads();                      // s_pv is rewritten by the next helper duty
        }
        (void)lane;
    };

    // ================= helper duty for layer L (every workgroup, every layer): biogpt.cpp:729-764 for keys [hx_j0, hx_j0 + KR) of head hx_head =================
    auto helper = [&](const int L, const uint32_t epoch, const int n_past, const bool own_first, const bool more, const bool defer_fetch = false) __attribute__((always_inline)) {
        const int T = n_past + 1;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = RES ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
        if (hx_j0 < T) {
            const XpLayerK &Y = XPL(L);
            xp_u64 *const G = p.gran + (size_t)L * XP_G_LAYER;
            xp_u64 *const GL = p.gran_l + (size_t)L * XL_G_LAYER;
            const bool has_new = n_past < hx_j0 + KR;                  // (and n_past >= hx_j0): this range holds the token's own key
            // ---- the head's query row; the new key / value rows where they belong to this range (also appended to the cache: biogpt.cpp:721-727) ----
            if (wave == 0 || (has_new && wave < 3)) {
                uint32_t v[1];
                xl_sweep<RES, 1, 1, true>(G + XP_G_QKV + wave * 1024 + hx_head * 64 + lane, true, epoch, v, p, s_dead);
                s_cur[tid] = __uint_as_float(v[0]);
                if (wave != 0) {
                    float *kc_ = Y.kcache, *vc_ = Y.vcache;
                    ((__attribute__((address_space(1))) float *)((wave == 1) ? kc_ : vc_))[((size_t)hx_head * P + n_past) * DK + lane] = __uint_as_float(v[0]);
                }
            }
            __syncthreads();
            xl_live<RES>(s_dead);
            XL_WALL(16);
            // ---- scores of the own keys ----
            const int ksub = tid & (LPK - 1), kidx = tid / LPK, j = hx_j0 + kidx;
            {
                if (j == n_past) {
#pragma unroll
                    for (int m = 0; m < NF4; m++) kr[m] = *reinterpret_cast<const float4 *>(s_cur + 64 + 4 * (LPK * m + ksub));
                }
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int m = 0; m < NF4; m++) {
                    const float4 qm = *reinterpret_cast<const float4 *>(s_cur + 4 * (LPK * m + ksub));
                    a0 += (double)__fmul_rn(kr[m].x, qm.x); a1 += (double)__fmul_rn(kr[m].y, qm.y);
                    a2 += (double)__fmul_rn(kr[m].z, qm.z); a3 += (double)__fmul_rn(kr[m].w, qm.w);
                }
                double acc = (a0 + a1) + (a2 + a3);
                acc += dpp_d<DPP_QUAD_XOR1>(acc);
                acc += dpp_d<DPP_QUAD_XOR2>(acc);
                acc += dpp_d<DPP_ROW_HALF_MIRROR>(acc);
                if (LPK >= 16) acc += dpp_d<DPP_ROW_MIRROR>(acc);
                if (ksub == 0 && j < T) xp_put_local(GL + XL_G_SC + hx_head * 1024 + j, epoch, __float_as_uint((float)acc));
            }
            XL_WALL(17);
            // ---- all T scores of the head: global maximum, fp16-table exp, double sum (ggml_soft_max) -- identical in the head's 16 helpers ----
            float sc[NSC];
            {
                uint32_t v[NSC];
                const xp_u64 *g = GL + XL_G_SC + hx_head * 1024 + tid;
                for (uint32_t spins = 0;; spins++) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < NSC; k++) {
                        if (tid + NT * k < T) {
                            const xp_u64 a = __hip_atomic_load(g + NT * k, XP_RLX);
                            v[k] = (uint32_t)a;
                            ok &= (uint32_t)(a >> 32) == epoch;
                        }
                    }
                    if (__all(ok)) break;
                    if (spins >= XP_SPIN_MAX) { if (lane == 0) xp_fail(p, 6u); if (RES) xl_die(s_dead); break; }
                    if ((spins & (RES ? 255u : 1023u)) == (RES ? 255u : 1023u) && (RES ? xl_err(p) : __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u)) { if (RES) xl_die(s_dead); break; }
                }
#pragma unroll
                for (int k = 0; k < NSC; k++) sc[k] = (tid + NT * k < T) ? __uint_as_float(v[k]) : -INFINITY;
            }
            XL_WALL(18);
            float mx = sc[0];
#pragma unroll
            for (int k = 1; k < NSC; k++) mx = fmaxf(mx, sc[k]);
            mx = wave_max_f32(mx);
            if (lane == 0) s_redf[wave] = mx;
            __syncthreads();
            mx = s_redf[0];
#pragma unroll
            for (int w = 1; w < NW; w++) mx = fmaxf(mx, s_redf[w]);
            double sum = 0.0;
#pragma unroll
            for (int k = 0; k < NSC; k++) {
                const int jj = tid + NT * k;
                if (jj < T) {
                    const float val = h2f(p.exp_tab[f2h(__fsub_rn(sc[k], mx))]);
                    sum += (double)val;
                    if (jj >= hx_j0 && jj < hx_j0 + KR) s_S[jj - hx_j0] = val;
                }
            }
            sum = wave_sum_f64(sum);
            if (lane == 0) s_redd[wave] = sum;
            __syncthreads();
            sum = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w++) sum += s_redd[w];
            const float inv = inv_sum_f32(sum);
            XL_WALL(19);
            // ---- partial sum_j V_jd p_j over the own keys, in double (attn_split_pv_kernel) ----
            {
                const int dd = tid & (DK - 1), sl = tid >> 6;
                const float vcur = s_cur[128 + dd];
                float pj[NV];
#pragma unroll
                for (int k = 0; k < NV; k++) pj[k] = s_S[sl + NW * k];
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int k = 0; k < NV; k += 2) {
                    const int j0 = hx_j0 + sl + NW * k, j1 = j0 + NW;
                    const double c0 = (double)__fmul_rn(j0 == n_past ? vcur : vr[k], __fmul_rn(pj[k], inv));
                    const double c1 = (double)__fmul_rn(j1 == n_past ? vcur : vr[k + 1], __fmul_rn(pj[k + 1], inv));
                    a0 += (j0 < T) ? c0 : 0.0;
                    a1 += (j1 < T) ? c1 : 0.0;
                }
                s_pv[tid] = a0 + a1;
            }
            __syncthreads();
            xl_live<RES>(s_dead);
            if (tid < DK) {
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int s2 = 0; s2 < NW; s2 += 2) { t0 += s_pv[s2 * DK + tid]; t1 += s_pv[(s2 + 1) * DK + tid]; }
                const double part = t0 + t1;
                xp_u64 *const gp = GL + XL_G_PV + (hx_head * 16 + hx_r) * 128 + tid;
                // the partials go straight to the layer's own XCD (write-through), where the head's workgroup adds them (stage C's prologue)
                xp_put(gp, epoch, (uint32_t)__double2loint(part)); xp_put(gp + 64, epoch, (uint32_t)__double2hiint(part));
            }
            XL_WALL(20);
        }
        // this workgroup's K / V rows for its next helper duty (the next layer of this token, or layer 0 of the next token) -- issued AFTER the combiner's
        // polls, and on the layer's own XCD after stage C has published its rows: 32 KB of cache-missing loads occupy the compute unit's memory
        // pipeline for ~1.5 us, and whatever is issued behind them -- a poll (a wave's loads return in order), even a hand-off store -- waits
        // (measured: +1.9 .. 2.6 us on the chain in every earlier place)
        if (!own_first && !defer_fetch) {
            if (L + 1 < n_layer) fetch_kv(L + 1, n_past);
            else if (more) fetch_kv(0, n_past + 1);
        }
    };

    for (int tk = 0; tk < p.n_tok; tk++) {
        const uint32_t epoch = epoch0 + (uint32_t)tk;
        const int n_past = n_past0 + tk;
        const int n_gen = n_gen0 + tk;
        const bool more = tk + 1 < p.n_tok;
        if (tk > 0 && __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) {       // a disturbed launch drains token by token -- a resident one ends on the spot
            if (RES) __builtin_amdgcn_endpgm();
            break;
        }
        if (tk == 0) fetch_kv(0, n_past);
        // the unit weights are per-token objects: nothing of them is carried from one token to the next (register budget: 14-16 units beside
        // the lm_head's 16 would not fit)
        Unit<WT> wqkv[FIRST && ROLE == 0 ? QS : 1], wo[FIRST ? OS : 1], w1[SECOND ? FS : 1], w2[SECOND ? F2R : 1][2];
        // ---- the weights of own layer L into registers (unpacked there while they wait), its small vectors into LDS ----
        // part -1: the whole unit (a token's first own layer).  Parts 0 / 1 / 2: a third of it each -- the next own layer's weights are fetched in the idle time
        // behind the three helper duties in between (32 KB of key / value rows + <= 110 KB of weights per compute unit each time), never right behind the own
        // stages: a unit's 164 (Q4_0) .. 295 KB (Q8_0) occupy the compute unit's memory pipeline for 7 .. 12 us, and every poll of the next layer's helper duty
        // issued behind them waits -- on the chain of a layer this XCD does not even own (measured: 15.3 us per layer, Q8_0 19)
        auto load_unit_weights = [&](const int L, const int part) __attribute__((always_inline)) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63, wave = RES ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6), sub = lane & 31, rsub = lane >> 5;
            const bool worker = tid < 256;
            const XpLayerK &Y = XPL(L);
            auto sel = [&](const int unit) -> bool {      // ROLE 0: units 0-11 q/k/v, 12-13 out_proj; ROLE 1: 12-13; ROLE 2: 0-7 fc1, 8 + 2 r + it fc2
                if (part < 0) return true;
                if (ROLE == 2) return part == 0 ? (unit < 4 || (unit >> 1) == 4) : part == 1 ? ((unit >= 4 && unit < 8) || (unit >> 1) == 5) : unit >= 12;
                return part == 0 ? unit < 5 : part == 1 ? (unit >= 5 && unit < 10) : unit >= 10;
            };
            if (part <= 0) {      // the small vectors travel with the first part
                float4 l0 = make_float4(0.f, 0.f, 0.f, 0.f), l1 = l0;
                if (worker) {
                    if (FIRST) { l0 = xp_ldg4(Y.ln0_w, tid); l1 = xp_ldg4(Y.ln0_b, tid); }
                    else { l0 = xp_ldg4(Y.ln1_w, tid); l1 = xp_ldg4(Y.ln1_b, tid); }
                }
                float bv = 0.0f;
                if (tid < 192) { if (ROLE == 0) bv = ((xp_gf)Y.bqkv)[(tid >> 6) * 1024 + slot * 64 + (tid & 63)]; }
                else if (tid < 224) { if (FIRST) bv = ((xp_gf)Y.bo)[slot * 32 + tid - 192]; }
                else if (tid < 352) { if (SECOND) bv = ((xp_gf)Y.b1)[slot * 128 + tid - 224]; }
                else if (tid < 384) { if (SECOND) bv = ((xp_gf)Y.b2)[slot * 32 + tid - 352]; }
                if (worker) { reinterpret_cast<float4 *>(s_ln)[tid] = l0; reinterpret_cast<float4 *>(s_ln + 1024)[tid] = l1; }
                if (tid < 384) s_bias[tid] = bv;
            }
            if constexpr (ROLE == 0) {
#pragma unroll
                for (int s = 0; s < QS; s++) {
                    const int jj = s * 2 * NW + wave * 2 + rsub;
                    if (sel(s)) load_unit<WT>(wqkv[s], XPL_MATRIX(Y.Wqkv), (int64_t)((jj >> 6) * 1024 + slot * 64 + (jj & 63)) * 32 + sub);
                }
            }
            if constexpr (FIRST) {
#pragma unroll
                for (int s = 0; s < OS; s++)
                    if (sel(12 + s)) load_unit<WT>(wo[s], XPL_MATRIX(Y.Wo), (int64_t)(slot * 32 + s * 2 * NW + wave * 2 + rsub) * 32 + sub);
            }
            if constexpr (SECOND) {
#pragma unroll
                for (int s = 0; s < FS; s++)
                    if (sel(s)) load_unit<WT>(w1[s], XPL_MATRIX(Y.W1), (int64_t)(slot * 128 + s * 2 * NW + wave * 2 + rsub) * 32 + sub);
#pragma unroll
                for (int r = 0; r < F2R; r++)
#pragma unroll
                    for (int it = 0; it < 2; it++)
                        if (sel(8 + 2 * r + it)) load_unit<WT>(w2[r][it], XPL_MATRIX(Y.W2), (int64_t)(slot * 32 + wave * F2R + r) * 128 + lane + 64 * it);
            }
            // unpacked right here: the wait for the loads falls into this workgroup's idle time
            if constexpr (ROLE == 0) {
#pragma unroll
                for (int s = 0; s < QS; s++)
                    if (sel(s)) xp_settle<WT, true>(wqkv[s]);
            }
            if constexpr (FIRST) {
#pragma unroll
                for (int s = 0; s < OS; s++)
                    if (sel(12 + s)) xp_settle<WT, true>(wo[s]);
            }
            if constexpr (SECOND) {
#pragma unroll
                for (int s = 0; s < FS; s++)
                    if (sel(s)) xp_settle<WT, true>(w1[s]);
#pragma unroll
                for (int r = 0; r < F2R; r++)
#pragma unroll
                    for (int it = 0; it < 2; it++)
                        if (sel(8 + 2 * r + it)) xp_settle<WT, true>(w2[r][it]);
            }
        };

        // An own layer: [stage A] -> helper duty -> [stage C | stages D, E] (+ the next own unit's weights); every other layer: helper duty only.
        // (Three separate walks below, so that no unit weights are live before the first and after the last own layer.)
        auto stage_pre = [&](const int L) __attribute__((always_inline)) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63, wave = RES ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
            const int sub = lane & 31, rsub = lane >> 5;
            const bool worker = tid < 256;
            constexpr bool own_first = FIRST;
            xp_u64 *const G = p.gran + (size_t)L * XP_G_LAYER;
            if constexpr (FIRST) {
                {
                    // ---- the layer input: the embedding of the sampled token (layer 0) or the previous layer's granules ----
                    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (L == 0) {
                        int tok;
                        // greedy sampler of the previous token (main.cpp:109-128, top_k = 1): arg-max over the per-block partials of its logits --
                        // granules from the lm_head XCDs inside a multi-token launch, else the partials the previous launch left; lowest id wins ties
                        auto sample_prev = [&]() __attribute__((always_inline)) -> int {
                            float bv = -INFINITY;
                            int bi = 0x7fffffff;
                            if (tk > 0) {
                                uint32_t v[4] = {0u, 0u, 0u, 0u};
                                const bool a0 = tid < XPK(lm_blocks), a1 = tid + NT < XPK(lm_blocks);
                                const uint32_t prev = epoch - 1u;
                                for (uint32_t spins = 0;; spins++) {
                                    bool ok = true;
                                    if (a0) {
                                        const xp_u64 x0 = __hip_atomic_load(XPK(samp) + tid, XP_RLX), x1 = __hip_atomic_load(XPK(samp) + 1024 + tid, XP_RLX);
                                        v[0] = (uint32_t)x0; v[1] = (uint32_t)x1;
                                        ok &= (uint32_t)(x0 >> 32) == prev && (uint32_t)(x1 >> 32) == prev;
                                    }
                                    if (a1) {
                                        const xp_u64 x0 = __hip_atomic_load(XPK(samp) + tid + NT, XP_RLX), x1 = __hip_atomic_load(XPK(samp) + 1024 + tid + NT, XP_RLX);
                                        v[2] = (uint32_t)x0; v[3] = (uint32_t)x1;
                                        ok &= (uint32_t)(x0 >> 32) == prev && (uint32_t)(x1 >> 32) == prev;
                                    }
                                    if (__all(ok)) break;
                                    if (spins >= XP_SPIN_MAX) { if (lane == 0) xp_fail(p, 4u); if (RES) xl_die(s_dead); break; }
                                    if ((spins & (RES ? 255u : 1023u)) == (RES ? 255u : 1023u) && (RES ? xl_err(p) : __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u)) { if (RES) xl_die(s_dead); break; }
                                }
                                if (a0) { bv = __uint_as_float(v[0]); bi = (int)v[1]; }
                                if (a1) {
                                    const float ov = __uint_as_float(v[2]);
                                    const int oi = (int)v[3];
                                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                                }
                            } else {
                                for (int k = tid; k < XPK(nparts); k += NT) {
                                    const float v = XPK(pmax_val)[k];
                                    const int ix = XPK(pmax_idx)[k];
                                    if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
                                }
                            }
#pragma unroll
                            for (int off = 32; off > 0; off >>= 1) {
                                const float ov = __shfl_xor(bv, off, 64);
                                const int oi = __shfl_xor(bi, off, 64);
                                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                            }
                            if (lane == 0) { s_redf[wave] = bv; s_redi[wave] = bi; }
                            __syncthreads();
                            xl_live<RES>(s_dead);
                            bv = s_redf[0]; bi = s_redi[0];
#pragma unroll
                            for (int w = 1; w < NW; w++)
                                if (s_redf[w] > bv || (s_redf[w] == bv && s_redi[w] < bi)) { bv = s_redf[w]; bi = s_redi[w]; }
                            if (bi < 0 || bi >= XPK(n_vocab)) bi = 0;
                            return bi;
                        };
                        if (RES && p.resident != 0 && tk == 0 && slot == 0 && tid == 0) { s_spec[0] = (uint32_t)XPK(res_spec0); s_spec[1] = 0xffffffffu; }
                        if (RES && p.resident != 0 && tk > 0) {
                            // resident launch (kernels_xpipe.hip.h, the same protocol): the token of this pass is the one the next biogpt_eval() call posts in the
                            // pinned mailbox -- or, running ahead of a greedy caller, the device's own arg-max, which that post must then confirm.  Workgroup 0
                            // decides and hands the token to the XCD's other workgroups; a wait for the host lasts at most idle_ticks, then the launch ends
                            xp_u64 *const gt = XPK(samp) + 2048;
                            if (slot == 0) {
                                const uint32_t want = XPK(mbox_seq0) + (uint32_t)tk;
                                auto wait_post = [&](uint32_t seq, int np, int &spec) __attribute__((always_inline)) -> int {
                                    const xp_u64 *mb = reinterpret_cast<const xp_u64 *>(XPK(mbox)) + (size_t)(seq & 63u) * 4;
                                    const unsigned long long t0 = wall_clock64();
                                    for (;;) {
                                        const xp_u64 w = __hip_atomic_load(mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                        if ((uint32_t)(w >> 40) == (seq & 0xffffffu)) {
                                            const int tv = (int)(w & 0xffffffu), pn = (int)((w >> 24) & 0x1fffu);
                                            spec = (int)((w >> 37) & 1u);
                                            return (pn == np && tv < XPK(n_vocab)) ? tv : -1;      // anything else is the host's request to leave
                                        }
                                        if (wall_clock64() - t0 > (unsigned long long)XPK(idle_ticks)) return -1;
                                        if (__hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) return -1;
                                        __builtin_amdgcn_s_sleep(4);
                                    }
                                };
                                // a pass that was started unasked: the host's post for it is read NOW, while this workgroup waits for the pass to end anyway
                                if (tid == 0) {
                                    const int pending = (int)s_spec[1];
                                    int unused = 0;
                                    s_spec[2] = (pending < 0 || wait_post(want - 1u, n_past - 1, unused) == pending) ? 1u : 0u;
                                }
                                const int guess = sample_prev();       // arg-max of the previous token's logits (barriers inside: the whole workgroup)
                                __syncthreads();                       // s_redf is reused by the helper duty
                                if (tid == 0) {
                                    int spec = (int)s_spec[0];
                                    const bool ahead = (int)s_spec[1] >= 0 || spec != 0;
                                    int got = -1;
                                    if (s_spec[2] != 0u) {
                                        if (XPK(spec_rec)) __hip_atomic_store(XPK(spec_rec), ((unsigned long long)want << 32) | (unsigned long long)(uint32_t)guess, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                        got = ahead ? guess : wait_post(want, n_past, spec);
                                    }
                                    if (got >= 0) {
                                        xp_put_local(gt, epoch, (uint32_t)got);
                                        xp_put(gt + 1, epoch, 1u);      // the lm_head workgroups write this pass's rows only behind this word
                                        s_spec[0] = (uint32_t)spec; s_spec[1] = ahead ? (uint32_t)guess : 0xffffffffu;
                                    } else {
                                        // leaving: the two words the launch's end would have written (the workgroups below end inside their polls), then the quit word
                                        __hip_atomic_store(p.ctl, epoch0 + (uint32_t)p.n_tok, XP_RLX);
                                        __hip_atomic_store(p.ctl + 2, launch0 + 1u, XP_RLX);
                                        xp_quit(p);
                                    }
                                }
                            }
                            uint32_t v[1];
                            xl_sweep<RES, 1>(gt, lane == 0, epoch, v, p, s_dead);
                            tok = __builtin_amdgcn_readfirstlane((int)v[0]);
                            if (tok < 0 || tok >= XPK(n_vocab)) tok = 0;
                        } else if (tk > 0 || XPK(tok_src) == 2) {
                            tok = sample_prev();
                            if (slot == 0 && tid == 0) {
                                int32_t *tokens = state_tokens(p.st);
                                if (n_gen < XPK(n_positions)) tokens[XPK(n_positions) + n_gen] = tok;
                                tokens[0] = tok;
                            }
                            __syncthreads();       // s_redf is reused by the helper duty
                        } else {
                            tok = (RES && p.resident != 0) ? XPK(res_tok0) : state_tokens(p.st)[0];
                        }
                        if (worker) {      // biogpt.cpp:664-686: embed_tokens[tok] * sqrt(D) + embed_positions[n_past + 2]
                            float e[4];
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                e[j] = __fadd_rn(__fmul_rn(dequant_elem(XPK_MATRIX(tok_emb), tok, 4 * tid + j), XPK(embed_scale)), dequant_elem(XPK_MATRIX(pos_emb), n_past + 2, 4 * tid + j));
                            xv = make_float4(e[0], e[1], e[2], e[3]);
                        }
                    } else if (wave < 4) {
                        uint32_t v[4];
                        xl_sweep<RES, 4, 256, true>(XPL(L - 1).gx + tid, true, epoch, v, p, s_dead);
                        xv = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
                    }
                    if (worker) reinterpret_cast<float4 *>(s_x)[tid] = xv;      // residual of stage C
                    XL_WALL(0);
                    if constexpr (ROLE == 0) {
                        // ================= stage A: LayerNorm -> Q8 -> the 192 q / k / v rows of head `slot`, published for the head's helpers =================
                        float4 lnw = xv, lnb = xv;
                        if (worker) { lnw = reinterpret_cast<const float4 *>(s_ln)[tid]; lnb = reinterpret_cast<const float4 *>(s_ln + 1024)[tid]; }
                        ln4_q8_1024<TI::q81, TI::q81>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
                        xl_live<RES>(s_dead);
                        XL_WALL(6);
                        uint32_t ax[8];
                        const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
                        ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
                        const float axd = s_xd[sub];
                        const uint32_t axs = s_xs[sub];
                        float *const part = s_part + wave * 2 * QS * DEC_PS;
                        // the q rows first (units 0-3 = rows 0-63 of the head): all 16 helpers of the head wait for them, the k / v rows are needed by ONE helper and later
#pragma unroll
                        for (int s = 0; s < 4; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, true>(wqkv[s], ax, axd, __uint_as_float(axs), (int)axs);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (lane < 8) {
                            const int jj = (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);      // < 64
                            const float v = __fmul_rn(__fadd_rn(s_bias[jj], sum32_in_order(part + lane * DEC_PS)), p.q_scale);      // Q scaled AFTER the bias (biogpt.cpp:708-710)
                            xp_put(G + XP_G_QKV + slot * 64 + jj, epoch, __float_as_uint(v));
                        }
#pragma unroll
                        for (int s = 4; s < QS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, true>(wqkv[s], ax, axd, __uint_as_float(axs), (int)axs);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (lane >= 8 && lane < 2 * QS) {
                            const int jj = (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);
                            const float v = __fadd_rn(s_bias[jj], sum32_in_order(part + lane * DEC_PS));
                            xp_put(G + XP_G_QKV + (jj >> 6) * 1024 + slot * 64 + (jj & 63), epoch, __float_as_uint(v));
                        }
                        XL_WALL(1);
                    }
                }
            }
        };
        auto stage_post = [&](const int L) __attribute__((always_inline)) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63, wave = RES ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
            const int sub = lane & 31, rsub = lane >> 5;
            constexpr bool own_first = FIRST;
            xp_u64 *const G = p.gran + (size_t)L * XP_G_LAYER;
            if constexpr (FIRST) {
                {
                    if constexpr (ROLE == 0) combine(L, epoch, n_past + 1, slot);      // this workgroup's head: partials of its 16 key ranges -> attention output, in-XCD
                    // ================= stage C: out_proj + bias + residual (biogpt.cpp:767-772) =================
                    if (wave < 5) {
                        uint32_t v[1];
                        xl_sweep<RES, 1, 1, false>(G + XP_G_ATT + tid, TI::q81 || tid < 288, epoch, v, p, s_dead);      // (the block sums travel only with Q8_1 activations)
                        if (tid < 256) s_xq[tid] = v[0];
                        else if (tid < 288) s_xd[tid - 256] = __uint_as_float(v[0]);
                        else s_xs[tid - 288] = v[0];
                    }
                    __syncthreads();
                    xl_live<RES>(s_dead);
                    XL_WALL(8);
                    {
                        uint32_t ax[8];
                        const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
                        ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
                        const float axd = s_xd[sub];
                        const uint32_t axs = s_xs[sub];
                        float *const part = s_part + wave * 2 * OS * DEC_PS;
#pragma unroll
                        for (int s = 0; s < OS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, true>(wo[s], ax, axd, __uint_as_float(axs), (int)axs);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (lane < 2 * OS) {
                            const int lr = (lane >> 1) * 2 * NW + wave * 2 + (lane & 1), row = slot * 32 + lr;
                            const float v = __fadd_rn(__fadd_rn(sum32_in_order(part + lane * DEC_PS), s_bias[192 + lr]), s_x[row]);
                            xp_put(XPL(L).gx1 + xp_col_slot(row), epoch, __float_as_uint(v));       // the MLP half runs on the next XCD
                        }
                    }
                    XL_WALL(3);
                    if (L + 1 < n_layer) fetch_kv(L + 1, n_past);      // deferred from the helper duty (see there)
                    else if (more) fetch_kv(0, n_past + 1);
                    __syncthreads();       // s_ln / s_bias are rewritten by the next unit's load
                }
            }
            if constexpr (SECOND) {
                {
                    // ================= stage D: LayerNorm -> Q8 -> fc1 -> GELU -> Q8 (biogpt.cpp:777-787) =================
                    float4 x1v = make_float4(0.f, 0.f, 0.f, 0.f), lnw = x1v, lnb = x1v;
                    if (wave < 4) {
                        uint32_t v[4];
                        xl_sweep<RES, 4, 256, true>(XPL(L).gx1 + tid, true, epoch, v, p, s_dead);
                        x1v = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
                        reinterpret_cast<float4 *>(s_x1)[tid] = x1v;
                        XL_WALL2(9);
                        lnw = reinterpret_cast<const float4 *>(s_ln)[tid]; lnb = reinterpret_cast<const float4 *>(s_ln + 1024)[tid];
                    }
                    ln4_q8_1024<TI::q81, TI::q81>(x1v, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
                    xl_live<RES>(s_dead);
                    XL_WALL2(10);
                    {
                        uint32_t ax[8];
                        const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
                        ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
                        const float axd = s_xd[sub];
                        const uint32_t axs = s_xs[sub];
                        float *const part = s_part + wave * 2 * FS * DEC_PS;
#pragma unroll
                        for (int s = 0; s < FS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, true>(w1[s], ax, axd, __uint_as_float(axs), (int)axs);
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
                    XL_WALL2(11);
                    if (tid < 128) {
                        int8_t q8; float d8; uint32_t s8;
                        q8_block32(s_g[tid], TI::q81, q8, d8, s8, TI::q81);
                        const uint32_t packed = xp_pack4(q8);
                        const int blk = slot * 4 + (tid >> 5);
                        if ((tid & 3) == 0) xp_put_local(G + XP_G_H + slot * 32 + (tid >> 2), epoch, packed);
                        if ((tid & 31) == 0) { xp_put_local(G + XP_G_H + 1024 + blk, epoch, __float_as_uint(d8)); if (TI::q81) xp_put_local(G + XP_G_H + 1152 + blk, epoch, s8); }
                    }
                    XL_WALL2(4);
                    // ================= stage E: fc2 + bias + residual (biogpt.cpp:790-795) =================
                    {
                        constexpr int NQ = 1024 / NT;
                        uint32_t v[NQ + 1];
                        const xp_u64 *g = G + XP_G_H + tid;
                        const bool tail = tid < (TI::q81 ? 256 : 128);      // scales, and with Q8_1 activations the block sums
                        for (uint32_t spins = 0;; spins++) {
                            bool ok = true;
#pragma unroll
                            for (int k = 0; k < NQ; k++) {
                                const xp_u64 a = __hip_atomic_load(g + k * NT, XP_RLX);
                                v[k] = (uint32_t)a;
                                ok &= (uint32_t)(a >> 32) == epoch;
                            }
                            if (tail) {
                                const xp_u64 a = __hip_atomic_load(g + 1024, XP_RLX);
                                v[NQ] = (uint32_t)a;
                                ok &= (uint32_t)(a >> 32) == epoch;
                            }
                            if (__all(ok)) break;
                            if (spins >= XP_SPIN_MAX) { if (lane == 0) xp_fail(p, 1u); if (RES) xl_die(s_dead); break; }
                            if ((spins & (RES ? 255u : 1023u)) == (RES ? 255u : 1023u) && (RES ? xl_err(p) : __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u)) { if (RES) xl_die(s_dead); break; }
                        }
#pragma unroll
                        for (int k = 0; k < NQ; k++) s_hq[tid + k * NT] = v[k];
                        if (tid < 128) s_hd[tid] = __uint_as_float(v[NQ]);
                        else if (tid < 256) s_hs[tid - 128] = v[NQ];
                    }
                    __syncthreads();
                    xl_live<RES>(s_dead);
                    XL_WALL2(12);
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
                            for (int r = 0; r < F2R; r++) part[r * DEC_PS2 + u] = xp_dot<WT, true>(w2[r][it], ax, axd, __uint_as_float(axs), (int)axs);
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
                            xp_put(XPL(L).gx + xp_col_slot(row), epoch, __float_as_uint(v));
                            if (L == n_layer - 1) XPK(x_final)[row] = v;
                        }
                    }
                    XL_WALL2(5);
                    if (L + 1 < n_layer) fetch_kv(L + 1, n_past);      // deferred from the helper duty: in front of the x1 poll it cost that poll 0.16 us
                    else if (more) fetch_kv(0, n_past + 1);
                    __syncthreads();       // s_ln / s_bias / s_x1 are rewritten by the next unit's load
                }
            }
        };

        // ---- walk 1: layers before this XCD's first own one (first token of the launch only: its first weight load is issued one layer ahead of
        //      its turn, not all eight XCDs' at kernel start -- 57 MB at once would delay layer 0) ----
        const int l_pre = (my_last < 0) ? n_layer : ((tk == 0) ? (my_l0 > 0 ? my_l0 - 1 : 0) : 0);
        for (int L = 0; L < l_pre; L++) helper(L, epoch, n_past, false, more);
        if (my_last >= 0) {
            load_unit_weights(my_l0, -1);
            for (int L = l_pre; L < my_l0; L++) helper(L, epoch, n_past, false, more);
            // ---- walk 2: own layer, then the three layers up to the next own one -- behind each of their helper duties a third of that layer's weights ----
            for (int Lo = my_l0; Lo <= my_last; Lo += 4) {
                if (FIRST) stage_pre(Lo);
                helper(Lo, epoch, n_past, FIRST, more, SECOND);      // the MLP half fetches its K / V rows behind stage E (its x1 poll comes first)
                stage_post(Lo);
                if (Lo < my_last) {
                    helper(Lo + 1, epoch, n_past, false, more);
                    load_unit_weights(Lo + 4, 0);
                    helper(Lo + 2, epoch, n_past, false, more);
                    load_unit_weights(Lo + 4, 1);
                    helper(Lo + 3, epoch, n_past, false, more);
                    load_unit_weights(Lo + 4, 2);
                }
            }
        }
        // ---- final LayerNorm + lm_head (biogpt.cpp:799-811) on the XCDs that are done: four 64-row blocks per workgroup, loaded now, used when the
        //      last layer's output arrives; the layers in between only need this workgroup's helper duty ----
        if (!lm_mine) {
            // ---- walk 3: the layers after the last own one ----
            for (int L = (my_last < 0 ? n_layer : my_last + 1); L < n_layer; L++) helper(L, epoch, n_past, false, more);
        } else {
            Unit<WT> wl[LMS];
            const int row0 = lm_rank * 256;
            // the lm_head rows in thirds, behind the (up to three) helper duties that are left -- like the unit weights, never right behind the own stages
            auto load_lm = [&](const int s0, const int s1) __attribute__((always_inline)) {
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                const int lane = tid & 63, wave = RES ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6), sub = lane & 31, rsub = lane >> 5;
#pragma unroll
                for (int s = 0; s < LMS; s++) {
                    if (s < s0 || s >= s1) continue;
                    const int row = row0 + s * 2 * NW + wave * 2 + rsub;
                    if (row < XPK(n_vocab)) load_unit<WT>(wl[s], XPK_MATRIX(Wlm), (int64_t)row * 32 + sub);
                    else { wl[s].q0 = make_uint4(0u, 0u, 0u, 0u); wl[s].q1 = wl[s].q0; wl[s].sc = 0u; wl[s].qh = 0u; }
                }
#pragma unroll
                for (int s = 0; s < LMS; s++)
                    if (s >= s0 && s < s1) xp_settle<WT, true>(wl[s]);
            };
            // ---- walk 3 with the lm_head rows arriving ----
            {
                int L = (my_last < 0 ? n_layer : my_last + 1);
                for (; L + 3 < n_layer; L++) helper(L, epoch, n_past, false, more);
                if (L < n_layer) { helper(L, epoch, n_past, false, more); L++; }
                load_lm(0, 6);
                if (L < n_layer) { helper(L, epoch, n_past, false, more); L++; }
                load_lm(6, 11);
                if (L < n_layer) { helper(L, epoch, n_past, false, more); L++; }
                load_lm(11, LMS);
            }
            {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63, wave = RES ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6), sub = lane & 31, rsub = lane >> 5;
            const bool worker = tid < 256;
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), lnw = xv, lnb = xv;
            if (worker) { lnw = reinterpret_cast<const float4 *>(XPK(lm_ln_w))[tid]; lnb = reinterpret_cast<const float4 *>(XPK(lm_ln_b))[tid]; }
            // a resident pass with an odd sequence number writes the alternate row / partial buffers (XpParams::spec_rec)
            const bool alt = RES && p.resident != 0 && ((XPK(mbox_seq0) + (uint32_t)tk) & 1u) != 0u;
            float *const lg_dev = alt ? XPK(logits_alt) : XPK(logits);
            float *const lg_host = alt ? XPK(logits_host_alt) : XPK(logits_host);
            if (RES && p.resident != 0 && tk > 0) {      // "the rows of this pass may be written" (published by XCD 0 at the start of the pass, long ago: one poll)
                uint32_t go[1];
                xl_sweep<RES, 1>(XPK(samp) + 2049, lane == 0, epoch, go, p, s_dead);
            }
            if (wave < 4) {
                uint32_t v[4];
                xl_sweep<RES, 4, 256, true>(XPL(n_layer - 1).gx + tid, true, epoch, v, p, s_dead);
                xv = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
            ln4_q8_1024<TI::q81, TI::q81>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
            xl_live<RES>(s_dead);
            uint32_t ax[8];
            const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
            ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
            const float axd = s_xd[sub];
            const uint32_t axs = s_xs[sub];
            float *const part = s_part + wave * 2 * LMS * DEC_PS;
#pragma unroll
            for (int s = 0; s < LMS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, true>(wl[s], ax, axd, __uint_as_float(axs), (int)axs);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float best_val = -INFINITY;
            int best_idx = 0x7fffffff;
            if (lane < 2 * LMS) {
                const int row = row0 + (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);
                if (row < XPK(n_vocab)) {
                    const float v = sum32_in_order(part + lane * DEC_PS);
                    if constexpr (RES) s_S[row - row0] = v;        // staged for the copies below (s_S: this workgroup's helper duties of the token are over)
                    else { XPK(logits)[row] = v; if (XPK(logits_host)) XPK(logits_host)[row] = v; }
                    best_val = v; best_idx = row;
                }
            }
            // per-block partial arg-max (lowest index wins ties): groups of 64 / NW lanes, then the NW waves through LDS
#pragma unroll
            for (int off = 1; off < 64 / NW; off <<= 1) {
                const float ov = __shfl_xor(best_val, off, 64);
                const int oi = __shfl_xor(best_idx, off, 64);
                if (ov > best_val || (ov == best_val && oi < best_idx)) { best_val = ov; best_idx = oi; }
            }
            constexpr int LPB = 64 / NW;
            if (lane < 2 * LMS && (lane & (LPB - 1)) == 0) { s_redf[(lane / LPB) * NW + wave] = best_val; s_redi[(lane / LPB) * NW + wave] = best_idx; }
            __syncthreads();
            if (RES && wave == 1) {
                // the workgroup's 256 logits as ONE kilobyte of 16-byte stores each: the host's copy (write-through, pinned memory) and the device row; behind
                // its own stores the completion word of this token (kernels_xpipe.hip.h, the same lines)
                const int r = row0 + 4 * lane;
                if (r + 3 < XPK(n_vocab)) {
                    const xp_v4f v4 = *reinterpret_cast<const xp_v4f *>(s_S + 4 * lane);
                    if (lg_host) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(lg_host + r), "v"(v4) : "memory");
                    *reinterpret_cast<xp_v4f *>(lg_dev + r) = v4;
                } else {
                    for (int j = r; j < XPK(n_vocab); j++) { if (lg_host) __hip_atomic_store(lg_host + j, s_S[j - row0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); lg_dev[j] = s_S[j - row0]; }
                }
                if (p.resident != 0) {
                    if (lg_host && lane < 4 && lm_rank * 4 + lane < XPK(lm_blocks)) {      // the block maxima behind the row (kernels_xpipe.hip.h, XpParams::logits_host)
                        float bm = s_redf[lane * NW];
#pragma unroll
                        for (int w = 1; w < NW; w++) bm = fmaxf(bm, s_redf[lane * NW + w]);
                        __hip_atomic_store(lg_host + xp_blockmax_offset(XPK(n_vocab)) + lm_rank * 4 + lane, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(XPK(done_host) + lm_rank, XPK(mbox_seq0) + (uint32_t)tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            if (tid < 4) {
                float bv = s_redf[tid * NW];
                int bi = s_redi[tid * NW];
#pragma unroll
                for (int w = 1; w < NW; w++) {
                    const float ov = s_redf[tid * NW + w];
                    const int oi = s_redi[tid * NW + w];
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                const int blk = lm_rank * 4 + tid;
                if (blk < XPK(lm_blocks)) {
                    if (RES && alt) { XPK(pmax_alt_val)[blk] = bv; XPK(pmax_alt_idx)[blk] = bi; }
                    else { XPK(pmax_out_val)[blk] = bv; XPK(pmax_out_idx)[blk] = bi; }
                    if (more) {        // the sampler of the next token runs on XCD 0
                        xp_put(XPK(samp) + blk, epoch, __float_as_uint(bv));
                        xp_put(XPK(samp) + 1024 + blk, epoch, (uint32_t)bi);
                    }
                }
            }
            __syncthreads();       // s_redf / s_part / s_xq are rewritten by this workgroup's next token
            }
        }
    }   // tokens
    if (threadIdx.x == 0) {
        if (xcd == last_xcd && slot == 0) {
            if (epoch0 + (uint32_t)p.n_tok > 0xF0000000u) xp_fail(p, 5u);
            __hip_atomic_store(p.ctl, epoch0 + (uint32_t)p.n_tok, XP_RLX);
            __hip_atomic_store(p.ctl + 2, launch0 + 1u, XP_RLX);      // (a resident launch that leaves early has written the same two words: XCD 0's workgroup 0)
        }
        if (xcd == (last_xcd == 1 ? 2 : 1) && slot == 0 && p.adv != 0) { p.st->n_past = n_past0 + p.n_tok; p.st->n_gen = n_gen0 + p.n_tok; }
    }
}

template <int WT, int KR, bool RES = false>
__global__ __launch_bounds__(512) void dec_xlong_kernel(const XpParams p) {
    constexpr int NT = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *const s_redi = reinterpret_cast<int *>(smem + XP_S_REDF + 256);
    uint16_t *const s_gelu = reinterpret_cast<uint16_t *>(smem + XP_S_TOTAL);
    // XCD and rank inside the XCD: as in dec_xpipe_kernel (HW_REG_XCC_ID + a per-XCD arrival ticket)
    const uint32_t epoch0 = __hip_atomic_load(p.ctl, XP_RLX);
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;
        const uint32_t t = __hip_atomic_fetch_add(p.ctl + 8 + xcc, 1u, XP_RLX);
        const uint32_t launch = __hip_atomic_load(p.ctl + 2, XP_RLX);
        s_redi[0] = (int)xcc;
        s_redi[1] = (int)(t - 32u * (launch - 1u));
        s_redi[2] = (int)launch;
        *reinterpret_cast<uint32_t *>(smem + XP_S_REDD + 96) = 0u;      // xl_run's s_dead
    }
    __syncthreads();
    const int xcd = __builtin_amdgcn_readfirstlane(s_redi[0]), slot = __builtin_amdgcn_readfirstlane(s_redi[1]);
    const uint32_t launch0 = (uint32_t)__builtin_amdgcn_readfirstlane(s_redi[2]);
    __syncthreads();
    if ((unsigned)slot >= 32u) { if (threadIdx.x == 0) xp_fail(p, 2u); return; }
    const int n_past0 = (RES && p.resident != 0) ? p.res_n_past0 : p.st->n_past, n_gen0 = p.st->n_gen;
    if (xcd & 1) {      // the MLP halves look GELU up in LDS
        const uint4 *src = reinterpret_cast<const uint4 *>(p.gelu_tab);
        const int np8 = p.gelu_p / 8, nn8 = p.gelu_n / 8;
        for (int i = threadIdx.x; i < np8; i += NT) reinterpret_cast<uint4 *>(s_gelu)[i] = src[i];
        for (int i = threadIdx.x; i < nn8; i += NT) reinterpret_cast<uint4 *>(s_gelu + p.gelu_p)[i] = src[0x8000 / 8 + i];
        xl_run<WT, KR, 2, RES>(p, smem, xcd, slot, epoch0, n_past0, n_gen0, launch0);
        return;
    }
    if (slot < 16) xl_run<WT, KR, 0, RES>(p, smem, xcd, slot, epoch0, n_past0, n_gen0, launch0);
    else xl_run<WT, KR, 1, RES>(p, smem, xcd, slot, epoch0, n_past0, n_gen0, launch0);
}

}  // namespace bgk
