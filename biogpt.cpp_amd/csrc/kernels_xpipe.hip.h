// The decode step as ONE persistent launch, pipelined over the MI355X's 8 XCDs (accelerator dies: 32 compute units and one
// 4 MB L2 each).  biogpt.cpp:664-811 for all layers + the output projection; the arithmetic per element is that of
// kernels_decode.hip.h.
//
// Why: the five-launch layer of kernels_decode.hip.h spends 7.5 of its 20 us in launch boundaries (1.3-1.6 us each:
// every boundary is a device-wide all-to-all), and each kernel then starts cold (0.7-1.5 us until its first weights
// arrive).  What a layer needs between its five stages is an all-to-all among the workgroups THAT COMPUTE THE LAYER --
// and on this chip 32 compute units behind one L2 can do that in 0.4-0.8 us with tagged 8-byte granules
// (tools/microbench11.hip: all six hand-offs of a layer, model-sized, 4.7 us per layer end to end).
//
// Shape: grid = 256 workgroups x 512 threads, one per compute unit.  A workgroup reads its XCD from HW_REG_XCC_ID and takes
// its rank inside the XCD from a per-XCD arrival ticket (the host probes the device once: 8 XCDs, workgroups dealt evenly).
// A layer is two pipeline units: stages A-C (LayerNorm, q/k/v, attention, out_proj) on the 32 workgroups of an EVEN XCD, stages
// D-E (LayerNorm, fc1, fc2) on those of the next, ODD XCD; unit u runs on XCD u % 8.  While the other XCDs compute their units,
// an XCD loads the weights of ITS next unit into registers (8-16 block units per lane) and unpacks the nibble formats there
// to one byte per weight, so a stage never waits for weights and its chain holds 8 v_dot4 per unit: stage latency = hand-off
// + arithmetic.  (SPLIT = false -- a whole layer per XCD, packed units -- was this kernel's first form: 3.37 k against 3.65 k tok/s;
// the host no longer instantiates it.)
//
//   stage A  (workgroups 16-31 of the XCD) x -- granules from the previous layer's XCD, or the embedding of the sampled token --
//            -> LayerNorm -> Q8 -> all 192 q/k/v rows of head slot - 16; KV append; the rows go to workgroup slot - 16
//   stage B  (workgroups 0-15) attention of head `slot`: old keys / values from the cache (in registers since the XCD's
//            previous layer finished), the new ones from stage A
//   stage C  out_proj rows (32 per workgroup) + bias + residual
//   stage D  LayerNorm -> Q8 -> fc1 rows (128 per workgroup = 4 Q8 blocks) -> GELU table (LDS slice) -> Q8
//   stage E  fc2 rows (32 per workgroup) + bias + residual -> x for the next layer's XCD
//   then     final LayerNorm + lm_head on the XCDs that are done (four 64-row blocks per workgroup, loaded ahead), and -- in a
//            multi-token launch (biogpt_hip_generate_greedy) -- the greedy sampler of the next token on XCD 0: the whole loop
//            over the tokens of a context bucket runs inside ONE launch
//
// Hand-offs: "R2" granules of MI355X_MICROARCH.md -- one naturally aligned 8-byte {value, tag} written by ONE relaxed
// agent-scope atomic store and polled with relaxed agent-scope atomic loads; tag = the context's hand-off counter + the token's
// index in the launch (ctl[0], moved on by the last layer's first workgroup when every workgroup has provably read it), buffers are per
// layer, so a tag can only match data of THIS launch.  No fences, no flags.  EVERY spin is bounded: after XP_SPIN_MAX
// passes a poller raises ctl[1] (and the pinned host word), every other poller sees it within 1024 passes, the launch
// drains in milliseconds with garbage outputs, and the host reports the failure and leaves this path for good.
#pragma once

#include "kernels_decode.hip.h"

namespace bgk {

typedef unsigned long long xp_u64;
typedef float xp_v4f __attribute__((ext_vector_type(4)));
#define XP_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr uint32_t XP_SPIN_MAX = 400000u;      // ~0.3 s of polling: far beyond any kernel another stream could hold the compute units with

struct XpLayer {               // one layer's constants (device memory, read with scalar loads)
    const float *ln0_w, *ln0_b, *ln1_w, *ln1_b;
    const float *bqkv, *bo, *b1, *b2;
    DevMatrix Wqkv, Wo, W1, W2;
    float *kcache, *vcache;    // layer slice, head-major [H][P][64]
    // the two hand-offs of the layer that cross XCDs, 1024 granules (8 KB) each: x1 (attention half -> MLP half) and x (the layer's output -> the next layer's XCD).
    // Device memory is interleaved over the two halves of the chip in 8 KB units and a granule costs 0.2 us more per far side (tools/microbench18.hip: XCD a -> b
    // over a line near both 0.40 us, far from both 0.60 us); the host therefore owns two 8 KB-aligned candidates per region and points these at the one its
    // calibration launch found faster for THIS hop (xpipe_place_hops).
    unsigned long long *gx1, *gx;
};

// granules of one layer
constexpr int XP_G_QKV = 0;                    // [3072] stacked q (scaled) / k / v rows; only the rows of workgroups 16-31 are published
constexpr int XP_G_ATT = 3072;                 // [256] 4 x int8, [32] block scale, [32] block sum
constexpr int XP_G_X1 = XP_G_ATT + 320;        // [1024] (unused since round 4: XpLayer::gx1)
constexpr int XP_G_H = XP_G_X1 + 1024;         // [1024] 4 x int8, [128] scale, [128] sum
constexpr int XP_G_X = XP_G_H + 1280;          // [1024] the layer's OUTPUT (unused since round 4: XpLayer::gx)
constexpr int XP_G_LAYER = XP_G_X + 1024;

// The layer table is read through the CONSTANT address space (scalar loads), and what its pointers lead to through the GLOBAL one.  Through generic pointers hipcc cannot
// prove that the kernel's own stores leave the table alone: every `XPL(L).field` was a VECTOR load of the field + s_waitcnt vmcnt(0) in front of its use -- in front of
// every poll sweep of a stage -- and every access through such a pointer a FLAT one, behind which every wait of the wave is a vmcnt(0) (round 6, found in kernels_fpipe.hip.h).
typedef const XpLayer __attribute__((address_space(4))) XpLayerK;
#define XPL(L_) (((const XpLayerK *)p.layers)[L_])
#define XPL_MATRIX(m_) DevMatrix{(m_).qs, (m_).sc, (m_).qh, (m_).type, (m_).M, (m_).K}
typedef const xp_u64 __attribute__((address_space(1))) *xp_gq;      // granules (polls)
typedef xp_u64 __attribute__((address_space(1))) *xp_gw;            // granules (publication)
typedef const float __attribute__((address_space(1))) *xp_gf;
typedef const xp_v4f __attribute__((address_space(1))) *xp_gf4;      // (an ext-vector: HIP's float4 class cannot be copied out of an address space)
__device__ __forceinline__ float4 xp_ldg4(const float *q, int i) { const xp_v4f t = ((xp_gf4)q)[i]; return make_float4(t.x, t.y, t.z, t.w); }

struct XpParams {
    const XpLayer *layers;
    int32_t n_layer;
    xp_u64 *gran;              // [n_layer][XP_G_LAYER], zeroed once at allocation
    xp_u64 *gran_l;            // [n_layer][XL_G_LAYER] (kernels_xlong.hip.h: scores and partial outputs of the key-range helpers), zeroed once; null: no long-context variant
    uint32_t *ctl;             // [0] hand-off tag of the launch's first token (starts at 1, + n_tok per launch), [1] error word, [2] launch counter (starts at 1), [8..15] per-XCD arrival tickets
    uint32_t *err_host;        // pinned mirror of the error word
    DevState *st;
    DevMatrix tok_emb, pos_emb;
    float embed_scale;
    int32_t tok_src;           // 1: token in the state; 2: arg-max of the previous step's lm_head partials
    const float *pmax_val; const int32_t *pmax_idx; int32_t nparts;
    int32_t n_positions, n_vocab;
    float eps, q_scale;
    int32_t P, t_cap;
    const uint16_t *exp_tab, *gelu_tab;
    int32_t gelu_p, gelu_n, gelu_z;   // every workgroup keeps gelu_tab[0 .. gelu_p) and [0x8000 .. 0x8000 + gelu_n) in LDS; above: identity up to
                               //   +inf, below: the constant gelu_z down to the most negative finite value (host-checked); 0 / 0: no slice
    int32_t exp_n;             // the attention workgroups (even XCDs: no GELU slice there) keep exp_tab[0x8000 .. 0x8000 + exp_n) in LDS: every argument of ggml_soft_max's table for
                               //   which the entry is not 0 (host-checked: [0] = 1.0, 0 from 0x8000 + exp_n down to the most negative finite value); 0: no slice
    float *x_final;            // [1024] input of the final LayerNorm + lm_head launch (written also when the lm_head runs in here)
    // final LayerNorm + lm_head inside this launch (lm != 0): the workgroups of the XCDs that do NOT compute the last layer take
    // three 64-row blocks of the output projection each -- the blocks, and the per-block arg-max partials, of the stand-alone
    // lm_head launch (matvec_fast_kernel<EPI_LOGITS>, 64 rows per workgroup), so every consumer of the partials is unchanged
    int32_t lm, lm_blocks, adv;
    int32_t n_tok;             // tokens in this launch (>= 1; > 1 only with lm != 0 and adv != 0: the device-resident greedy loop)
    xp_u64 *samp;              // [2][1024] granules: per-block arg-max partials {value, index} of the previous token of this launch
    DevMatrix Wlm;
    const float *lm_ln_w, *lm_ln_b;
    float *logits;
    float *logits_host;        // optional pinned host copy of the row (biogpt_eval's output): written by the same lanes, no copy node behind the launch.  A resident
                               //   launch also leaves the maxima of its 64-row blocks behind the row, at [xp_blockmax_offset(n_vocab) + block]: the host's top-k selection
                               //   (biogpt_hip_eval_topk, a caller that samples) then looks at the k blocks that can hold a candidate instead of at 42 k logits
    float *pmax_out_val; int32_t *pmax_out_idx;
    // resident mode (biogpt_hip_eval, one token per API call): the launch stays on the device after its first token; token tk >= 1 is taken from the
    // pinned mailbox slot (mbox_seq0 + tk) % 64 = {n_past, causal, token, seq} that the NEXT biogpt_eval() call fills -- workgroup 0 of XCD 0 waits for it,
    // at most idle_ticks of the 100 MHz clock -- and every lm_head workgroup reports its share of the host logits row with a word in done_host
    int32_t dual;              // contexts of 257 .. 512 keys (every launch form): 1 = dec_xpipe_kernel's two-workgroups-per-head variant (needs gran_l), 0 = kernels_xlong.hip.h
    int32_t as_res;            // measurement only (BIOGPT_HIP_XPIPE_AS_RES=1, read with the context's options): ordinary launches go through the RES instantiations with resident = 0
    int32_t resident;
    int32_t res_tok0, res_n_past0;   // token 0 of a resident launch and its position
    int32_t res_dbg;                 // measurement only (BIOGPT_HIP_RES_DBG): 1 completion word without waiting for the row stores, 2 no sleep in the mailbox poll, 4 no row store
    const int32_t *mbox;             // 64 slots of 32 bytes, the first 8 of each = xp_post(...)
    uint32_t mbox_seq0;
    uint32_t idle_ticks;
    uint32_t *done_host;       // pinned [lm workgroups]: sequence number of the last token whose rows this workgroup has written to logits_host
    // Speculative continuation (greedy callers): when the post of token tk - 1 carried the "speculate" flag, token tk starts from the device's OWN arg-max of
    // token tk - 1 as soon as those logits exist -- while the host is still reading that row -- and the host's post of token tk, checked when token tk + 1 is due,
    // must name the same token and position (anything else ends the launch).  A resident pass with an odd sequence number writes the *_alt buffers, an even one
    // the ordinary ones, so a pass the host never asked for overwrites nothing the host (or a later launch) still reads.  spec_rec = {sequence number << 32 |
    // arg-max of the previous token}: written at the start of every pass tk >= 1, the host compares its token with it.
    int32_t res_spec0;               // speculate token 1 of this launch
    unsigned long long *spec_rec;    // pinned
    float *logits_alt, *logits_host_alt;
    float *pmax_alt_val; int32_t *pmax_alt_idx;
    unsigned long long *wall;  // profiling (BIOGPT_HIP_PROFILE_HOOKS): [n_layer][16] wall clock of workgroups 0 and 16, then [32][16] of every workgroup of the last layer
};

#ifdef BIOGPT_HIP_PROFILE_HOOKS
#define XP_WALL(k) do { if (p.wall && tid == 0) { const unsigned long long t_ = wall_clock64(); if ((slot & 15) == 0) p.wall[L * 16 + (k)] = t_; \
        if (L == p.n_layer - 1) p.wall[(p.n_layer + slot) * 16 + (k)] = t_; } } while (0)
// the token's tail (round 4, tools/tail_timeline.py): bank tk % 8 of 16 stamps at wall[4096 ..]: 0 the last layer's output published (token tk), 1 .. 4 lm_head
// workgroup 0: that output seen / LayerNorm done / rows done / partials published, 5 .. 6 XCD 0's workgroup 0: token tk sampled / its embedding done
#define XP_TAIL(tok, k) do { if (p.wall && tid == 0) p.wall[4096 + ((tok) & 7) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define XP_WALL(k) do {} while (0)
#define XP_TAIL(tok, k) do {} while (0)
#endif

// Parameters that are read once per token (the sampler's and the lm_head pass's buffers, everything of the resident form) are NOT held in scalar registers over the launch:
// XPK(field) reads the field from the kernel-argument segment where it is used, through a pointer the optimiser cannot look through (otherwise every field is loaded at
// the top of the kernel and lives -- or is spilled to VGPR lanes: 149 / 276 spilled SGPRs in the ordinary / resident 128-key kernels, v_readlane on the stages' critical
// path -- for the whole layer loop).  XpParams is the launch's ONLY argument: offset 0 of the segment.
typedef const XpParams __attribute__((address_space(4))) *XpKernargPtr;
__device__ __forceinline__ XpKernargPtr xp_kernarg() {
    XpKernargPtr k = (XpKernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}
#define XPK(field) (xp_kernarg()->field)
#define XPK_MATRIX(field) DevMatrix{XPK(field.qs), XPK(field.sc), XPK(field.qh), XPK(field.type), XPK(field.M), XPK(field.K)}

// to ANOTHER XCD (the layer output): write-through (sc1) store, visible at the memory side
__device__ __forceinline__ void xp_put(xp_u64 *g, uint32_t epoch, uint32_t v) { __hip_atomic_store((xp_gw)g, ((xp_u64)epoch << 32) | v, XP_RLX); }
// inside the XCD (every other hand-off): a plain 8-byte store keeps the line in the XCD's L2, where the pollers' sc1 loads
// (L1 bypassed, L2 served) find it -- tools/microbench11.hip: 4.7 us per layer for the six hand-offs against 7.0 with
// write-through stores, which drop the line and send every poll of 32 workgroups across the fabric.  Valid only because
// producer and consumers share ONE L2: that is what the XCC_ID check at the top of the kernel establishes.
__device__ __forceinline__ void xp_put_local(xp_u64 *g, uint32_t epoch, uint32_t v) {
    __hip_atomic_store((xp_gw)g, ((xp_u64)epoch << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// x / x1 columns are stored so that LayerNorm worker t (elements 4t .. 4t+3) polls granules t, t + 256, t + 512, t + 768:
// every poll instruction of a wave covers 512 contiguous bytes
__device__ __forceinline__ int xp_col_slot(int row) { return (row & 3) * 256 + (row >> 2); }

__device__ __forceinline__ void xp_fail(const XpParams &p, uint32_t code) {
    __hip_atomic_store(p.ctl + 1, code, XP_RLX);
    __hip_atomic_store(p.err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (The compile-time arms of rounds 3 - 4 that were measured and lost -- 16-byte fc1 granules, integer softmax sums, an in-XCD two-pass sweep, s_sleep between poll
// passes, the resident form's pieces compiled out one by one, sticky / early dead-wave words, scalarised uniform values, a pinned parameter block -- are gone from this
// file; what each one measured is in profiles/xpipe_ab_r4.txt and profiles/res_instantiation_ab_r4c.txt, the source of the arms in the repository's history before round 5.
// What they left behind as the one form of the code: the two cross-XCD sweeps keep TWO poll passes in flight; the head's workgroup of the <= 192-key variants computes its
// own 64 q rows while workgroup 16 + head hands over only the token's k / v rows; the <= 64-key softmax exchanges the scores once through LDS; the softmax's exp slice sits
// in LDS beside the GELU slices; the XCD of the last-but-one unit takes no lm_head rows.)
template <int N, int S = 1>
__device__ __forceinline__ void xp_sweep(const xp_u64 *g, bool active, uint32_t epoch, uint32_t (&v)[N], const XpParams &p);
template <int N, int S>
__device__ __forceinline__ void xp_sweep_pipelined(const xp_u64 *g, bool active, uint32_t epoch, uint32_t (&v)[N], const XpParams &p);

// Which 256 rows (four 64-row blocks) of the output projection does workgroup (xcd, slot) take inside the pipelined launch ?  -1: none.  Not XCD 0's attention
// workgroups (the next token's layer 0), not the last unit's XCD, and (XP_LM_LATE_XCD) not the XCD of the unit before it; the other XCDs in order, then
// workgroups 16 .. 31 of XCD 0.  The host checks that the ranks cover the blocks (xpipe_lm_capacity).
__host__ __device__ inline int xp_lm_rank(int xcd, int slot, int n_units) {
    const int last_xcd = (n_units - 1) & 7, late_xcd = (n_units - 2) & 7;
    int n_full = 0, before = 0;
    for (int x = 1; x < 8; x++) {
        if (x == last_xcd || x == late_xcd) continue;
        if (x < xcd) before++;
        n_full++;
    }
    if (xcd == 0) return (slot >= 16) ? 32 * n_full + (slot - 16) : -1;
    if (xcd == last_xcd || xcd == late_xcd) return -1;
    return 32 * before + slot;
}
__host__ __device__ inline int xp_lm_capacity(int n_units) {      // workgroups that can take rows
    int n = 0;
    for (int x = 0; x < 8; x++)
        for (int sl = 0; sl < 32; sl++) n += xp_lm_rank(x, sl, n_units) >= 0 ? 1 : 0;
    return n;
}
// where the block maxima of a resident launch's row start in the pinned row buffers (floats; the buffers hold xp_blockmax_offset + 1024 floats)
__host__ __device__ inline int xp_blockmax_offset(int n_vocab) { return (n_vocab + 1023) & ~1023; }
// a post of the host in the resident launch's mailbox: one word, one PCIe read.  token 24 bits (0xffffff: leave), position 13 bits, speculate-next 1 bit, sequence number 24 bits
__host__ __device__ inline xp_u64 xp_post(uint32_t seq, int n_past, int token, int spec) {
    return ((xp_u64)(seq & 0xffffffu) << 40) | ((xp_u64)(spec & 1) << 37) | ((xp_u64)((uint32_t)n_past & 0x1fffu) << 24) | (xp_u64)((uint32_t)token & 0xffffffu);
}
// a clean end of a resident launch (no token from the host within the idle time, or the host asked for it): the same drain as a failure, its own code
constexpr uint32_t XP_QUIT = 0x80000000u;
__device__ __forceinline__ void xp_quit(const XpParams &p) {
    uint32_t expected = 0u;
    if (__hip_atomic_compare_exchange_strong(p.ctl + 1, &expected, XP_QUIT, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        __hip_atomic_store(p.err_host, XP_QUIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr uint32_t XP_RES_CADENCE = 255u;      // poll passes between two looks at the error word in the resident sweeps
// Sweep with a publishing tag: etag is the tag this wave publishes with (the token's epoch), or 0 once the wave has seen the error / quit word -- from
// then on it polls nothing and publishes only tag 0, which no poller accepts: a draining launch can never hand valid-looking garbage downstream (the
// host may be waiting for exactly that token's completion words).
template <bool RES, int N, int S = 1, bool CROSS = false>
__device__ __forceinline__ void xp_sweep_q(const xp_u64 *g, bool active, uint32_t epoch, uint32_t (&v)[N], const XpParams &p, uint32_t &etag) {
    if constexpr (!RES) {       // ordinary launches: the plain sweep (declared below), etag stays the epoch
        if constexpr (CROSS) xp_sweep_pipelined<N, S>(g, active, epoch, v, p);
        else xp_sweep<N, S>(g, active, epoch, v, p);
        return;
    }      // ordinary launches: the plain sweep (declared below), etag stays the epoch
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = 0u;
    if (etag == 0u) return;
    if constexpr (CROSS) {      // two passes in flight (xp_sweep_pipelined), with the resident launch's exits
        xp_u64 cur[N];
#pragma unroll
        for (int k = 0; k < N; k++) cur[k] = active ? __hip_atomic_load((xp_gq)g + k * S, XP_RLX) : ((xp_u64)epoch << 32);
        for (uint32_t spins = 0;; spins++) {
            xp_u64 nxt[N];
#pragma unroll
            for (int k = 0; k < N; k++) nxt[k] = active ? __hip_atomic_load((xp_gq)g + k * S, XP_RLX) : ((xp_u64)epoch << 32);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < N; k++) { ok &= (uint32_t)(cur[k] >> 32) == epoch; }
            if (__all(ok)) {
#pragma unroll
                for (int k = 0; k < N; k++) v[k] = active ? (uint32_t)cur[k] : 0u;
                return;
            }
            if ((spins >= XP_SPIN_MAX)) { if ((threadIdx.x & 63) == 0) xp_fail(p, 1u); etag = 0u; return; }
            if (((spins & XP_RES_CADENCE) == XP_RES_CADENCE) && __any(__hip_atomic_load(p.ctl + 1, XP_RLX) != 0u)) { etag = 0u; return; }
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
        if ((spins >= XP_SPIN_MAX)) { if ((threadIdx.x & 63) == 0) xp_fail(p, 1u); etag = 0u; return; }
        if (((spins & XP_RES_CADENCE) == XP_RES_CADENCE) && __any(__hip_atomic_load(p.ctl + 1, XP_RLX) != 0u)) { etag = 0u; return; }
        
    }
}
// two passes in flight: a pass is a round trip to the memory side (0.3 - 0.4 us across XCDs); looked at one after the other, a granule that lands just behind a
// pass's request waits a whole round trip for the next one
template <int N, int S>
__device__ __forceinline__ void xp_sweep_pipelined(const xp_u64 *g, bool active, uint32_t epoch, uint32_t (&v)[N], const XpParams &p) {
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
        if (spins >= XP_SPIN_MAX) { if ((threadIdx.x & 63) == 0) xp_fail(p, 1u); return; }
        if ((spins & 1023u) == 1023u && __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) return;
#pragma unroll
        for (int k = 0; k < N; k++) cur[k] = nxt[k];
    }
}
// every ACTIVE lane polls its N granules (stride S) until all their tags carry this launch's counter; wave-uniform exit
template <int N, int S>
__device__ __forceinline__ void xp_sweep(const xp_u64 *g, bool active, uint32_t epoch, uint32_t (&v)[N], const XpParams &p) {
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
        if (spins >= XP_SPIN_MAX) { if ((threadIdx.x & 63) == 0) xp_fail(p, 1u); return; }
        if ((spins & 1023u) == 1023u && __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) return;
        
    }
}

// K / V cache rows inside a persistent launch.  In the 192- / 256-key variants workgroup 16 + h appends the rows that workgroup h -- another compute unit of the
// XCD -- reads in later tokens of the same launch, and rows >= T are requested every token before they exist, so a stale line of them may sit in the reader's
// L1.  There (SC1 = true) the loads carry agent scope (sc1: the L1 is not consulted, the XCD's L2 -- where every store of the launch lands -- answers), which is
// what a relaxed agent-scope atomic load compiles to; as buffer loads, so that 16 bytes stay one instruction and the compiler still counts them (vmcnt).  `nt`
// (__builtin_nontemporal_load) is only a replacement hint: an L1 hit on a stale line stays possible.  Where the SAME workgroup appends and re-reads its rows
// (up to 128 keys: the head computes its own q / k / v rows; kernels_xlong.hip.h: the helper that owns the key range appends) the L1 is the writer's own and
// follows its stores (workgroup-scope coherence needs no invalidate on this target outside threadgroup-split mode): streaming loads, SC1 = false.
// Measured (profiles/ab_kv_sc1_r4.txt): sc1 everywhere costs the single-token launch 1 % at 104 keys (280.6 -> 283.3 us) and the long-context launch 6 %
// (366 -> 389 us at 1024 keys: 8 MB of K / V per layer); in the 192- / 256-key variants 0.9 %.
typedef uint32_t xp_v4u __attribute__((ext_vector_type(4)));
constexpr int XP_CPOL_SC1 = 16;                   // gfx940+ cache-policy bits of the buffer intrinsics: 1 = sc0, 2 = nt, 16 = sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xp_kv_rsrc(const float *base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, bytes, 0x00027000);      // raw buffer, 32-bit data format, offsets beyond `bytes` read 0
}
template <bool SC1>
__device__ __forceinline__ float4 xp_kv_load4(__amdgpu_buffer_rsrc_t r, const float *base, int elem) {
    if constexpr (SC1) {
        const xp_v4u t = __builtin_amdgcn_raw_buffer_load_b128(r, elem * 4, 0, XP_CPOL_SC1);
        return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
    } else {
        const xp_v4f t4 = __builtin_nontemporal_load((const __attribute__((address_space(1))) xp_v4f *)(base + elem));
        return make_float4(t4.x, t4.y, t4.z, t4.w);
    }
}
template <bool SC1>
__device__ __forceinline__ float xp_kv_load1(__amdgpu_buffer_rsrc_t r, const float *base, int elem) {
    if constexpr (SC1) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, elem * 4, 0, XP_CPOL_SC1));
    else return __builtin_nontemporal_load((xp_gf)(base + elem));
}

// arg-max steps without the LDS crossbar (round 4: __shfl_xor is a ds_bpermute, ~120 cycles per step; a DPP move is one VALU instruction): the larger value wins,
// equal values: the lower index (std::max_element's rule)
template <int CTRL>
__device__ __forceinline__ void xp_argmax_dpp(float &bv, int &bi) {
    const float ov = dpp_f<CTRL>(bv);
    const int oi = dpp_i<CTRL>(bi);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}
// ... over the whole wave: four DPP steps inside the 16-lane rows, then the four row results through readlane; the result is uniform
__device__ __forceinline__ void xp_argmax_wave(float &bv, int &bi) {
    xp_argmax_dpp<DPP_QUAD_XOR1>(bv, bi); xp_argmax_dpp<DPP_QUAD_XOR2>(bv, bi); xp_argmax_dpp<DPP_ROW_HALF_MIRROR>(bv, bi); xp_argmax_dpp<DPP_ROW_MIRROR>(bv, bi);
    float rv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bv), 0));
    int ri = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const float ov = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bv), r));
        const int oi = __builtin_amdgcn_readlane(bi, r);
        if (ov > rv || (ov == rv && oi < ri)) { rv = ov; ri = oi; }
    }
    bv = rv; bi = ri;
}
// Four consecutive elements (4 t .. 4 t + 3, inside one 32-element block) of row `row` of a block-quantized matrix of type WT, dequantized as dequant_elem does
// (get_rows) -- with ONE 4-byte load of the quants (+ the scale, + Q5's fifth bits) instead of four byte loads and four scale loads: the embedding of
// the sampled token is a dependent load on every token's chain (round 4, tools/tail_timeline.py: 3.1 us of a 251 us token with the generic form).
template <int WT>
__device__ __forceinline__ void xp_row4_request(const DevMatrix &m, int row, int t, uint32_t &q, uint32_t &sc, uint32_t &qh) {
    using TI = TypeInfo<WT>;
    const int64_t blk = (int64_t)row * (m.K / QK) + (t >> 3);
    const int j = (4 * t) & 31;
    if (WT == W_Q8_0) q = *reinterpret_cast<const uint32_t *>(m.qs + blk * 32 + j);
    else q = *reinterpret_cast<const uint32_t *>(m.qs + blk * 16 + (j & 15));
    sc = TI::q81 ? reinterpret_cast<const uint32_t *>(m.sc)[blk] : (uint32_t)reinterpret_cast<const uint16_t *>(m.sc)[blk];
    qh = (WT == W_Q5_0 || WT == W_Q5_1) ? m.qh[blk] : 0u;
}
template <int WT>
__device__ __forceinline__ void xp_row4_values(uint32_t q, uint32_t sc, uint32_t qh, int t, float (&e)[4]) {
    const int j = (4 * t) & 31;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (WT == W_Q8_0) { e[i] = __fmul_rn((float)(int8_t)((q >> (8 * i)) & 0xFFu), h2f((uint16_t)sc)); continue; }
        const uint32_t byte = (q >> (8 * i)) & 0xFFu;
        int v = (j < 16) ? (int)(byte & 0x0Fu) : (int)(byte >> 4);
        if (WT == W_Q5_0 || WT == W_Q5_1) v |= (int)((qh >> (j + i)) & 1u) << 4;
        if (WT == W_Q4_0) e[i] = __fmul_rn((float)(v - 8), h2f((uint16_t)sc));
        else if (WT == W_Q5_0) e[i] = __fmul_rn((float)(v - 16), h2f((uint16_t)sc));
        else e[i] = __fadd_rn(__fmul_rn((float)v, h2f((uint16_t)(sc & 0xFFFFu))), h2f((uint16_t)(sc >> 16)));
    }
}

// a lane's 4 x int8 and the three lanes above it packed into one word (valid in lanes with lane % 4 == 0)
__device__ __forceinline__ uint32_t xp_pack4(int8_t q) {
    const int b = (int)(uint8_t)q;
    const int b1 = __builtin_amdgcn_update_dpp(0, b, 0x101, 0xf, 0xf, true);   // row_shl:1
    const int b2 = __builtin_amdgcn_update_dpp(0, b, 0x102, 0xf, 0xf, true);
    const int b3 = __builtin_amdgcn_update_dpp(0, b, 0x103, 0xf, 0xf, true);
    return (uint32_t)b | ((uint32_t)b1 << 8) | ((uint32_t)b2 << 16) | ((uint32_t)b3 << 24);
}

// granules of one layer in the long-context buffer (XpParams::gran_l): kernels_xlong.hip.h, and the 512-key variant here (DUAL)
constexpr int XL_G_SC = 0;                     // [16 heads][1024 keys] scores
constexpr int XL_G_PV = 16 * 1024;             // [16 heads][16 ranges][64 lo + 64 hi] partial sum_j V_jd p_j (double)
constexpr int XL_G_LAYER = XL_G_PV + 16 * 16 * 128;

// LDS carve (bytes)
constexpr int XP_S_X = 0;                        // [1024] f32 layer input (residual of out_proj)
constexpr int XP_S_X1 = XP_S_X + 4096;           // [1024] f32 (residual of fc2)
constexpr int XP_S_XQ = XP_S_X1 + 4096;          // [256] u32 Q8 activation of qkv / out_proj / fc1
constexpr int XP_S_XD = XP_S_XQ + 1024;          // [32]
constexpr int XP_S_XS = XP_S_XD + 128;           // [32]
constexpr int XP_S_RED = XP_S_XS + 128;          // [8] double
constexpr int XP_S_HQ = XP_S_RED + 64;           // [1024] u32 fc1 output as Q8
constexpr int XP_S_HD = XP_S_HQ + 4096;          // [128]
constexpr int XP_S_HS = XP_S_HD + 512;           // [128]
constexpr int XP_S_PART = XP_S_HS + 512;         // [256 rows][DEC_PS] f32 block terms (lm_head: 256 rows per workgroup, q/k/v: 192, fc1: 128; fc2: [32][DEC_PS2])
constexpr int XP_S_G = XP_S_PART + 256 * DEC_PS * 4;   // [128] GELU outputs
constexpr int XP_S_LN = XP_S_G + 512;            // [4][1024] f32 ln0_w, ln0_b, ln1_w, ln1_b
constexpr int XP_S_BIAS = XP_S_LN + 16384;       // [192 + 32 + 128 + 32] f32: q/k/v rows of the head, out_proj / fc1 / fc2 rows of the workgroup
constexpr int XP_S_CUR = XP_S_BIAS + 1536;       // [192] f32 q, k, v of this token (head = slot)
constexpr int XP_S_S = XP_S_CUR + 768;           // [256] softmax numerators
constexpr int XP_S_REDF = XP_S_S + 1024;         // [64] f32 + [64] int
constexpr int XP_S_REDD = XP_S_REDF + 512;       // [16] double
constexpr int XP_S_PV = XP_S_REDD + 128;         // [1024] double
constexpr int XP_S_TOTAL = XP_S_PV + 8192;
static_assert(32 * DEC_PS2 <= 256 * DEC_PS, "fc2 block terms fit the shared region");
__host__ __device__ inline size_t xpipe_smem_bytes(int gelu_entries) { return XP_S_TOTAL + (size_t)gelu_entries * 2; }

// units travel through the stages settled -- nibbles unpacked to bytes, fp16 scales converted to f32 (EXPAND; needs split layers: 9-10
// registers per unit)
template <int WT, bool EXPAND>
__device__ __forceinline__ float xp_dot(const Unit<WT> &u, const uint32_t *xq, float xd, float xs_f, int xs_i) {
    if constexpr (EXPAND) return unit_dot_settled<WT>(u, xq, xd, xs_f, xs_i);
    else return unit_dot_quant<WT>(u, xq, xd, xs_f, xs_i);
}
// unpack a freshly loaded unit and keep the result in registers HERE (the empty asm stops the scheduler from sinking the unpack
// instructions back to the unit's use, where they would sit on the stage's dependent chain again)
template <int WT, bool EXPAND>
__device__ __forceinline__ void xp_settle(Unit<WT> &u) {
    if constexpr (EXPAND) {
        settle_unit<WT>(u);
        asm volatile("" : "+v"(u.q0.x), "+v"(u.q0.y), "+v"(u.q0.z), "+v"(u.q0.w), "+v"(u.q1.x), "+v"(u.q1.y), "+v"(u.q1.z), "+v"(u.q1.w), "+v"(u.sc));
    }
}

// sum of one unsigned value per lane over the wave, uniform result (4 DPP adds, 4 v_readlane, scalar adds)
__device__ __forceinline__ uint32_t xp_wave_sum_u32(uint32_t v) {
    int s = (int)v;
    s += dpp_i<DPP_QUAD_XOR1>(s); s += dpp_i<DPP_QUAD_XOR2>(s); s += dpp_i<DPP_ROW_HALF_MIRROR>(s); s += dpp_i<DPP_ROW_MIRROR>(s);
    return (uint32_t)((__builtin_amdgcn_readlane(s, 0) + __builtin_amdgcn_readlane(s, 16)) + (__builtin_amdgcn_readlane(s, 32) + __builtin_amdgcn_readlane(s, 48)));
}

// The launch's work for one role: ATTN = this workgroup is one of the XCD's 16 attention heads (else it computes q/k/v rows).  The
// role is a template parameter because the register allocator, given ONE function with a run-time role branch, spills 21-39 VGPRs
// although each role alone fits (measured: 198-240 of 256 registers per role): two copies of the loop, no spills.
// SPLIT (Q8_0: 9 registers per weight unit, a whole layer does not fit one XCD's registers): a layer is two pipeline units -- LayerNorm,
// q/k/v, attention, out_proj on an EVEN XCD (roles 0 / 1), LayerNorm, fc1, fc2 on the next, ODD XCD (role 2: all 32 workgroups) -- and
// the out_proj output crosses XCDs like the layer output does.
// RES: a resident launch (biogpt_hip_eval): tokens from the host's mailbox, every publish / append gated by the wave's tag (see xp_sweep_q); the ordinary
// instantiation compiles all of that away
template <int WT, int LPK, int NW, int KCAP, int ROLE, bool SPLIT, bool RES>
__device__ __forceinline__ void xp_run(const XpParams &p, unsigned char *smem, const int xcd, const int slot, const uint32_t epoch0, const int n_past0,
                                       const int n_gen0) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant && (WT != W_Q8_0 || SPLIT), "18-30 weight units per lane must fit the register file (Q8_0: 9 registers per unit: split layers)");
    static_assert(ROLE == 0 || ROLE == 1 || (ROLE == 2 && SPLIT), "0 attention head, 1 q/k/v rows, 2 (split layers) the MLP half");
    static_assert(SPLIT || TI::q81, "the producers publish Q8_0 block sums only for Q4_1 / Q5_1 (want_sum = q81): the symmetric formats need the signed units of the split form (unit_dot_settled)");
    constexpr bool ATTN = ROLE == 0;
    constexpr bool EXPAND = SPLIT;      // units are settled (nibbles unpacked, scales converted) while they wait; Q8_0: the scale only
    constexpr bool FIRST = !SPLIT || ROLE != 2, SECOND = !SPLIT || ROLE == 2;      // which stages this workgroup runs
    static_assert(LPK == 2 || LPK == 4 || LPK == 8 || LPK == 16, "lanes per key");
    static_assert(NW == 8 || NW == 16, "waves per workgroup");
    constexpr int D = 1024, DK = 64, NT = NW * 64;
    constexpr int QS = 96 / NW, OS = 16 / NW, FS = 64 / NW, F2R = 32 / NW;     // 2-row steps of qkv / out_proj / fc1 per wave; fc2 rows per wave
    // DUAL (257 .. 512 keys): BOTH workgroups of a head -- h and 16 + h -- are attention workgroups, each with 256 keys' K / V rows in its registers and half
    // of the head's 192 q / k / v rows to compute; they exchange those rows, their scores and (16 + h -> h) the partial PV sums inside the XCD
    constexpr bool DUAL = SPLIT && KCAP > 256;
    constexpr int KW = DUAL ? KCAP / 2 : KCAP;          // keys per workgroup
    constexpr int HI = (DUAL && ROLE == 1) ? 1 : 0;     // which half of the keys (and of the q / k / v rows)
    static_assert(KW % NW == 0 && KW <= NW * 64 / LPK, "key capacity of the launch");
    constexpr int NF4 = 16 / LPK, NV = KW / NW;        // float4 of a key row per lane; values per lane (key slices of NW)
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
    uint32_t *const s_dead = reinterpret_cast<uint32_t *>(smem + XP_S_REDD + 96);      // [4] behind the 8 doubles the attention stage uses
    uint32_t *const s_kvdead = reinterpret_cast<uint32_t *>(smem + XP_S_REDD + 64);    // [2] resident launch, XP_SPLIT_Q: the waves that took the token's k / v rows in saw the launch drain
    uint32_t *const s_spec = reinterpret_cast<uint32_t *>(smem + XP_S_REDD + 112);     // [3] resident launch, workgroup 0 of XCD 0: {speculate the next token, the token the current pass was started with unasked (-1: none)}
    const int t_cap = p.t_cap;
    const int n_units = SPLIT ? 2 * p.n_layer : p.n_layer;                     // pipeline units: layers, or half layers
    const int last_xcd = (n_units - 1) & 7;
    // Several tokens per launch (n_tok > 1: the device-resident generation loop): token t + 1 starts from the arg-max partials of
    // token t's logits, handed to XCD 0 as granules; the hand-off tag is the launch counter + the token's index in the launch.
    for (int tk = 0; tk < p.n_tok; tk++) {
    const uint32_t epoch = epoch0 + (uint32_t)tk;
    uint32_t etag = epoch;          // publishing tag of this wave: 0 once it has seen the error / quit word (xp_sweep_q)
    const int n_past = n_past0 + tk, T = n_past + 1;
    if (tk > 0 && __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) break;       // a disturbed launch drains token by token
    for (int U = xcd; U < n_units; U += 8) {
        const int L = SPLIT ? (U >> 1) : U;
        // the thread index goes through an empty asm in every iteration: without it the compiler hoists a few hundred
        // per-thread addresses (LDS carve, granule slots, weight rows) out of the layer loop and spills them (120 VGPRs
        // of "folded spills" measured); recomputing them costs a handful of integer instructions per stage
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const int sub = lane & 31, rsub = lane >> 5;
        const bool worker = tid < 256;
        if (tk == 0 && U == xcd && U >= 2) {      // start this XCD's first weight load when unit U - 1 starts, not all eight at once
            if (tid == 0) {
                // the output of unit U - 2: a layer's x, or (split layers, U - 2 has U's parity) the first half's x1 / the second half's x
                const xp_u64 *g = SPLIT ? (ROLE != 2 ? XPL(L - 1).gx1 : XPL(L - 1).gx) : XPL(L - 2).gx;
                for (uint32_t spins = 0;; spins++) {
                    if ((uint32_t)(__hip_atomic_load(g, XP_RLX) >> 32) == epoch) break;
                    if (spins >= XP_SPIN_MAX) { xp_fail(p, 3u); break; }
                    if ((spins & 255u) == 255u && __hip_atomic_load(p.ctl + 1, XP_RLX) != 0u) break;
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            __syncthreads();
        }
        const XpLayerK &Y = XPL(L);
        xp_u64 *const G = p.gran + (size_t)L * XP_G_LAYER;
        // ---- this layer's weights into registers, its small vectors into LDS: issued as soon as the previous layer of this
        //      XCD is done, i.e. seven layers ahead of their use.  Workgroups 0-15 are the layer's attention heads and hold the
        //      head's old keys / values instead of q/k/v weights; workgroup 16 + h computes all 192 q/k/v rows of head h.
        constexpr bool attn_wg = ATTN || (DUAL && ROLE == 1);
        const int head = slot & 15;
        Unit<WT> wo[OS], w1[FS], w2[F2R][2];
        {
            float4 l0 = make_float4(0.f, 0.f, 0.f, 0.f), l1 = l0, l2 = l0, l3 = l0;
            if (worker) {
                if (FIRST) { l0 = xp_ldg4(Y.ln0_w, tid); l1 = xp_ldg4(Y.ln0_b, tid); }
                if (SECOND) { l2 = xp_ldg4(Y.ln1_w, tid); l3 = xp_ldg4(Y.ln1_b, tid); }
            }
            float bv = 0.0f;
            if (tid < 192) { if (FIRST) bv = ((xp_gf)Y.bqkv)[(tid >> 6) * 1024 + head * 64 + (tid & 63)]; }
            else if (tid < 224) { if (FIRST) bv = ((xp_gf)Y.bo)[slot * 32 + tid - 192]; }
            else if (tid < 352) { if (SECOND) bv = ((xp_gf)Y.b1)[slot * 128 + tid - 224]; }
            else if (tid < 384) { if (SECOND) bv = ((xp_gf)Y.b2)[slot * 32 + tid - 352]; }
            if (FIRST) {
#pragma unroll
                for (int s = 0; s < OS; s++) load_unit<WT>(wo[s], XPL_MATRIX(Y.Wo), (int64_t)(slot * 32 + s * 2 * NW + wave * 2 + rsub) * 32 + sub);
            }
            if (SECOND) {
#pragma unroll
                for (int s = 0; s < FS; s++) load_unit<WT>(w1[s], XPL_MATRIX(Y.W1), (int64_t)(slot * 128 + s * 2 * NW + wave * 2 + rsub) * 32 + sub);
#pragma unroll
                for (int r = 0; r < F2R; r++)
#pragma unroll
                    for (int it = 0; it < 2; it++) load_unit<WT>(w2[r][it], XPL_MATRIX(Y.W2), (int64_t)(slot * 32 + wave * F2R + r) * 128 + lane + 64 * it);
            }
            if (worker) {
                if (FIRST) { reinterpret_cast<float4 *>(s_ln)[tid] = l0; reinterpret_cast<float4 *>(s_ln + 1024)[tid] = l1; }
                if (SECOND) { reinterpret_cast<float4 *>(s_ln + 2048)[tid] = l2; reinterpret_cast<float4 *>(s_ln + 3072)[tid] = l3; }
            }
            if (tid < 384) s_bias[tid] = bv;
            if constexpr (EXPAND) {      // waits for the loads: idle time, the unit's turn is layers away
                if (FIRST) {
#pragma unroll
                    for (int s = 0; s < OS; s++) xp_settle<WT, EXPAND>(wo[s]);
                }
                if (SECOND) {
#pragma unroll
                    for (int s = 0; s < FS; s++) xp_settle<WT, EXPAND>(w1[s]);
#pragma unroll
                    for (int r = 0; r < F2R; r++) { xp_settle<WT, EXPAND>(w2[r][0]); xp_settle<WT, EXPAND>(w2[r][1]); }
                }
            }
        }
        // the layer input, 4 elements per LayerNorm worker (waves 0-3): the embedding (layer 0) or the previous layer's granules
        auto layer_input = [&]() -> float4 {
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (L == 0) {
                int tok;
                // the position's row of embed_positions does not depend on the token: requested before the sampler, in registers when the token is known
                const bool emb_fast = XPK(tok_emb.type) == WT && XPK(pos_emb.type) == WT;
                uint32_t pq = 0u, psc = 0u, pqh = 0u;
                if (emb_fast && worker) xp_row4_request<WT>(XPK_MATRIX(pos_emb), n_past + 2, tid, pq, psc, pqh);
                // greedy sampler of the previous token of THIS launch: its per-block partials arrive as granules from the
                // XCDs that computed the logits (two blocks per thread at most: lm_blocks <= 1024); every thread returns the arg-max
                auto sample_prev = [&]() __attribute__((always_inline)) -> int {
                    float bv = -INFINITY;
                    int bi = 0x7fffffff;
                    {
                        uint32_t v[4] = {0u, 0u, 0u, 0u};
                        const bool a0 = tid < XPK(lm_blocks), a1 = tid + NT < XPK(lm_blocks);
                        const uint32_t prev = epoch - 1u;
                        for (uint32_t spins = 0; !RES || etag != 0u; spins++) {
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
                            if (spins >= XP_SPIN_MAX) { if (lane == 0) xp_fail(p, 4u); if (RES) etag = 0u; break; }
                            if ((spins & (RES ? 255u : 1023u)) == (RES ? 255u : 1023u) && __any(__hip_atomic_load(p.ctl + 1, XP_RLX) != 0u)) { if (RES) etag = 0u; break; }
                            
                        }
                        if (a0) { bv = __uint_as_float(v[0]); bi = (int)v[1]; }
                        if (a1) {
                            const float ov = __uint_as_float(v[2]);
                            const int oi = (int)v[3];
                            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                        }
                    }
                    xp_argmax_wave(bv, bi);
                    if (lane == 0) { s_redf[wave] = bv; s_redi[wave] = bi; }
                    __syncthreads();
                    bv = s_redf[0]; bi = s_redi[0];
#pragma unroll
                    for (int w = 1; w < NW; w++)
                        if (s_redf[w] > bv || (s_redf[w] == bv && s_redi[w] < bi)) { bv = s_redf[w]; bi = s_redi[w]; }
                    if (bi < 0 || bi >= XPK(n_vocab)) bi = 0;
                    return bi;
                };
                if (RES && p.resident != 0 && tk == 0 && slot == 0 && tid == 0) { s_spec[0] = (uint32_t)XPK(res_spec0); s_spec[1] = 0xffffffffu; }
                if (RES && p.resident != 0 && tk > 0) {
                    // resident launch: the next token is the one the NEXT biogpt_eval() call posts in the pinned mailbox -- or, speculating, the device's own
                    // arg-max of the previous one, which that post must then confirm.  Workgroup 0 decides and hands the token to the XCD's other workgroups
                    // as a granule; a wait for the host lasts at most idle_ticks, then the launch ends cleanly (xp_quit)
                    xp_u64 *const gt = XPK(samp) + 2048;
                    if (slot == 0) {
                        unsigned long long *const wl4 = ((XPK(res_dbg) & 32) && XPK(wall) && tid == 0) ? XPK(wall) + 32768 + (size_t)((XPK(mbox_seq0) + (uint32_t)tk) & 4095u) * 4 : nullptr;
                        if (wl4) wl4[0] = wall_clock64();
                        const uint32_t want = XPK(mbox_seq0) + (uint32_t)tk;
                        // the post with sequence number seq, for position np (ONE 8-byte word, xp_post): its token (>= 0) and "speculate the token after this one",
                        // or -1: time-out / the host asks the launch to leave / the launch is failing
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
                                if (!(XPK(res_dbg) & 2)) __builtin_amdgcn_s_sleep(4);
                            }
                        };
                        // A pass that was started unasked (from the device's own arg-max): the host's post for it -- made while that pass was running, long ago
                        // when the host keeps up -- is read NOW, while this workgroup waits for the pass to end anyway: behind the logits rows, which are about
                        // to go out, a PCIe read queues for microseconds.  It must name the same token and position; it also says that the host is done
                        // with the buffers the coming pass will write.
                        if (tid == 0) {
                            const int pending = (int)s_spec[1];
                            int unused = 0;
                            s_spec[2] = (etag != 0u && (pending < 0 || wait_post(want - 1u, n_past - 1, unused) == pending)) ? 1u : 0u;
                        }
                        const int guess = sample_prev();       // arg-max of the previous token's logits (barriers inside: the whole workgroup)
                        __syncthreads();                       // s_redf is reused by the attention stage
                        if (wl4) wl4[1] = wall_clock64();
                        if (tid == 0 && etag != 0u) {
                            int spec = (int)s_spec[0];
                            // once a pass has been started unasked the following ones are too: the host asks for that as long as its tokens match, and a mismatch ends the launch
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
                            } else xp_quit(p);
                            if ((XPK(res_dbg) & 32) && XPK(wall)) XPK(wall)[(size_t)(want & 4095u) * 2] = wall_clock64();
                            if (wl4) wl4[2] = wall_clock64();
                        }
                    }
                    uint32_t v[1];
                    xp_sweep_q<RES, 1>(gt, true, epoch, v, p, etag);
                    tok = (int)v[0];
                    if (tok < 0 || tok >= XPK(n_vocab)) tok = 0;
                } else if (tk > 0) {
                    tok = sample_prev();
                    if (slot == 0 && tid == 0) {
                        int32_t *tokens = state_tokens(XPK(st));
                        const int g = n_gen0 + tk;
                        if (g < XPK(n_positions)) tokens[XPK(n_positions) + g] = tok;
                        tokens[0] = tok;
                    }
                    __syncthreads();       // s_redf is reused by the attention workgroups
                } else if (XPK(tok_src) == 2) {
                    // greedy sampler of the PREVIOUS token (main.cpp:109-128, top_k = 1): arg-max over the lm_head kernel's
                    // per-workgroup partials, lowest id wins ties; workgroup 0 records it
                    float bv = -INFINITY;
                    int bi = 0x7fffffff;
                    for (int k = tid; k < XPK(nparts); k += NT) {
                        const float v = XPK(pmax_val)[k];
                        const int ix = XPK(pmax_idx)[k];
                        if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
                    }
    #pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        const float ov = __shfl_xor(bv, off, 64);
                        const int oi = __shfl_xor(bi, off, 64);
                        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                    }
                    if (lane == 0) { s_redf[wave] = bv; s_redi[wave] = bi; }
                    __syncthreads();
                    bv = s_redf[0]; bi = s_redi[0];
    #pragma unroll
                    for (int w = 1; w < NW; w++)
                        if (s_redf[w] > bv || (s_redf[w] == bv && s_redi[w] < bi)) { bv = s_redf[w]; bi = s_redi[w]; }
                    tok = bi;
                    if (tok < 0 || tok >= XPK(n_vocab)) tok = 0;
                    if (slot == 0 && tid == 0) {
                        int32_t *tokens = state_tokens(XPK(st));
                        const int g = n_gen0;
                        if (g < XPK(n_positions)) tokens[XPK(n_positions) + g] = tok;
                        tokens[0] = tok;
                    }
                } else {
                    tok = (RES && p.resident != 0) ? XPK(res_tok0) : state_tokens(XPK(st))[0];
                }
                if (slot == 0) XP_TAIL(tk, 5);
                if (worker) {      // biogpt.cpp:664-686: embed_tokens[tok] * sqrt(D) + embed_positions[n_past + 2]
                    float e[4];
                    if (emb_fast) {
                        uint32_t tq, tsc, tqh;
                        xp_row4_request<WT>(XPK_MATRIX(tok_emb), tok, tid, tq, tsc, tqh);
                        float te[4], pe[4];
                        xp_row4_values<WT>(pq, psc, pqh, tid, pe);
                        xp_row4_values<WT>(tq, tsc, tqh, tid, te);
    #pragma unroll
                        for (int j = 0; j < 4; j++) e[j] = __fadd_rn(__fmul_rn(te[j], XPK(embed_scale)), pe[j]);
                    } else {
    #pragma unroll
                        for (int j = 0; j < 4; j++)
                            e[j] = __fadd_rn(__fmul_rn(dequant_elem(XPK_MATRIX(tok_emb), tok, 4 * tid + j), XPK(embed_scale)), dequant_elem(XPK_MATRIX(pos_emb), n_past + 2, 4 * tid + j));
                    }
                    xv = make_float4(e[0], e[1], e[2], e[3]);
                }
            } else if (wave < 4) {
                uint32_t v[4];
                xp_sweep_q<RES, 4, 256, true>(XPL(L - 1).gx + tid, true, epoch, v, p, etag);
                xv = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
            // the waves that took the layer input in tell the others whether it was real (waves 4-7 sweep nothing here, yet their lanes append K / V rows):
            // read behind LayerNorm's barriers
            if (RES && wave < 4 && lane == 0) { s_dead[wave] = (etag == 0u) ? 1u : 0u; }
            return xv;
        };
        if constexpr (FIRST) {
        // split layers: the head's workgroup computes its OWN q / k / v rows (12 units beside the K / V rows fit now) -- the hand-over
        // of the rows from a partner workgroup (0.58 us per layer) is gone; workgroups 16-31 only take part in out_proj
        // (up to 128 keys: with the 56-64 K / V registers of the 192- / 256-key variants the 12 extra units spill 10-43 VGPRs, or fit
        // exactly and run slower: 303 against 297 us per token at 161 keys)
        constexpr bool MERGE = SPLIT && KCAP <= 128;
        if (!attn_wg && MERGE) {
            const float4 xv = layer_input();
            if (worker) reinterpret_cast<float4 *>(s_x)[tid] = xv;
            XP_WALL(0);
        }
        if (!attn_wg && !MERGE) {
            // ================= stage A (workgroups 16-31): LayerNorm -> Q8 -> the 192 q / k / v rows of head `head` =================
            constexpr int Q0 = (KCAP <= 192) ? QS / 3 : 0;      // XP_SPLIT_Q: the q rows (units 0 .. QS / 3 - 1) are the head's own workgroup's
            Unit<WT> wqkv[QS];
#pragma unroll
            for (int s = Q0; s < QS; s++) {
                const int jj = s * 2 * NW + wave * 2 + rsub;
                load_unit<WT>(wqkv[s], XPL_MATRIX(Y.Wqkv), (int64_t)((jj >> 6) * 1024 + head * 64 + (jj & 63)) * 32 + sub);
            }
#pragma unroll
            for (int s = Q0; s < QS; s++) xp_settle<WT, EXPAND>(wqkv[s]);
            const float4 xv = layer_input();
            XP_WALL(0);
            float4 lnw = xv, lnb = xv;
            if (worker) {
                reinterpret_cast<float4 *>(s_x)[tid] = xv;
                lnw = reinterpret_cast<const float4 *>(s_ln)[tid]; lnb = reinterpret_cast<const float4 *>(s_ln + 1024)[tid];
            }
            ln4_q8_1024<TI::q81, TI::q81>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
            XP_WALL(6);
            uint32_t ax[8];
            const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
            ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
            const float axd = s_xd[sub];
            const uint32_t axs = s_xs[sub];
            float *const part = s_part + wave * 2 * QS * DEC_PS;
#pragma unroll
            for (int s = Q0; s < QS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, EXPAND>(wqkv[s], ax, axd, __uint_as_float(axs), (int)axs);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane >= 2 * Q0 && lane < 2 * QS) {
                const int jj = (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);
                float v = __fadd_rn(s_bias[jj], sum32_in_order(part + lane * DEC_PS));
                const int which = jj >> 6, d = jj & 63;
                if (which == 0) v = __fmul_rn(v, p.q_scale);                   // Q scaled AFTER the bias (biogpt.cpp:708-710)
                xp_put_local(G + XP_G_QKV + which * 1024 + head * 64 + d, etag, __float_as_uint(v));
                if (which != 0 && (!RES || (etag != 0u && (s_dead[0] | s_dead[1] | s_dead[2] | s_dead[3]) == 0u))) {      // KV append (biogpt.cpp:721-727), head-major cache; never from a draining launch
                    float *kc_ = Y.kcache, *vc_ = Y.vcache;
                    asm volatile("" : "+s"(kc_), "+s"(vc_));      // both by scalar loads (a per-lane choice of the table's FIELD is a vector load of the pointer)
                    ((__attribute__((address_space(1))) float *)((which == 1) ? kc_ : vc_))[((size_t)head * p.P + n_past) * DK + d] = v;
                }
            }
            XP_WALL(1);
        }
        if (attn_wg) {
            // ================= stage B (workgroups 0-15): attention of head `head` (biogpt.cpp:729-764) =================
            // the old keys / values of this head: in flight since the previous layer of this XCD finished
            float4 kr[NF4];
            float vr[NV];
            const int ksub = tid & (LPK - 1), kidx = tid / LPK;
            const int dd = tid & (DK - 1), sl = tid >> 6;
            {       // rows appended by earlier tokens of this launch: by this workgroup itself (MERGE), else by workgroup 16 + head -> agent-scope loads (xp_kv_load*)
                constexpr bool KV_SC1 = !MERGE;
                const float *kb = Y.kcache + (size_t)head * p.P * DK, *vb = Y.vcache + (size_t)head * p.P * DK;
                const __amdgpu_buffer_rsrc_t krs = xp_kv_rsrc(kb, p.P * DK * 4), vrs = xp_kv_rsrc(vb, p.P * DK * 4);
                // (KV_SC1: unconditional -- a row beyond the cache slice reads as 0, buffer loads are range-checked -- so that no register is left undefined on any path)
                if (KV_SC1 || kidx < t_cap) {
#pragma unroll
                    for (int m = 0; m < NF4; m++) kr[m] = xp_kv_load4<KV_SC1>(krs, kb, ((HI * KW + kidx) * (DK / 4) + ksub + LPK * m) * 4);
                } else {
#pragma unroll
                    for (int m = 0; m < NF4; m++) kr[m] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const int j = HI * KW + sl + NW * k;
                    vr[k] = (KV_SC1 || j < t_cap) ? xp_kv_load1<KV_SC1>(vrs, vb, j * DK + dd) : 0.0f;
                }
            }
            if constexpr (MERGE) {
                // ---- stage A in here: LayerNorm -> Q8 -> the 192 q / k / v rows of this head, straight into s_cur ----
                Unit<WT> wqkv[QS];
#pragma unroll
                for (int s = 0; s < QS; s++) {
                    const int jj = s * 2 * NW + wave * 2 + rsub;
                    load_unit<WT>(wqkv[s], XPL_MATRIX(Y.Wqkv), (int64_t)((jj >> 6) * 1024 + head * 64 + (jj & 63)) * 32 + sub);
                }
#pragma unroll
                for (int s = 0; s < QS; s++) xp_settle<WT, EXPAND>(wqkv[s]);
                const float4 xv = layer_input();
                XP_WALL(0);
                if (L == 0 && slot == 0) { asm volatile("" :: "v"(xv.x)); XP_TAIL(tk, 6); }
                float4 lnw = xv, lnb = xv;
                if (worker) {
                    reinterpret_cast<float4 *>(s_x)[tid] = xv;
                    lnw = reinterpret_cast<const float4 *>(s_ln)[tid]; lnb = reinterpret_cast<const float4 *>(s_ln + 1024)[tid];
                }
                ln4_q8_1024<TI::q81, TI::q81>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
                XP_WALL(6);
                if (L == 0 && slot == 0) XP_TAIL(tk, 7);
                uint32_t ax[8];
                const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
                ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
                const float axd = s_xd[sub];
                const uint32_t axs = s_xs[sub];
                float *const part = s_part + wave * 2 * QS * DEC_PS;
#pragma unroll
                for (int s = 0; s < QS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, EXPAND>(wqkv[s], ax, axd, __uint_as_float(axs), (int)axs);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < 2 * QS) {
                    const int jj = (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);
                    float v = __fadd_rn(s_bias[jj], sum32_in_order(part + lane * DEC_PS));
                    const int which = jj >> 6, d = jj & 63;
                    if (which == 0) v = __fmul_rn(v, p.q_scale);                   // Q scaled AFTER the bias (biogpt.cpp:708-710)
                    s_cur[jj] = v;
                    if (which != 0 && (!RES || (etag != 0u && (s_dead[0] | s_dead[1] | s_dead[2] | s_dead[3]) == 0u))) {      // KV append (biogpt.cpp:721-727), head-major cache; never from a draining launch
                        float *kc_ = Y.kcache, *vc_ = Y.vcache;
                        asm volatile("" : "+s"(kc_), "+s"(vc_));      // both by scalar loads (a per-lane choice of the table's FIELD is a vector load of the pointer)
                        ((__attribute__((address_space(1))) float *)((which == 1) ? kc_ : vc_))[((size_t)head * p.P + n_past) * DK + d] = v;
                    }
                }
                XP_WALL(1);
            } else if constexpr (DUAL) {
                // ---- half of the head's 192 q / k / v rows in here (rows 96 HI .. 96 HI + 95: q and k[0..31], or k[32..63] and v), the other half from the partner ----
                constexpr int QH = QS / 2;
                Unit<WT> wq[QH];
#pragma unroll
                for (int s = 0; s < QH; s++) {
                    const int jj = (HI * QH + s) * 2 * NW + wave * 2 + rsub;
                    load_unit<WT>(wq[s], XPL_MATRIX(Y.Wqkv), (int64_t)((jj >> 6) * 1024 + head * 64 + (jj & 63)) * 32 + sub);
                }
#pragma unroll
                for (int s = 0; s < QH; s++) xp_settle<WT, EXPAND>(wq[s]);
                const float4 xv = layer_input();
                XP_WALL(0);
                float4 lnw = xv, lnb = xv;
                if (worker) {
                    reinterpret_cast<float4 *>(s_x)[tid] = xv;
                    lnw = reinterpret_cast<const float4 *>(s_ln)[tid]; lnb = reinterpret_cast<const float4 *>(s_ln + 1024)[tid];
                }
                ln4_q8_1024<TI::q81, TI::q81>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
                XP_WALL(6);
                uint32_t ax[8];
                const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
                ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
                const float axd = s_xd[sub];
                const uint32_t axs = s_xs[sub];
                float *const part = s_part + wave * 2 * QH * DEC_PS;
#pragma unroll
                for (int s = 0; s < QH; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, EXPAND>(wq[s], ax, axd, __uint_as_float(axs), (int)axs);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < 2 * QH) {
                    const int jj = (HI * QH + (lane >> 1)) * 2 * NW + wave * 2 + (lane & 1);
                    float v = __fadd_rn(s_bias[jj], sum32_in_order(part + lane * DEC_PS));
                    const int which = jj >> 6, d = jj & 63;
                    if (which == 0) v = __fmul_rn(v, p.q_scale);                   // Q scaled AFTER the bias (biogpt.cpp:708-710)
                    s_cur[jj] = v;
                    xp_put_local(G + XP_G_QKV + which * 1024 + head * 64 + d, etag, __float_as_uint(v));
                    if (which != 0 && (!RES || (etag != 0u && (s_dead[0] | s_dead[1] | s_dead[2] | s_dead[3]) == 0u))) {      // KV append (biogpt.cpp:721-727), head-major cache; never from a draining launch
                        float *kc_ = Y.kcache, *vc_ = Y.vcache;
                        asm volatile("" : "+s"(kc_), "+s"(vc_));      // both by scalar loads (a per-lane choice of the table's FIELD is a vector load of the pointer)
                        ((__attribute__((address_space(1))) float *)((which == 1) ? kc_ : vc_))[((size_t)head * p.P + n_past) * DK + d] = v;
                    }
                }
                XP_WALL(1);
                if (wave < 2) {       // the partner's 96 rows
                    uint32_t v[1];
                    const int jo = (1 - HI) * 96 + (tid < 96 ? tid : 0);
                    xp_sweep_q<RES, 1>(G + XP_G_QKV + (jo >> 6) * 1024 + head * 64 + (jo & 63), tid < 96, epoch, v, p, etag);
                    if (tid < 96) s_cur[jo] = __uint_as_float(v[0]);
                    if (RES && lane == 0) { s_kvdead[wave] = (etag == 0u) ? 1u : 0u; }      // a draining launch: every wave of the workgroup must stop publishing (read behind the barrier below)
                }
            } else if constexpr (KCAP <= 192) {
                // ---- the head's 64 q rows in here (XP_SPLIT_Q): LayerNorm -> Q8 -> 4 units per lane -> s_cur[0 .. 63]; k / v of this token come from workgroup 16 + head ----
                constexpr int Q0 = QS / 3;
                Unit<WT> wq[Q0];
#pragma unroll
                for (int s = 0; s < Q0; s++) {
                    const int jj = s * 2 * NW + wave * 2 + rsub;      // < 64: a q row
                    load_unit<WT>(wq[s], XPL_MATRIX(Y.Wqkv), (int64_t)(head * 64 + jj) * 32 + sub);
                }
#pragma unroll
                for (int s = 0; s < Q0; s++) xp_settle<WT, EXPAND>(wq[s]);
                const float4 xv = layer_input();
                XP_WALL(0);
                float4 lnw = xv, lnb = xv;
                if (worker) {
                    reinterpret_cast<float4 *>(s_x)[tid] = xv;
                    lnw = reinterpret_cast<const float4 *>(s_ln)[tid]; lnb = reinterpret_cast<const float4 *>(s_ln + 1024)[tid];
                }
                ln4_q8_1024<TI::q81, TI::q81>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
                XP_WALL(6);
                uint32_t ax[8];
                const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
                ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
                const float axd = s_xd[sub];
                const uint32_t axs = s_xs[sub];
                float *const part = s_part + wave * 2 * Q0 * DEC_PS;
#pragma unroll
                for (int s = 0; s < Q0; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, EXPAND>(wq[s], ax, axd, __uint_as_float(axs), (int)axs);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < 2 * Q0) {
                    const int jj = (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);
                    s_cur[jj] = __fmul_rn(__fadd_rn(s_bias[jj], sum32_in_order(part + lane * DEC_PS)), p.q_scale);      // Q scaled AFTER the bias (biogpt.cpp:708-710)
                }
                XP_WALL(1);
            } else {
                {   // the layer input is the residual of stage C; it arrives about 2 us before the q / k / v rows
                    const float4 xv = layer_input();
                    if (worker) reinterpret_cast<float4 *>(s_x)[tid] = xv;
                }
                XP_WALL(0);
                if (wave < 3) {
                    uint32_t v[1];
                    xp_sweep_q<RES, 1>(G + XP_G_QKV + wave * 1024 + head * 64 + lane, true, epoch, v, p, etag);
                    s_cur[tid] = __uint_as_float(v[0]);
                }
            }
            __syncthreads();
            if constexpr (RES && DUAL) { if ((s_kvdead[0] | s_kvdead[1]) != 0u) etag = 0u; }
            XP_WALL(7);
            constexpr bool LATE_KV = !MERGE && KCAP <= 192;      // the token's own k / v rows arrive while the old keys' scores are computed
            auto key_score = [&](const float4 (&kk)[NF4]) __attribute__((always_inline)) -> float {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int m = 0; m < NF4; m++) {
                    const float4 qm = *reinterpret_cast<const float4 *>(s_cur + 4 * (LPK * m + ksub));
                    a0 += (double)__fmul_rn(kk[m].x, qm.x); a1 += (double)__fmul_rn(kk[m].y, qm.y);
                    a2 += (double)__fmul_rn(kk[m].z, qm.z); a3 += (double)__fmul_rn(kk[m].w, qm.w);
                }
                double acc = (a0 + a1) + (a2 + a3);
                acc += dpp_d<DPP_QUAD_XOR1>(acc);
                if (LPK >= 4) acc += dpp_d<DPP_QUAD_XOR2>(acc);
                if (LPK >= 8) acc += dpp_d<DPP_ROW_HALF_MIRROR>(acc);
                if (LPK >= 16) acc += dpp_d<DPP_ROW_MIRROR>(acc);
                return (float)acc;
            };
            if constexpr (LATE_KV) {      // the two highest waves (no keys of their own up to 192 keys) take the k / v rows in: wave NW - 2 the key row, wave NW - 1 the value row
                if (wave >= NW - 2) {
                    uint32_t v[1];
                    const int which = wave - (NW - 3);
                    xp_sweep_q<RES, 1>(G + XP_G_QKV + which * 1024 + head * 64 + lane, true, epoch, v, p, etag);
                    s_cur[which * 64 + lane] = __uint_as_float(v[0]);
                    if (RES && lane == 0) { s_kvdead[which - 1] = (etag == 0u) ? 1u : 0u; }      // a draining launch: every wave of the workgroup must stop publishing (read behind the barrier below)
                }
            }
            float sc = -INFINITY;
            const int jg = HI * KW + kidx;      // this lane's key
            if (HI * KW * LPK + (tid & ~63) < LPK * T) {
                if (!LATE_KV && jg == n_past) {
#pragma unroll
                    for (int m = 0; m < NF4; m++) kr[m] = *reinterpret_cast<const float4 *>(s_cur + 64 + 4 * (LPK * m + ksub));
                }
                if (jg < T && !(LATE_KV && jg == n_past)) sc = key_score(kr);
            }
            static_assert(NW == 8, "key j = wave + 8 k sits in lane (wave + 8 k) & 63 of slot k >> 3");
            constexpr bool SMW = KCAP <= 64;      // (measured: -1.5 us per token with 64 keys, +-0 with 128, +1 / +9 us with 192 / 256 -- 2-4 look-ups per lane and 16-32 v_readlane per wave)
            constexpr int KC = (KCAP + 63) / 64;
            [[maybe_unused]] float ev[KC];
            float inv;
            if constexpr (SMW) {
                // One exchange instead of two: the scores go to LDS, and behind ONE barrier every wave works the softmax out for itself -- the maximum, the
                // table look-ups and the sum of all keys (<= 4 per lane), identically in the 8 waves -- so that the PV lanes take their weights from the wave's
                // own registers (v_readlane).  The sum of the numerators is formed in integers: they are fp16 values <= 1.0 (exp of a non-positive argument),
                // i.e. multiples of 2^-24, <= 256 of them: sums of the low and high 16 bits of value * 2^24, exact -- as the reference's double sum is.
                if (ksub == 0) s_S[kidx] = sc;      // every key slot of the variant: -inf beyond the context (and, LATE_KV, for the token's own key)
                __syncthreads();                    // the scores -- and (LATE_KV) the token's k / v rows in s_cur
                float scl[KC];
#pragma unroll
                for (int c = 0; c < KC; c++) scl[c] = s_S[lane + 64 * c];
                if constexpr (LATE_KV) {
                    if (RES && (s_kvdead[0] | s_kvdead[1]) != 0u) etag = 0u;
                    float4 kn[NF4];                 // the token's own key: every group of LPK lanes of every wave forms its score (the association of the old keys' scores)
#pragma unroll
                    for (int m = 0; m < NF4; m++) kn[m] = *reinterpret_cast<const float4 *>(s_cur + 64 + 4 * (LPK * m + ksub));
                    const float snew = key_score(kn);
#pragma unroll
                    for (int c = 0; c < KC; c++) if (lane + 64 * c == n_past) scl[c] = snew;
                }
                float mx = scl[0];
#pragma unroll
                for (int c = 1; c < KC; c++) mx = fmaxf(mx, scl[c]);
                mx = wave_max_f32(mx);
                XP_WALL(13);
                uint32_t ulo = 0u, uhi = 0u;
#pragma unroll
                for (int c = 0; c < KC; c++) {
                    float val = 0.0f;
                    if (lane + 64 * c < T) {
                        // ggml_soft_max: fp16 exp table; its non-zero negative slice sits in LDS (XP_EXP_LDS; sc - mx <= 0: the code is 0x0000 or a negative one)
                        const uint32_t ix = f2h(__fsub_rn(scl[c], mx)), neg = ix - 0x8000u;
                        uint16_t e16;
                        if (neg < (uint32_t)p.exp_n) e16 = s_gelu[neg];
                        else if (p.exp_n > 0 && ix == 0u) e16 = 0x3C00;
                        else if (p.exp_n > 0 && neg < 0x7C00u) e16 = 0;
                        else e16 = p.exp_tab[ix];                                   // -inf, NaN, a positive argument (or no slice)
                        val = h2f(e16);
                    }
                    ev[c] = val;
                    const uint32_t u = (uint32_t)__fmul_rn(val, 16777216.0f);
                    ulo += u & 0xFFFFu; uhi += u >> 16;
                }
                ulo = xp_wave_sum_u32(ulo); uhi = xp_wave_sum_u32(uhi);
                const double sum = ((double)uhi * 65536.0 + (double)ulo) * 0x1p-24;
                inv = inv_sum_f32(sum);
            } else {
                if constexpr (LATE_KV) {
                    __syncthreads();      // the token's k / v rows are in s_cur
                    if (RES && (s_kvdead[0] | s_kvdead[1]) != 0u) etag = 0u;
                    if (kidx == n_past) {      // the LPK lanes of the new key
#pragma unroll
                        for (int m = 0; m < NF4; m++) kr[m] = *reinterpret_cast<const float4 *>(s_cur + 64 + 4 * (LPK * m + ksub));
                        sc = key_score(kr);
                    }
                }
                bool valid = jg < T && ksub == 0;      // this lane holds a score of the head
                if constexpr (DUAL) {
                    // the partner's 256 scores travel to the odd lanes (ksub == 1): maximum, look-ups and sum run over all of the head's keys in both workgroups
                    static_assert(LPK == 2, "two lanes per key: the second one takes the partner's score of the same index");
                    xp_u64 *const gsc = p.gran_l + (size_t)L * XL_G_LAYER + XL_G_SC + head * 1024;
                    if (valid) xp_put_local(gsc + jg, etag, __float_as_uint(sc));
                    const int jo = (1 - HI) * KW + kidx;
                    const bool theirs = ksub == 1 && jo < T;
                    uint32_t v[1];
                    xp_sweep_q<RES, 1>(gsc + jo, theirs, epoch, v, p, etag);
                    if (theirs) { sc = __uint_as_float(v[0]); valid = true; }
                }
                float mx = wave_max_f32(sc);
                if (lane == 0) s_redf[wave] = mx;
                if (RES && DUAL && lane == 0) reinterpret_cast<uint32_t *>(s_redf)[NW + wave] = (etag == 0u) ? 1u : 0u;      // (resident two-workgroup variant: a wave whose partner scores never came)
                __syncthreads();
                mx = s_redf[0];
#pragma unroll
                for (int w = 1; w < NW; w++) mx = fmaxf(mx, s_redf[w]);
                if constexpr (RES && DUAL) {       // ... takes every wave of the workgroup with it: nothing this workgroup publishes from here on carries a valid tag
                    uint32_t dead = 0u;
#pragma unroll
                    for (int w = 0; w < NW; w++) dead |= reinterpret_cast<const uint32_t *>(s_redf)[NW + w];
                    if ((dead) != 0u) etag = 0u;
                }
                XP_WALL(13);
                double sum = 0.0;
                if (valid) {
                    // ggml_soft_max: fp16 exp table; its non-zero negative slice sits in LDS (XP_EXP_LDS; sc - mx <= 0: the code is 0x0000 or a negative one)
                    const uint32_t ix = f2h(__fsub_rn(sc, mx)), neg = ix - 0x8000u;
                    uint16_t e16;
                    if (neg < (uint32_t)p.exp_n) e16 = s_gelu[neg];
                    else if (p.exp_n > 0 && ix == 0u) e16 = 0x3C00;
                    else if (p.exp_n > 0 && neg < 0x7C00u) e16 = 0;
                    else e16 = p.exp_tab[ix];                                   // -inf, NaN, a positive argument (or no slice)
                    const float val = h2f(e16);
                    if (ksub == 0) s_S[kidx] = val;
                    sum = (double)val;
                }
                sum = wave_sum_f64(sum);
                if (lane == 0) s_redd[wave] = sum;
                __syncthreads();
                sum = 0.0;
#pragma unroll
                for (int w = 0; w < NW; w++) sum += s_redd[w];
                inv = inv_sum_f32(sum);
            }
            XP_WALL(14);
            {
                // all LDS reads first, no branches in the loop: a key past the context adds +0.0 (exact), never its stale weight
                const float vcur = s_cur[128 + dd];
                [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
                constexpr int CH = NV < 16 ? NV : (NV % 16 == 0 ? 16 : 8);      // softmax weights fetched at most 16 at a time (register budget)
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int k0 = 0; k0 < NV; k0 += CH) {
                    float pj[CH];
#pragma unroll
                    for (int k = 0; k < CH; k++) {
                        if constexpr (SMW) pj[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ev[(k0 + k) >> 3]), wave_u + 8 * ((k0 + k) & 7)));      // key sl + 8 (k0 + k)
                        else pj[k] = s_S[sl + NW * (k0 + k)];
                    }
#pragma unroll
                    for (int k = 0; k < CH; k += 2) {
                        const int j0 = HI * KW + sl + NW * (k0 + k), j1 = j0 + NW;
                        const double c0 = (double)__fmul_rn(j0 == n_past ? vcur : vr[k0 + k], __fmul_rn(pj[k], inv));
                        const double c1 = (double)__fmul_rn(j1 == n_past ? vcur : vr[k0 + k + 1], __fmul_rn(pj[k + 1], inv));
                        a0 += (j0 < T) ? c0 : 0.0;
                        a1 += (j1 < T) ? c1 : 0.0;
                    }
                }
                s_pv[tid] = a0 + a1;
            }
            __syncthreads();
            XP_WALL(15);
            if (tid < DK) {
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int s2 = 0; s2 < NW; s2 += 2) { t0 += s_pv[s2 * DK + tid]; t1 += s_pv[(s2 + 1) * DK + tid]; }
                double tot = t0 + t1;
                [[maybe_unused]] xp_u64 *const gpv = DUAL ? p.gran_l + (size_t)L * XL_G_LAYER + XL_G_PV + head * 16 * 128 : nullptr;
                if constexpr (DUAL && HI == 1) {      // the upper keys' partial sums go to the head's first workgroup
                    xp_put_local(gpv + tid, etag, (uint32_t)__double2loint(tot)); xp_put_local(gpv + 64 + tid, etag, (uint32_t)__double2hiint(tot));
                } else {
                    if constexpr (DUAL) {
                        uint32_t v[2];
                        xp_sweep_q<RES, 2, 64>(gpv + tid, true, epoch, v, p, etag);
                        tot += __hiloint2double((int)v[1], (int)v[0]);      // lower keys + upper keys (attn_split_combine_kernel's range order)
                    }
                    const float o = (float)tot;
                    int8_t q8; float d8; uint32_t s8;
                    q8_block32(o, TI::q81, q8, d8, s8, TI::q81);
                    const uint32_t packed = xp_pack4(q8);
                    const int blk = head * 2 + (tid >> 5);
                    if ((tid & 3) == 0) xp_put_local(G + XP_G_ATT + head * 16 + (tid >> 2), etag, packed);
                    if ((tid & 31) == 0) { xp_put_local(G + XP_G_ATT + 256 + blk, etag, __float_as_uint(d8)); if (TI::q81) xp_put_local(G + XP_G_ATT + 288 + blk, etag, s8); }
                }
            }
        }
        XP_WALL(2);
        // ================= stage C: out_proj + bias + residual (biogpt.cpp:767-772) =================
        if (wave < 5) {
            uint32_t v[1];
            xp_sweep_q<RES, 1, 1, false>(G + XP_G_ATT + tid, TI::q81 || tid < 288, epoch, v, p, etag);      // (the block sums travel only with Q8_1 activations)
            if (tid < 256) s_xq[tid] = v[0];
            else if (tid < 288) s_xd[tid - 256] = __uint_as_float(v[0]);
            else s_xs[tid - 288] = v[0];
        }
        __syncthreads();
        XP_WALL(8);
        {
            uint32_t ax[8];
            const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
            ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
            const float axd = s_xd[sub];
            const uint32_t axs = s_xs[sub];
            float *const part = s_part + wave * 2 * OS * DEC_PS;
#pragma unroll
            for (int s = 0; s < OS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, EXPAND>(wo[s], ax, axd, __uint_as_float(axs), (int)axs);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 2 * OS) {
                const int lr = (lane >> 1) * 2 * NW + wave * 2 + (lane & 1), row = slot * 32 + lr;
                const float v = __fadd_rn(__fadd_rn(sum32_in_order(part + lane * DEC_PS), s_bias[192 + lr]), s_x[row]);
                if (SPLIT) xp_put(Y.gx1 + xp_col_slot(row), etag, __float_as_uint(v));       // the MLP half runs on the next XCD
                else xp_put_local(Y.gx1 + xp_col_slot(row), etag, __float_as_uint(v));
            }
        }
        XP_WALL(3);
        }   // FIRST
        if constexpr (SECOND) {
        // ================= stage D: LayerNorm -> Q8 -> fc1 -> GELU -> Q8 (biogpt.cpp:777-787) =================
        float4 x1v = make_float4(0.f, 0.f, 0.f, 0.f), lnw = x1v, lnb = x1v;
        if (wave < 4) {
            uint32_t v[4];
            xp_sweep_q<RES, 4, 256, true>(Y.gx1 + tid, true, epoch, v, p, etag);
            x1v = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            reinterpret_cast<float4 *>(s_x1)[tid] = x1v;
            XP_WALL(9);
            lnw = reinterpret_cast<const float4 *>(s_ln + 2048)[tid]; lnb = reinterpret_cast<const float4 *>(s_ln + 3072)[tid];
        }
        ln4_q8_1024<TI::q81, TI::q81>(x1v, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
        XP_WALL(10);
        {
            uint32_t ax[8];
            const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
            ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
            const float axd = s_xd[sub];
            const uint32_t axs = s_xs[sub];
            float *const part = s_part + wave * 2 * FS * DEC_PS;
#pragma unroll
            for (int s = 0; s < FS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, EXPAND>(w1[s], ax, axd, __uint_as_float(axs), (int)axs);
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
        XP_WALL(11);
        if (tid < 128) {
            int8_t q8; float d8; uint32_t s8;
            q8_block32(s_g[tid], TI::q81, q8, d8, s8, TI::q81);
            const uint32_t packed = xp_pack4(q8);
            const int blk = slot * 4 + (tid >> 5);
            if ((tid & 3) == 0) xp_put_local(G + XP_G_H + slot * 32 + (tid >> 2), etag, packed);
            if ((tid & 31) == 0) { xp_put_local(G + XP_G_H + 1024 + blk, etag, __float_as_uint(d8)); if (TI::q81) xp_put_local(G + XP_G_H + 1152 + blk, etag, s8); }
        }
        XP_WALL(4);
        // ================= stage E: fc2 + bias + residual (biogpt.cpp:790-795) =================
        {   // 1024 + 128 + 128 granules in ONE poll loop: every pass has all of a lane's loads in flight together
            constexpr int NQ = 1024 / NT;
            uint32_t v[NQ + 1];
#pragma unroll
            for (int k = 0; k <= NQ; k++) v[k] = 0u;
            const xp_u64 *g = G + XP_G_H + tid;
            const bool tail = tid < (TI::q81 ? 256 : 128);      // scales, and with Q8_1 activations the block sums
            for (uint32_t spins = 0; !RES || etag != 0u; spins++) {
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
                if (spins >= XP_SPIN_MAX) { if (lane == 0) xp_fail(p, 1u); if (RES) etag = 0u; break; }
                if ((spins & (RES ? 255u : 1023u)) == (RES ? 255u : 1023u) && __any(__hip_atomic_load(p.ctl + 1, XP_RLX) != 0u)) { if (RES) etag = 0u; break; }
                
            }
#pragma unroll
            for (int k = 0; k < NQ; k++) s_hq[tid + k * NT] = v[k];
            if (tid < 128) s_hd[tid] = __uint_as_float(v[NQ]);
            else if (tid < 256) s_hs[tid - 128] = v[NQ];
        }
        __syncthreads();
        XP_WALL(12);
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
                for (int r = 0; r < F2R; r++) part[r * DEC_PS2 + u] = xp_dot<WT, EXPAND>(w2[r][it], ax, axd, __uint_as_float(axs), (int)axs);
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
                xp_put(Y.gx + xp_col_slot(row), etag, __float_as_uint(v));
                if (L == p.n_layer - 1) XPK(x_final)[row] = v;
            }
        }
        XP_WALL(5);
        if (L == p.n_layer - 1 && slot == 0) XP_TAIL(tk, 0);
        }   // SECOND
        __syncthreads();       // s_ln / s_bias / s_x1 are rewritten by the next unit of this XCD
    }
    // ================= final LayerNorm + lm_head (biogpt.cpp:799-811): the XCDs that are done with their layers =================
    // Their weights (four 64-row blocks = 16 units per lane) are loaded as soon as the workgroup's last layer of this token is
    // finished -- 1 .. 6 layers before the last layer's output exists -- so the logits cost one hop + LayerNorm + 16 block dots
    // instead of a launch boundary plus a 24.6 MB stream.  XCD 0 (next token's layer 0) and the last layer's XCD take no part.
    // Round 4 (XP_LM_LATE_XCD): the XCD of the last-but-one unit takes no part either -- it is done with its unit only ~5 us before the last layer's output
    // exists, its rows were still arriving then, and its workgroups delivered their logits (and the partials the next token's sampler waits for) 2.5 us
    // behind everyone else (profiles/api_loop_device_clock_r4.txt: "XCD group of the last one: 0 0 0 0 0 64").  Their share goes to workgroups 16 .. 31 of
    // XCD 0, which have been idle since that XCD's last unit (7 units ago) and are only needed again for the next token's out_proj.
    const int lm_rank = xp_lm_rank(xcd, slot, n_units);
    if (p.lm != 0 && lm_rank >= 0 && lm_rank * 4 < XPK(lm_blocks)) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const int sub = lane & 31, rsub = lane >> 5;
        const bool worker = tid < 256;
        constexpr int LMS = 128 / NW;                                         // 2-row steps per wave: 256 rows per workgroup
        const int row0 = lm_rank * 256;
        // a resident pass with an odd sequence number writes the alternate row / partial buffers (XpParams::spec_rec)
        const bool alt = RES && p.resident != 0 && ((XPK(mbox_seq0) + (uint32_t)tk) & 1u) != 0u;
        float *const lg_dev = alt ? XPK(logits_alt) : XPK(logits);
        float *const lg_host = alt ? XPK(logits_host_alt) : XPK(logits_host);
        Unit<WT> wl[LMS];
#pragma unroll
        for (int s = 0; s < LMS; s++) {
            const int row = row0 + s * 2 * NW + wave * 2 + rsub;
            if (row < XPK(n_vocab)) load_unit<WT>(wl[s], XPK_MATRIX(Wlm), (int64_t)row * 32 + sub);
            else { wl[s].q0 = make_uint4(0u, 0u, 0u, 0u); wl[s].q1 = wl[s].q0; wl[s].sc = 0u; wl[s].qh = 0u; }
        }
#pragma unroll
        for (int s = 0; s < LMS; s++) xp_settle<WT, EXPAND>(wl[s]);
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), lnw = xv, lnb = xv;
        if (worker) { lnw = reinterpret_cast<const float4 *>(XPK(lm_ln_w))[tid]; lnb = reinterpret_cast<const float4 *>(XPK(lm_ln_b))[tid]; }
        if (RES && p.resident != 0 && tk > 0 && !(XPK(res_dbg) & 64)) {      // "the rows of this pass may be written" (published by XCD 0 at the start of the pass, long ago: one poll)
            uint32_t go[1];
            xp_sweep_q<RES, 1>(XPK(samp) + 2049, lane == 0, epoch, go, p, etag);
        }
        if (wave < 4) {
            uint32_t v[4];
            xp_sweep_q<RES, 4, 256, true>(XPL(p.n_layer - 1).gx + tid, true, epoch, v, p, etag);
            xv = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            if (lm_rank == 0) XP_TAIL(tk, 1);
        }
        ln4_q8_1024<TI::q81, TI::q81>(xv, lnw, lnb, p.eps, s_red, s_xq, s_xd, s_xs);
        if (lm_rank == 0) XP_TAIL(tk, 2);
        uint32_t ax[8];
        const uint4 a = *reinterpret_cast<const uint4 *>(s_xq + sub * 8), b = *reinterpret_cast<const uint4 *>(s_xq + sub * 8 + 4);
        ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w; ax[4] = b.x; ax[5] = b.y; ax[6] = b.z; ax[7] = b.w;
        const float axd = s_xd[sub];
        const uint32_t axs = s_xs[sub];
        float *const part = s_part + wave * 2 * LMS * DEC_PS;
#pragma unroll
        for (int s = 0; s < LMS; s++) part[(s * 2 + rsub) * DEC_PS + sub] = xp_dot<WT, EXPAND>(wl[s], ax, axd, __uint_as_float(axs), (int)axs);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lane < 2 LMS finishes local row (lane >> 1) 2 NW + 2 wave + (lane & 1); lanes 64 / NW * j .. hold rows of block j
        float best_val = -INFINITY;
        int best_idx = 0x7fffffff;
        if (lane < 2 * LMS) {
            const int row = row0 + (lane >> 1) * 2 * NW + wave * 2 + (lane & 1);
            if (row < XPK(n_vocab)) {
                const float v = sum32_in_order(part + lane * DEC_PS);
                if constexpr (RES) s_S[row - row0] = v;        // staged for the copies below (s_S: no attention runs in this workgroup now)
                else { XPK(logits)[row] = v; if (XPK(logits_host)) XPK(logits_host)[row] = v; }
                best_val = v; best_idx = row;
            }
        }
        if (lm_rank == 0) XP_TAIL(tk, 3);
        // per-block partial arg-max (lowest index wins ties): groups of 64 / NW lanes, then the NW waves through LDS
        static_assert(64 / NW == 8 || 64 / NW == 4, "finisher lanes per block");
        xp_argmax_dpp<DPP_QUAD_XOR1>(best_val, best_idx); xp_argmax_dpp<DPP_QUAD_XOR2>(best_val, best_idx);
        if (64 / NW == 8) xp_argmax_dpp<DPP_ROW_HALF_MIRROR>(best_val, best_idx);
        constexpr int LPB = 64 / NW;                                          // finisher lanes per 64-row block in one wave
        if (lane < 2 * LMS && (lane & (LPB - 1)) == 0) { s_redf[(lane / LPB) * NW + wave] = best_val; s_redi[(lane / LPB) * NW + wave] = best_idx; }
        bool row_ok = true;
        if constexpr (RES) row_ok = __syncthreads_and(etag != 0u);
        else __syncthreads();
        // the host's copy of the row (biogpt_eval's output): wave 1 writes the workgroup's 256 logits as ONE kilobyte of 16-byte write-through stores
        // (four-byte stores from the finisher lanes were one PCIe write each: 42 k per token); a resident launch follows them with the workgroup's
        // completion word for this token -- same wave, behind its own stores (the host collects one word per lm_head workgroup)
        if (RES && lg_host && wave == 1 && row_ok) {
            const int r = row0 + 4 * lane;
            if (XPK(res_dbg) & 4) {
            } else if (r + 3 < XPK(n_vocab)) {
                const xp_v4f v4 = *reinterpret_cast<const xp_v4f *>(s_S + 4 * lane);
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(lg_host + r), "v"(v4) : "memory");
                *reinterpret_cast<xp_v4f *>(lg_dev + r) = v4;
            } else {
                for (int j = r; j < XPK(n_vocab); j++) { __hip_atomic_store(lg_host + j, s_S[j - row0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); lg_dev[j] = s_S[j - row0]; }
            }
            if (p.resident != 0) {
                if (lane < 4 && lm_rank * 4 + lane < XPK(lm_blocks)) {      // the block maxima behind the row (the same maxima tid < 4 records below)
                    float bm = s_redf[lane * NW];
#pragma unroll
                    for (int w = 1; w < NW; w++) bm = fmaxf(bm, s_redf[lane * NW + w]);
                    __hip_atomic_store(lg_host + xp_blockmax_offset(XPK(n_vocab)) + lm_rank * 4 + lane, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (!(XPK(res_dbg) & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(XPK(done_host) + lm_rank, XPK(mbox_seq0) + (uint32_t)tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((XPK(res_dbg) & 32) && XPK(wall) && lane == 0 && lm_rank == 0) XPK(wall)[(size_t)((XPK(mbox_seq0) + (uint32_t)tk) & 4095u) * 2 + 1] = wall_clock64();
                if ((XPK(res_dbg) & 32) && XPK(wall) && lane == 0) XPK(wall)[8192 + (size_t)((XPK(mbox_seq0) + (uint32_t)tk) & 63u) * 256 + lm_rank] = wall_clock64();
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
                // the device row and the partials, like the host row, only from a pass that really ran (a draining launch -- the host asked it to leave, or it gave up
                // waiting -- still walks through the pass it was about to start, with nothing valid in its hands)
                if (RES && !row_ok) {}
                else if (RES && alt) { XPK(pmax_alt_val)[blk] = bv; XPK(pmax_alt_idx)[blk] = bi; }
                else { XPK(pmax_out_val)[blk] = bv; XPK(pmax_out_idx)[blk] = bi; }
                if (tk + 1 < p.n_tok) {        // the sampler of the next token runs on XCD 0
                    xp_put(XPK(samp) + blk, etag, __float_as_uint(bv));
                    xp_put(XPK(samp) + 1024 + blk, etag, (uint32_t)bi);
                }
                if (RES && (XPK(res_dbg) & 32) && XPK(wall) && blk == 0) XPK(wall)[32768 + (size_t)((XPK(mbox_seq0) + (uint32_t)tk) & 4095u) * 4 + 3] = wall_clock64();
            }
        }
        if (lm_rank == 0) XP_TAIL(tk, 4);
        __syncthreads();       // s_redf / s_part / s_xq are rewritten by this workgroup's next layer
    }
    }   // tokens
    // every workgroup read the launch counter and the position when it started, long before the last layer's output existed
    if (threadIdx.x == 0) {
        if (xcd == last_xcd && slot == 0) {
            // tags must never wrap onto a stale granule: after 4.0e9 tokens (two weeks of continuous decode) the context leaves this
            // path through the ordinary error route (the host repeats the call on the five-launch layer and stays there)
            if (epoch0 + (uint32_t)p.n_tok > 0xF0000000u) xp_fail(p, 5u);
            __hip_atomic_store(p.ctl, epoch0 + (uint32_t)p.n_tok, XP_RLX);
            __hip_atomic_store(p.ctl + 2, __hip_atomic_load(p.ctl + 2, XP_RLX) + 1u, XP_RLX);
        }
        if (xcd == (last_xcd == 1 ? 2 : 1) && slot == 0 && p.adv != 0) { XPK(st)->n_past = n_past0 + p.n_tok; XPK(st)->n_gen = n_gen0 + p.n_tok; }
    }
}

template <int WT, int LPK, int NW, int KCAP, bool SPLIT, bool RES = false>
__global__ __launch_bounds__(NW * 64) void dec_xpipe_kernel(const XpParams p) {
    using TI = TypeInfo<WT>;
    static_assert(TI::quant && (WT != W_Q8_0 || SPLIT), "Q8_0 (9 registers per weight unit) runs with split layers");
    static_assert(LPK == 2 || LPK == 4 || LPK == 8 || LPK == 16, "lanes per key");
    static_assert(NW == 8 || NW == 16, "waves per workgroup");
    constexpr int D = 1024, DK = 64, NT = NW * 64;
    constexpr int QS = 96 / NW, OS = 16 / NW, FS = 64 / NW, F2R = 32 / NW;     // 2-row steps of qkv / out_proj / fc1 per wave; fc2 rows per wave
    static_assert(KCAP % NW == 0 && (KCAP > 256 ? KCAP / 2 : KCAP) <= NW * 64 / LPK, "key capacity of the launch");
    constexpr int NF4 = 16 / LPK, NV = (KCAP > 256 ? KCAP / 2 : KCAP) / NW;        // float4 of a key row per lane; values per lane (key slices of NW)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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

    // Which XCD am I on, and which of its 32 workgroups am I ?  HW_REG_XCC_ID says where; a per-XCD ticket (monotonic across
    // launches: launch n hands out 32 (n - 1) .. 32 n - 1) says which.  The dispatcher deals workgroups round-robin over the
    // XCDs, so a launch that has the device to itself gets exactly 32 per XCD whatever the starting point; a launch interleaved
    // with another stream's workgroups may not -- then a 33rd arrival raises the error word and the launch drains.
    const uint32_t epoch0 = __hip_atomic_load(p.ctl, XP_RLX);
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;     // HW_REG_XCC_ID, bits 0..3
        const uint32_t t = __hip_atomic_fetch_add(p.ctl + 8 + xcc, 1u, XP_RLX);
        s_redi[0] = (int)xcc;
        s_redi[1] = (int)(t - 32u * (__hip_atomic_load(p.ctl + 2, XP_RLX) - 1u));
    }
    __syncthreads();
    const int xcd = __builtin_amdgcn_readfirstlane(s_redi[0]), slot = __builtin_amdgcn_readfirstlane(s_redi[1]);
    __syncthreads();
    if ((unsigned)slot >= 32u) { if (threadIdx.x == 0) xp_fail(p, 2u); return; }
    const int n_past0 = (RES && p.resident != 0) ? p.res_n_past0 : p.st->n_past, n_gen0 = p.st->n_gen;
    const int t_cap = p.t_cap;

    // ggml_gelu's fp16 table (biogpt.cpp:784): 70 KB of it cover every argument for which GELU is neither the identity (x >= 3.38
    // in fp16) nor -0 (x <= -5.42): kept in LDS for the whole launch, fc1's 128 rows per workgroup look it up there
    if (!SPLIT || (xcd & 1)) {      // split layers: only the MLP halves (odd XCDs) look GELU up; XCD 0 starts layer 0 without the copy
        const uint4 *src = reinterpret_cast<const uint4 *>(p.gelu_tab);
        const int np8 = p.gelu_p / 8, nn8 = p.gelu_n / 8;
        for (int i = threadIdx.x; i < np8; i += NT) reinterpret_cast<uint4 *>(s_gelu)[i] = src[i];
        for (int i = threadIdx.x; i < nn8; i += NT) reinterpret_cast<uint4 *>(s_gelu + p.gelu_p)[i] = src[0x8000 / 8 + i];
    }
    if (SPLIT && !(xcd & 1) && (slot < 16 || KCAP > 256) && p.exp_n > 0) {      // the attention workgroups (512-key variant: all 32 of the XCD): the exp table's slice where the MLP halves keep GELU's
        const uint4 *src = reinterpret_cast<const uint4 *>(p.exp_tab + 0x8000);
        for (int i = threadIdx.x; i < p.exp_n / 8; i += NT) reinterpret_cast<uint4 *>(s_gelu)[i] = src[i];
    }
#ifdef XP_ONLY_ROLE      // (register census of one role: hipcc -Rpass-analysis=kernel-resource-usage -DXP_ONLY_ROLE=r; not a working kernel)
    xp_run<WT, LPK, NW, KCAP, XP_ONLY_ROLE, true, RES>(p, smem, xcd, slot, epoch0, n_past0, n_gen0);
#else
    if constexpr (SPLIT) {
        if (xcd & 1) { xp_run<WT, LPK, NW, KCAP, 2, true, RES>(p, smem, xcd, slot, epoch0, n_past0, n_gen0); return; }
    }
    if (slot < 16) xp_run<WT, LPK, NW, KCAP, 0, SPLIT, RES>(p, smem, xcd, slot, epoch0, n_past0, n_gen0);
    else xp_run<WT, LPK, NW, KCAP, 1, SPLIT, RES>(p, smem, xcd, slot, epoch0, n_past0, n_gen0);
#endif
}

// Calibration of the cross-XCD hand-off regions (xpipe_place_hops, once per context): ONE lane on XCD src and ONE on XCD dst play ping-pong over the granule at
// `addr` (ping) and addr + 1 (pong) with the stores and polls of the real hand-offs (write-through stores, agent-scope loads); ticks[i] = 100 MHz ticks of `reps`
// round trips.  256 workgroups: on every XCD the first arrival is the pinger of the probes that start there, the second the ponger of those that end there, all
// others leave.  Probe values carry 0xFFFF in their upper half: no hand-off tag ever gets there (xp_run gives up at 0xF0000000); the host zeroes the regions afterwards.
struct XpProbe { unsigned long long *addr; int32_t src, dst; };
__global__ void xp_hop_probe_kernel(const XpProbe *pr, int n, int reps, uint32_t *tickets, unsigned long long *ticks, uint32_t *err) {
    if (threadIdx.x != 0) return;
    const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);
    const uint32_t role = __hip_atomic_fetch_add(tickets + xcc, 1u, XP_RLX);
    if (role > 1u) return;
    for (int i = 0; i < n; i++) {
        const XpProbe q = pr[i];
        if ((role == 0u ? q.src : q.dst) != xcc) continue;
        xp_u64 *X = q.addr, *Y = q.addr + 1;
        const xp_u64 base = 0xFFFF000000000000ull | ((xp_u64)i << 16);
        const unsigned long long t0 = wall_clock64();
        for (int r = 1; r <= reps; r++) {
            const xp_u64 want = base + (xp_u64)r;
            if (role == 0u) __hip_atomic_store(X, want, XP_RLX);
            const xp_u64 *w = role == 0u ? Y : X;
            for (uint32_t spins = 0; __hip_atomic_load(w, XP_RLX) != want; spins++)
                if (spins > XP_SPIN_MAX || ((spins & 1023u) == 1023u && __hip_atomic_load(err, XP_RLX) != 0u)) { __hip_atomic_store(err, 1u, XP_RLX); return; }
            if (role == 1u) __hip_atomic_store(Y, want, XP_RLX);
        }
        if (role == 0u) ticks[i] = wall_clock64() - t0;
    }
}

// where workgroup b of a 256-workgroup launch runs: the host checks b % 8 once per device before it trusts the pipeline
__global__ void xp_probe_kernel(uint32_t *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;
}

}  // namespace bgk
