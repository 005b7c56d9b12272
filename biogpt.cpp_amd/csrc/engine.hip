// MI355X-native BioGPT decoder engine: weight arena + repack, launch sequence, hipGraph capture,
// and the extern "C" entry points declared in include/biogpt_hip.h.
//
// Reference path replaced: biogpt_model_load (biogpt.cpp:27-453), biogpt_graph (:624-810),
// biogpt_eval (:812-847) and the greedy decode loop of examples/main/main.cpp:91-151.
//
// Device memory (sized for 288 GB of HBM3E, nothing is streamed or paged):
//   arena     all weights, repacked to the SoA block layout of kernels.hip.h, plus the two fp16
//             tables; ONE contiguous allocation so that a multi-GPU launcher can broadcast it with a
//             single RCCL collective
//   memory_k / memory_v   F32 [n_layer][n_positions][d_model], as biogpt.cpp:331-335 (F4)
//   scratch   x, x1, q, att [n_positions][d_model], h [n_positions][d_ff], logits
//   state     DevState + token ring (kernels read n_past / ids from HBM so the captured decode graph
//             advances itself)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "host_common.h"
#include "kernels.hip.h"
#include "kernels_fast.hip.h"
#include "kernels_decode.hip.h"
#include "kernels_fdecode.hip.h"
#include "kernels_lmhead.hip.h"
#include "kernels_sweep.hip.h"
#include "kernels_xlong.hip.h"   // parameter blocks and layouts only: the pipelined kernels are instantiated in xpipe_tu.hip
#include "kernels_xcols.hip.h"   // (likewise: xcols_tu.hip)
#include "kernels_fpipe.hip.h"   // (likewise: fpipe_tu.hip)
#include "kernels_quant.hip.h"
#include "model_file.h"
#include "quant_host.h"

using namespace bg;

#define HIP_TRY(ret, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) BG_FAIL(ret, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

namespace {

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) & ~(ALIGN - 1); }

// ---- arena plan -----------------------------------------------------------------------------------
struct MatSlot {  // offsets of one (possibly row-fused) matrix inside the arena
    int32_t type = 0;
    int64_t M = 0, K = 0;
    size_t qs = 0, sc = 0, qh = 0;
    size_t qs_bytes = 0, sc_bytes = 0, qh_bytes = 0;
    size_t xq = 0, xs = 0;      // offsets in the expanded row-tiled image of the MFMA kernels (ensure_tile_images; kernels_mfma.hip.h)
};

struct LayerSlots {
    MatSlot qkv, o, fc1, fc2;
    size_t qkv_b, o_b, ln0_w, ln0_b, ln1_w, ln1_b, fc1_b, fc2_b;
};

struct ArenaPlan {
    MatSlot lm_head, embed_tokens, embed_pos;
    size_t ln_w = 0, ln_b = 0, gelu_tab = 0, exp_tab = 0;
    std::vector<LayerSlots> layers;
    size_t total = 0;
};

size_t unit_bytes_qs(int32_t t, int64_t k) {  // bytes of the qs stream per row
    switch (t) {
        case T_F32: return (size_t)k * 4;
        case T_F16: return (size_t)k * 2;
        case T_Q8_0: return (size_t)(k / QK) * 32;
        default: return (size_t)(k / QK) * 16;
    }
}
size_t unit_bytes_sc(int32_t t, int64_t k) {
    switch (t) {
        case T_Q4_0: case T_Q5_0: case T_Q8_0: return (size_t)(k / QK) * 2;
        case T_Q4_1: case T_Q5_1: return (size_t)(k / QK) * 4;
        default: return 0;
    }
}
size_t unit_bytes_qh(int32_t t, int64_t k) { return (t == T_Q5_0 || t == T_Q5_1) ? (size_t)(k / QK) * 4 : 0; }

ArenaPlan plan_arena(const biogpt_hip_hparams &hp, int64_t pos_rows) {
    ArenaPlan pl;
    const int32_t wt = ftype_to_type(hp.ftype);
    const int64_t D = hp.d_model, F = hp.d_ff, V = hp.n_vocab;
    size_t off = 0;
    auto mat = [&](MatSlot &m, int64_t M, int64_t K) {
        m.type = wt; m.M = M; m.K = K;
        m.qs_bytes = unit_bytes_qs(wt, K) * (size_t)M;
        m.sc_bytes = unit_bytes_sc(wt, K) * (size_t)M;
        m.qh_bytes = unit_bytes_qh(wt, K) * (size_t)M;
        m.qs = off; off = align_up(off + m.qs_bytes);
        m.sc = off; off = align_up(off + m.sc_bytes);
        m.qh = off; off = align_up(off + m.qh_bytes);
    };
    auto vec = [&](size_t &slot, int64_t n) { slot = off; off = align_up(off + (size_t)n * 4); };
    mat(pl.embed_tokens, V, D);
    mat(pl.embed_pos, pos_rows, D);
    pl.layers.resize((size_t)hp.n_layer);
    for (auto &L : pl.layers) {
        vec(L.ln0_w, D); vec(L.ln0_b, D);
        mat(L.qkv, 3 * D, D); vec(L.qkv_b, 3 * D);
        mat(L.o, D, D); vec(L.o_b, D);
        vec(L.ln1_w, D); vec(L.ln1_b, D);
        mat(L.fc1, F, D); vec(L.fc1_b, F);
        mat(L.fc2, D, F); vec(L.fc2_b, D);
    }
    vec(pl.ln_w, D); vec(pl.ln_b, D);
    mat(pl.lm_head, V, D);
    pl.gelu_tab = off; off = align_up(off + 65536 * 2);
    pl.exp_tab = off; off = align_up(off + 65536 * 2);
    pl.total = off;
    return pl;
}

// ---- repack one tensor (file layout -> SoA device layout) into host staging buffers ---------------
void repack_rows(int32_t type, const uint8_t *src, int64_t rows, int64_t K, uint8_t *qs, uint8_t *sc, uint8_t *qh) {
    if (type == T_F32 || type == T_F16) {
        std::memcpy(qs, src, file_row_bytes(type, K) * (size_t)rows);
        return;
    }
    const size_t bb = file_block_bytes(type);
    const int64_t nblocks = rows * (K / QK);
    const unsigned nw = nblocks > (1 << 16) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1;
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < nw; w++) {
        const int64_t b0 = nblocks * w / nw, b1 = nblocks * (w + 1) / nw;
        auto job = [=] {
            for (int64_t b = b0; b < b1; b++) {
                const uint8_t *p = src + (size_t)b * bb;
                switch (type) {
                    case T_Q4_0: std::memcpy(sc + b * 2, p, 2); std::memcpy(qs + b * 16, p + 2, 16); break;
                    case T_Q4_1: std::memcpy(sc + b * 4, p, 4); std::memcpy(qs + b * 16, p + 4, 16); break;
                    case T_Q5_0: std::memcpy(sc + b * 2, p, 2); std::memcpy(qh + b * 4, p + 2, 4); std::memcpy(qs + b * 16, p + 6, 16); break;
                    case T_Q5_1: std::memcpy(sc + b * 4, p, 4); std::memcpy(qh + b * 4, p + 4, 4); std::memcpy(qs + b * 16, p + 8, 16); break;
                    case T_Q8_0: std::memcpy(sc + b * 2, p, 2); std::memcpy(qs + b * 32, p + 2, 32); break;
                    default: break;
                }
            }
        };
        if (nw == 1) job(); else pool.emplace_back(job);
    }
    for (auto &t : pool) t.join();
}

inline float gelu_tanh_f32(float x) {  // ggml_gelu_f32 [SURVEY A.5]
    const float GELU_COEF_A = 0.044715f, SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    return 0.5f * x * (1.0f + tanhf(SQRT_2_OVER_PI * x * (1.0f + GELU_COEF_A * x * x)));
}

int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
int ilog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }

int env_int(const char *name, int dflt) {
    const char *s = std::getenv(name);
    return s ? std::atoi(s) : dflt;
}

}  // namespace

// Tuning / debugging switches.  Read from the environment ONCE, when a context is created (and again only on
// biogpt_hip_refresh_options): no getenv on any launch path.
struct EngineOptions {
    int mv_waves, max_wgs, lm_steps, fast_steps, no_fast, no_chain, mfma_min_cols, attn_group_min,
        split_min, attn_slim_min, dbg, target_wgs, prompt_cols, no_graph, causal, no_fused_decode, fc1_blocks, fc2_waves, oproj_waves, attn_tile, eval_graph_split, qkv_waves, fc1_waves, attn_waves, xpipe, xpipe_fault, xpipe_tables, xpipe_lm, xpipe_multi, xpipe_long, xpipe_dual, xpipe_as_res, proc_lock, graph_contended, fault_stale, xcols, resident, resident_us, res_dbg, res_spec, no_fdec, lm_stream, hop_place, verbose, topk_blocks, eval_sync, fpipe, fpipe_stamps, fpipe_lead, fpipe_fault;
    void load() {
        auto get = [](const char *name, int dflt) { return env_int(name, dflt); };
        mv_waves = get("BIOGPT_HIP_MV_WAVES", 4);
        max_wgs = get("BIOGPT_HIP_MAX_WGS", 1024);
        lm_steps = get("BIOGPT_HIP_LM_STEPS", 8);
        fast_steps = get("BIOGPT_HIP_FAST_STEPS", 1);
        no_fast = get("BIOGPT_HIP_NO_FAST", 0);
        no_chain = get("BIOGPT_HIP_NO_CHAIN", 0);
        no_fdec = get("BIOGPT_HIP_NO_FDEC", 0);             // 1: float-weight decode mat-vecs on the generic kernel (A/B arm of kernels_fdecode.hip.h)
        mfma_min_cols = get("BIOGPT_HIP_MFMA_MIN_COLS", -1);      // -1: measured cross-overs (48 decode columns / 64 prompt columns)
        attn_group_min = get("BIOGPT_HIP_ATTN_GROUP_MIN", 80);
        split_min = get("BIOGPT_HIP_SPLIT_MIN", 256);
        attn_slim_min = get("BIOGPT_HIP_ATTN_SLIM_MIN", 48);
        dbg = get("BIOGPT_HIP_DBG", 0);
        target_wgs = get("BIOGPT_HIP_TARGET_WGS", 256);
        prompt_cols = get("BIOGPT_HIP_PROMPT_COLS", 512);
        no_graph = get("BIOGPT_HIP_NO_GRAPH", 0);
        causal = get("BIOGPT_HIP_CAUSAL", 0);
        no_fused_decode = get("BIOGPT_HIP_NO_FUSED_DECODE", 0);
        fc1_blocks = get("BIOGPT_HIP_FC1_BLOCKS", 1);
        fc2_waves = get("BIOGPT_HIP_FC2_WAVES", 8);      // same-process sweeps (profiles/decode_shapes_r2.txt): 8 waves beat 16 and 4
        oproj_waves = get("BIOGPT_HIP_OPROJ_WAVES", 8);
        qkv_waves = get("BIOGPT_HIP_QKV_WAVES", 8);
        fc1_waves = get("BIOGPT_HIP_FC1_WAVES", 8);
        attn_waves = get("BIOGPT_HIP_ATTN_WAVES", 8);
        xpipe_tables = get("BIOGPT_HIP_XPIPE_TABLES", 3);   // bit 0: the MLP halves keep the GELU table's non-trivial slices in LDS (70 KB); bit 1: the attention workgroups the exp table's (39 KB)
        resident = get("BIOGPT_HIP_RESIDENT", 1);           // biogpt_hip_eval with one token: the pipelined launch stays on the device and takes the next call's token from a pinned mailbox
        res_dbg = get("BIOGPT_HIP_RES_DBG", 0);              // measurement only (kernels_xpipe.hip.h XpParams::res_dbg)
        lm_stream = get("BIOGPT_HIP_LM_STREAM", 1);          // the stand-alone lm_head as lm_stream_kernel (0: matvec_fast_kernel<PRO_LN, EPI_LOGITS>)
        res_spec = get("BIOGPT_HIP_SPEC", 1);                // a resident launch may start the next token from its own arg-max (greedy callers; resident_eval)
        resident_us = get("BIOGPT_HIP_RESIDENT_US", 1000);   // ... for at most this long without a new token (the device is not shared meanwhile)
        hop_place = get("BIOGPT_HIP_HOP_PLACE", 1);        // the cross-XCD hand-off regions: 1 placed by a calibration launch (xpipe_place_hops), 0 first candidate, 2 the slower one (A/B)
        verbose = get("BIOGPT_HIP_VERBOSE", 0);
        eval_sync = get("BIOGPT_HIP_EVAL_SYNC", 0);
        topk_blocks = get("BIOGPT_HIP_TOPK_BLOCKS", 1);    // biogpt_hip_eval_topk behind a resident launch: select from the blocks whose maximum can hold a candidate (0: scan the whole row)
        xpipe_dual = get("BIOGPT_HIP_XPIPE_DUAL", 1);       // contexts of 257 .. 512 keys (multi-token launches, graph replays, resident launches): dec_xpipe_kernel with two workgroups per head (0: kernels_xlong.hip.h, as in round 3)
        fpipe_lead = get("BIOGPT_HIP_FPIPE_LEAD", -1);       // the persistent float launch: 64-clock units a polling wave lets pass between its own workgroup's publication and its first sweep (-1: 26 for F32 files, 16 for F16 -- the measured optima, profiles/fpipe_lead_scan_r6.txt)
        fpipe_fault = get("BIOGPT_HIP_FPIPE_FAULT", 0);     // test hook: the context's first persistent float launch never gets one workgroup's out_proj rows and drains with an error
        fpipe_stamps = get("BIOGPT_HIP_FPIPE_STAMPS", 0);   // diagnostics: the persistent launch records stage-border times of three workgroups (biogpt_hip_fpipe_stamps)
        fpipe = get("BIOGPT_HIP_FPIPE", 1);                 // single-token steps of F32 / F16 files as ONE persistent launch for all layers (kernels_fpipe.hip.h); 0: five launches per layer
        xcols = get("BIOGPT_HIP_XCOLS", 1);                 // evals of 2 .. 8 tokens (the reference's prompt chunks) as ONE persistent launch, one column per XCD (kernels_xcols.hip.h); 0: the launch chain of kernels_fast.hip.h
        xpipe_long = get("BIOGPT_HIP_XPIPE_LONG", 1);       // contexts of 257 .. 1024 keys on the pipeline too (kernels_xlong.hip.h: attention spread over the chip)
        xpipe_multi = get("BIOGPT_HIP_XPIPE_MULTI", 1);     // biogpt_hip_generate_greedy: all tokens of a context bucket in one pipelined launch
        xpipe_lm = get("BIOGPT_HIP_XPIPE_LM", 1);           // final LayerNorm + lm_head inside the pipelined launch
        xpipe_fault = get("BIOGPT_HIP_XPIPE_FAULT", 0);   // test hook: the first pipelined launch finds a 33rd workgroup on XCD 0 and drains
        xpipe = get("BIOGPT_HIP_XPIPE", 1);             // the XCD-pipelined single-launch decode step (kernels_xpipe.hip.h)
        attn_tile = get("BIOGPT_HIP_ATTN_TILE", 1);
        graph_contended = get("BIOGPT_HIP_GRAPH_CONTENDED", 0);   // test switch: replay the five-launch eval graph even while ANOTHER context holds the pipeline slot (the arrangement of profiles/two_contexts_r4.txt)
        fault_stale = get("BIOGPT_HIP_FAULT_STALE", 0);           // test switch: every k-th replayed eval starts from the PREVIOUS mailbox slot (the stale-row symptom, injected)
        proc_lock = get("BIOGPT_HIP_PROC_LOCK", 1);               // one process per device drives the pipelined launches (engine_xpipe.inc, xpipe_process_lock)
        xpipe_as_res = get("BIOGPT_HIP_XPIPE_AS_RES", 0);          // measurement only: ordinary pipelined launches through the RES instantiations (tests/test_gpu_resident.py)
        eval_graph_split = get("BIOGPT_HIP_EVAL_GRAPH_SPLIT", 0);   // 0: per entry point (eval_topk: one graph, eval: two segments)
    }
    int mfma_min(int dflt) const { return mfma_min_cols >= 0 ? mfma_min_cols : dflt; }
};

namespace {
}  // namespace

// ---- context ----------------------------------------------------------------------------------------
struct biogpt_hip_ctx {
    EngineOptions opt{};
    biogpt_hip_hparams hp{};
    int device = 0;
    int n_tensors = 0;
    int64_t pos_rows = 0;
    std::vector<std::string> vocab, merges;
    biogpt_hip_vocab *tok_vocab = nullptr;   // token <-> id maps + merge ranks for the tokenizer entry points

    uint8_t *arena = nullptr;
    size_t arena_bytes = 0;
    bool owns_arena = false;
    ArenaPlan plan;

    float *memory_k = nullptr, *memory_v = nullptr;
    float *x = nullptr, *x1 = nullptr, *q = nullptr, *att = nullptr, *h = nullptr;
    // producer-quantized activations [columns][K]: [0] attention out (d_model), [1] fc1 out (d_ff), [2] LayerNorm out (d_model)
    int8_t *aq_q[3] = {nullptr, nullptr, nullptr};
    float *aq_d[3] = {nullptr, nullptr, nullptr};
    uint32_t *aq_s[3] = {nullptr, nullptr, nullptr};
    float *logits = nullptr;      // [n_vocab]
    float *logits_host = nullptr; // pinned staging for biogpt_hip_eval
    float *logits_all = nullptr;  // lazily [n][n_vocab]
    size_t logits_all_rows = 0;
    float *pmax_val = nullptr;
    float *sp_scores = nullptr, *sp_max = nullptr;   // key-split decode attention scratch (kernels_fast.hip.h)
    bool tile_img_failed = false;  // the image did not fit: passes stay on the VALU chain
    uint8_t *tile_img = nullptr;   // row-tiled copy of the chain matrices for the MFMA kernels (same offsets as the arena), built on first use
    double *sp_pv = nullptr;
    int32_t *pmax_idx = nullptr;
    int pmax_cap = 0;
    bgk::DevState *state = nullptr;  // device
    uint8_t *state_host = nullptr;   // pinned upload ring: STATE_SLOTS x (header + tokens)
    size_t state_bytes = 0;
    size_t slot_bytes = 0;
    int slot_idx = 0;

    // batched multi-sequence decode (biogpt_hip_generate_greedy_batch): per-sequence caches + state
    float *bk = nullptr, *bv = nullptr;   // [cap][n_layer][n_head][n_positions][dk]
    bgk::SeqState *seq = nullptr;         // [cap]
    bgk::SeqState *cols = nullptr;        // column states of a multi-sequence prompt pass
    size_t cols_cap = 0;
    int32_t *seq_gen = nullptr;           // [cap][n_positions]
    int batch_cap = 0;
    std::set<const void *> lds_attr_done;     // kernels whose > 64 KB dynamic-LDS opt-in attribute is set on this device
    unsigned long long *tstamp = nullptr;     // profiling only (opt.dbg & 32)
    int launch_parity = 0;
    hipGraphExec_t graph_batch[12] = {};  // [6 * (steps as column-per-XCD launches) + context bucket], captured for graph_batch_n sequences
    int graph_batch_n = 0;

    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipGraphExec_t graph_step[2][2][6] = {};  // [pipelined][advance][context bucket: 64,128,192,256,512,P keys]; pipelined = captured with the XCD-pipelined launch (replayed only while this context holds the device's pipeline slot)
    hipGraphExec_t graph_eval[2][2][6][4] = {};   // [pipelined][form][bucket][segment] single-token biogpt_hip_eval*: the decode step with the token taken
                                               // from the state; form 0 = one graph, form 1 = a short first segment + the rest
    int graph_eval_segs[2] = {0, 0};
    // XCD-pipelined decode step (kernels_xpipe.hip.h): layer table, hand-off granules, {launch counter, error word}, pinned error mirror
    bgk::XpLayer *xp_layers = nullptr;
    bgk::xp_u64 *xp_gran = nullptr;
    bgk::xp_u64 *xp_hop = nullptr;         // the hand-off regions that cross XCDs (x1 and x of every layer): two 8 KB candidates each, xpipe_place_hops picks
    bgk::xp_u64 *xp_gran_l = nullptr;      // long-context variant (kernels_xlong.hip.h): scores and partial outputs of the key-range helpers
    bgk::xp_u64 *fp_gran = nullptr; bgk::FpLayer *fp_layers = nullptr; uint32_t *fp_ctl = nullptr;   // the float-weight persistent launch (kernels_fpipe.hip.h): hand-off granules, layer table, {tag, error word}
    int64_t fp_launches = 0;               // single-token steps enqueued (or captured) as the float-weight persistent launch (biogpt_hip_fpipe_launches)
    int fp_force = -1;                     // while a step is being CAPTURED: 1 the float-weight persistent launch, 0 five launches per layer (the graph is filed under that choice); -1: decided live
    int fp_state = 0;                      // 0 not prepared yet, 1 usable, -1 unavailable (shape, device, memory) or abandoned after a failed launch
    bgk::xp_u64 *xc_gran = nullptr;        // column-per-XCD chunk launches (kernels_xcols.hip.h): [8 columns][n_layer][XP_G_LAYER] granules, allocated (zeroed) at the first such eval
    int xc_lds = 0;                        // 0 not tried, 1 the kernels' LDS attribute is set, -1 it could not be set (such evals keep the launch chain)
    int state_n_past = 0, state_chunk = 0; // what the last upload_state put into the device state
    int64_t xc_launches = 0;               // evals that went through the chunk launch (biogpt_hip_chunk_launches)
    int32_t gen_launch_tokens[16] = {0}; int gen_launches = 0;   // the last biogpt_hip_generate_greedy: tokens of each multi-token pipelined launch (biogpt_hip_generate_launches)
    int xc_batch = 0;                      // biogpt_hip_generate_greedy_batch with 2 .. 8 sequences holds the device's pipeline slot: its decode steps run as column-per-XCD launches (streams mode)
    uint32_t *xp_ctl = nullptr;
    bgk::xp_u64 *xp_samp = nullptr;        // arg-max partials handed from token t to token t + 1 inside a multi-token launch
    uint32_t *xp_err_host = nullptr;
    bool xp_proc_ref = false;              // this context counts in the process's hold on the device's lock file (guarded by g_xp_mu)
    bool xp_foreign = false;               // another process holds the device's lock file: no pipelined launches AND no replay of captured five-launch steps here
    bool xp_in_call = false;               // guarded by g_xp_mu: an API call of this context has taken the device's pipeline slot and has not returned yet
    bool xp_tripped = false;               // set by xpipe_check when the pipeline failed in the call that just synchronised: the API entry repeats the call once
    int xp_exp_n = 0;      // the exp table's non-zero negative slice the attention workgroups keep in LDS
    int xp_gelu_p = 0, xp_gelu_n = 0, xp_gelu_z = 0;   // the GELU table's slices every workgroup keeps in LDS (kernels_xpipe.hip.h)
    int xp_state = 0;                      // 0 not probed, 1 usable, -1 unusable on this device / model / after a failure
    uint8_t *topk_host = nullptr;          // pinned, device-visible [64 floats][64 ints][count]: biogpt_hip_eval_topk's kernel writes it directly
    int32_t *mbox_host = nullptr;          // pinned ring of 64 x {n_past, causal, token}: inputs of the graph-replayed single-token evals
    uint32_t *mbox_ctr = nullptr;          // device: replays consumed
    uint32_t *seq_dev = nullptr;           // device [4]: lineage words of a replayed single-token eval (kernels.hip.h, SEQ_*)
    uint32_t call_seq = 0;                 // host: sequence number of the last replayed eval (never 0)
    uint32_t dev_stamp_expect = 0; int32_t dev_stamp_tok = 0, dev_stamp_n_past = 0;   // the same for a row left on the device (biogpt_hip_eval_device + biogpt_hip_read_logits)
    uint32_t stamp_expect = 0;             // != 0: the pinned row of the eval in flight must carry this number (five-launch graph, form 1)
    int64_t stale_rows = 0, graph_evals = 0;   // rows found to be another call's and repaired / single-token evals replayed as five-launch graphs
    bool plain_inflight = false;           // (guarded by g_xp_mu) five-launch graphs of this context may be in flight: no OTHER context takes the pipeline slot meanwhile
    uint32_t mbox_sent = 0, mbox_synced = 0;
    int unsynced_from = -1;                // position of the first single-token eval enqueued since the stream was last synchronised (-1: none): what a tripped pipeline may have spoiled
    int lm_blocks = 0;
    // resident single-token evals (biogpt_hip_eval): a pipelined launch that is still on the device, fed through res_mbox
    int32_t *res_mbox = nullptr;           // pinned ring of 64 x 32 bytes, the first 8 = bgk::xp_post(seq, n_past, token, speculate-next)
    uint32_t *res_done = nullptr;          // pinned [256]: per lm_head workgroup, the sequence number of the last token whose logits rows it has written to logits_host
    bool res_live = false;                 // a resident launch is (or may still be) on the device
    uint32_t res_seq = 0;                  // sequence number of the last token / request posted
    int res_next = 0, res_left = 0;        // position the live launch expects next; tokens it will still take
    int res_nw = 0;                        // completion words to collect per token
    // speculative continuation (XpParams::spec_rec): passes with an odd sequence number write the *_alt row / partial buffers
    float *logits_alt = nullptr, *pmax_val_alt = nullptr;      // device
    int32_t *pmax_idx_alt = nullptr;
    float *logits_host_alt = nullptr;      // pinned
    unsigned long long *res_spec = nullptr;    // pinned: {sequence number << 32 | the device's arg-max of the token before}
    const float *row_cur = nullptr;        // the pinned row that holds the logits of the last biogpt_hip_eval / eval_inplace
    uint32_t res_acc = 0;                  // sequence number of the last pass the caller asked for (its parity says which buffers hold its results)
    bool spec_pending = false;             // the live launch starts the next position on its own account
    int spec_streak = 0, spec_need = 4;    // consecutive calls whose token was the device's arg-max; how many it takes to speculate (doubles with every miss)
    long spec_hits = 0, spec_misses = 0, spec_run = 0;
    double res_t_wait = 0.0, res_t_call = 0.0; long res_calls = 0;   // measurement only (BIOGPT_HIP_RES_DBG & 8)
    std::chrono::steady_clock::time_point res_t_last{};
    bool ready = false;  // weights present
};

static bool resident_stop(biogpt_hip_ctx *c);   // ends a resident single-token launch (defined with the eval entry points)
// An entry point that enqueues work of its own (generate_*, eval_all, eval_prompt, the bench calls) makes the row of an EARLIER replayed single-token eval history: its lineage
// watch must not fire on a later biogpt_hip_read_logits (it would re-evaluate the old token at the old position over this call's K / V rows).  ADVICE r5.
static inline void disarm_lineage(biogpt_hip_ctx *c) { c->dev_stamp_expect = 0; c->stamp_expect = 0; }

namespace {

bgk::DevMatrix dev_matrix(const biogpt_hip_ctx *c, const MatSlot &m) {
    bgk::DevMatrix d;
    d.qs = c->arena + m.qs;
    d.sc = c->arena + m.sc;
    d.qh = reinterpret_cast<const uint32_t *>(c->arena + m.qh);
    d.type = m.type;
    d.M = (int32_t)m.M;
    d.K = (int32_t)m.K;
    return d;
}
const float *dev_vec(const biogpt_hip_ctx *c, size_t off) { return reinterpret_cast<const float *>(c->arena + off); }

struct MvShape { int upr, lpr_log2, nit, rpw, nwaves, grid; };
// the context whose launches the calling thread is enqueueing (set at every entry point that launches): the launch
// helpers below read its cached options and per-device state -- one host thread drives one device at a time
thread_local biogpt_hip_ctx *t_ctx = nullptr;
const EngineOptions &opt() { return t_ctx->opt; }

MvShape mv_shape(int32_t type, int64_t M, int64_t K, int target_wgs, int N = 1) {
    MvShape s;
    const int elems = (type == T_F32) ? 4 : (type == T_F16) ? 8 : 32;
    s.upr = (int)(K / elems);
    const int lpr = std::min(64, pow2ceil(s.upr));
    s.lpr_log2 = ilog2(lpr);
    s.nit = (s.upr + lpr - 1) / lpr;
    const int rps = 64 / lpr;
    // waves per workgroup: as many as possible (up to 4) while keeping >= target_wgs workgroups
    int nw = opt().mv_waves;
    while (nw > 1 && (M + (int64_t)nw * rps - 1) / ((int64_t)nw * rps) < target_wgs) nw >>= 1;
    int steps = 1;
    const int max_wgs = opt().max_wgs;
    // one finisher lane per (row, column) of a wave: rows_per_wave * columns <= 64
    (void)N;
    while ( (M + (int64_t)nw * rps * steps - 1) / ((int64_t)nw * rps * steps) > max_wgs) steps++;
    s.nwaves = nw;
    s.rpw = rps * steps;
    s.grid = (int)((M + (int64_t)nw * s.rpw - 1) / ((int64_t)nw * s.rpw));
    return s;
}

template <int WT, int PRO, int EPI, int NC>
hipError_t launch_mv_kch(const bgk::MatvecParams &p, const MvShape &s, hipStream_t st) {
    // register chunks per lane the prologue needs: LN holds the whole column, plain only the wave's share
    const int njj = (p.W.K / 4 + 63) / 64;
    const int need = (PRO == bgk::PRO_LN || NC > 1) ? njj : (njj + s.nwaves - 1) / s.nwaves;  // chunks a wave holds per column
    const size_t sm = bgk::matvec_smem_bytes(WT, p.W.K, NC, s.upr, s.rpw, s.nwaves);
    const int gy = (p.N + NC - 1) / NC;
    if (need <= 4) {
        hipLaunchKernelGGL((bgk::matvec_kernel<WT, PRO, EPI, NC, 4>), dim3(s.grid, gy), dim3(s.nwaves * 64), sm, st, p);
    } else if (need <= 16) {
        hipLaunchKernelGGL((bgk::matvec_kernel<WT, PRO, EPI, NC, 16>), dim3(s.grid, gy), dim3(s.nwaves * 64), sm, st, p);
    } else {
        return hipErrorInvalidValue;  // K > 4096*nwaves: not a BioGPT shape
    }
    return hipGetLastError();
}

template <int WT, int PRO, int EPI>
hipError_t launch_mv_nc(const bgk::MatvecParams &p, const MvShape &s, hipStream_t st) {
    constexpr int NCW = bgk::TypeInfo<WT>::quant ? 8 : 4;
    // a column tile keeps whole activation columns in registers (<= 16 float4 chunks per lane, K <= 4096); wider
    // inputs (BioGPT-large fc2, K = 6400) go one column per workgroup row, where a wave only holds its share
    if (p.N == 1 || (p.W.K / 4 + 63) / 64 > 16) return launch_mv_kch<WT, PRO, EPI, 1>(p, s, st);
    return launch_mv_kch<WT, PRO, EPI, NCW>(p, s, st);
}

// ---- shape-specialised single-token path (kernels_fast.hip.h) --------------------------------------
template <int WT, int PRO, int EPI, int K>
hipError_t launch_fast_k(bgk::MatvecParams p, hipStream_t st) {
    constexpr int BPR = K / 32, LPR = BPR < 64 ? BPR : 64, RPS = 64 / LPR;
    const int M = p.W.M;
    int steps = (EPI == bgk::EPI_LOGITS) ? opt().lm_steps : opt().fast_steps;
    while (steps > 1 && (M + 4 * RPS * steps - 1) / (4 * RPS * steps) < 128) steps >>= 1;  // keep the chip covered
    p.rpw = RPS * steps;
    const int grid = (M + 4 * p.rpw - 1) / (4 * p.rpw);
    const size_t sm = bgk::matvec_fast_smem_bytes(K, p.rpw);
    if (steps > 1) hipLaunchKernelGGL((bgk::matvec_fast_kernel<WT, PRO, EPI, K, 4>), dim3(grid), dim3(256), sm, st, p);
    else hipLaunchKernelGGL((bgk::matvec_fast_kernel<WT, PRO, EPI, K, 1>), dim3(grid), dim3(256), sm, st, p);
    return hipGetLastError();
}

template <int WT, int PRO, int EPI>
bool try_launch_fast(const bgk::MatvecParams &p, hipStream_t st, hipError_t &err, int *grid_out) {
    if (p.N != 1 || opt().no_fast) return false;
    const int K = p.W.K;
    if (EPI == bgk::EPI_QKV && p.D != K) return false;
    if constexpr (EPI == bgk::EPI_LOGITS && PRO == bgk::PRO_LN) {
        // the single-token lm_head as one pass with all loads up front (kernels_lmhead.hip.h), when the partials are the 64-row blocks its consumers expect
        if (K == 1024 && opt().lm_stream && opt().lm_steps == 8 && p.W.M >= 64 * 128 && (p.dbg & 0xff & ~32) == 0) {
            constexpr int NB = 3;      // 64-row blocks per workgroup of 8 waves (2 x 8 and 3 / 4 x 16 waves measured: 8.9 / 6.8 / 8.4 us)
            const int blocks = (p.W.M + 63) / 64;
            hipLaunchKernelGGL((bgk::lm_stream_kernel<WT>), dim3((blocks + NB - 1) / NB), dim3(512), bgk::lm_stream_smem_bytes(bgk::TypeInfo<WT>::q81), st, p);
            err = hipGetLastError();
            if (grid_out) *grid_out = blocks;
            return true;
        }
    }
    if (K == 1024) {
        err = launch_fast_k<WT, PRO, EPI, 1024>(p, st);
    } else if (K == 4096) {
        if constexpr (PRO == bgk::PRO_PLAIN) err = launch_fast_k<WT, PRO, EPI, 4096>(p, st);
        else return false;
    } else {
        return false;
    }
    if (grid_out) {
        const int BPR = K / 32, LPR = BPR < 64 ? BPR : 64, RPS = 64 / LPR;
        int steps = (EPI == bgk::EPI_LOGITS) ? opt().lm_steps : opt().fast_steps;
        while (steps > 1 && (p.W.M + 4 * RPS * steps - 1) / (4 * RPS * steps) < 128) steps >>= 1;
        *grid_out = (p.W.M + 4 * RPS * steps - 1) / (4 * RPS * steps);
    }
    return true;
}

// fast-chain launches with producer-quantized activations (attention -> out_proj, fc1 -> fc2; for
// prefill chunks also lnq_kernel -> q/k/v and -> fc1), NC = 1 (decode) or 8 columns per workgroup
template <int WT, int PRO, int EPI, int K, int PF, int NC>
hipError_t launch_fast_explicit(bgk::MatvecParams p, int steps, hipStream_t st) {
    constexpr int BPR = K / 32, LPR = BPR < 64 ? BPR : 64, RPS = 64 / LPR;
    p.rpw = RPS * steps;
    const int grid = (p.W.M + 4 * p.rpw - 1) / (4 * p.rpw);
    const int gy = (p.N + NC - 1) / NC;
    hipLaunchKernelGGL((bgk::matvec_fast_kernel<WT, PRO, EPI, K, PF, NC>), dim3(grid, gy), dim3(256),
                       bgk::matvec_fast_smem_bytes(K, p.rpw, NC, EPI == bgk::EPI_GELU_Q8), st, p);
    return hipGetLastError();
}

enum ChainOp { CHAIN_QKV_Q8, CHAIN_OPROJ, CHAIN_FC1, CHAIN_FC1_Q8, CHAIN_FC2, CHAIN_LMHEAD_Q8 };
template <int WT, int NC>
hipError_t launch_chain_nc(ChainOp op, const bgk::MatvecParams &p, hipStream_t st) {
    switch (op) {
        case CHAIN_QKV_Q8: return launch_fast_explicit<WT, bgk::PRO_Q8IN, bgk::EPI_QKV, 1024, 1, NC>(p, 1, st);
        case CHAIN_OPROJ: return launch_fast_explicit<WT, bgk::PRO_Q8IN, bgk::EPI_RESID, 1024, 1, NC>(p, 1, st);
        case CHAIN_FC1:   // decode: LayerNorm fused; 32 rows / workgroup = one Q8 block of fc2's input
            if constexpr (NC == 1) return launch_fast_explicit<WT, bgk::PRO_LN, bgk::EPI_GELU_Q8, 1024, 4, 1>(p, 4, st);
            else return hipErrorInvalidValue;
        case CHAIN_FC1_Q8: return launch_fast_explicit<WT, bgk::PRO_Q8IN, bgk::EPI_GELU_Q8, 1024, 4, NC>(p, 4, st);
        case CHAIN_FC2: return launch_fast_explicit<WT, bgk::PRO_Q8IN, bgk::EPI_RESID, 4096, 1, NC>(p, 1, st);
        case CHAIN_LMHEAD_Q8:  // rows_per_wave * columns <= 64 finisher lanes: 4 row steps with 8 columns
            if constexpr (NC == 8) return launch_fast_explicit<WT, bgk::PRO_Q8IN, bgk::EPI_LOGITS, 1024, 4, 8>(p, 4, st);
            else return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}
// many columns (prompt passes, batched sequences): the same chain on the int8 matrix cores, reading the row-tiled
// weight image (kernels_mfma.hip.h; the 25 kernels live in their own translation unit, mfma_tu.hip)
extern "C" int bg_mfma_launch(int wt, int op, const void *params, size_t params_bytes, const void *img, size_t img_bytes, hipStream_t st);
extern "C" int bg_mfma_retile(int wt, const void *src, size_t src_bytes, uint8_t *dq, uint8_t *ds, hipStream_t st);
template <int WT>
hipError_t launch_chain_mfma(ChainOp op, const bgk::MatvecParams &p, const bgk::DevMatrix &img, hipStream_t st) {
    int site = -1;
    switch (op) {
        case CHAIN_QKV_Q8: site = 0; break;
        case CHAIN_OPROJ: site = 1; break;
        case CHAIN_FC1_Q8: site = 2; break;
        case CHAIN_FC2: site = 3; break;
        case CHAIN_LMHEAD_Q8: site = 4; break;
        default: return hipErrorInvalidValue;
    }
    return (hipError_t)bg_mfma_launch(WT, site, &p, sizeof(p), &img, sizeof(img), st);
}
template <int WT>
hipError_t launch_chain_typed(ChainOp op, const bgk::MatvecParams &p, hipStream_t st, const bgk::DevMatrix *img) {
    if constexpr (bgk::TypeInfo<WT>::quant) {
        if (img != nullptr && op != CHAIN_FC1) return launch_chain_mfma<WT>(op, p, *img, st);
        if (p.N == 1 && p.seq == nullptr && op != CHAIN_LMHEAD_Q8 && op != CHAIN_QKV_Q8 && op != CHAIN_FC1_Q8) return launch_chain_nc<WT, 1>(op, p, st);
        return launch_chain_nc<WT, 8>(op, p, st);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_chain(ChainOp op, const bgk::MatvecParams &p, hipStream_t st, const bgk::DevMatrix *img = nullptr) {
    switch (p.W.type) {
        case T_Q4_0: return launch_chain_typed<bgk::W_Q4_0>(op, p, st, img);
        case T_Q4_1: return launch_chain_typed<bgk::W_Q4_1>(op, p, st, img);
        case T_Q5_0: return launch_chain_typed<bgk::W_Q5_0>(op, p, st, img);
        case T_Q5_1: return launch_chain_typed<bgk::W_Q5_1>(op, p, st, img);
        case T_Q8_0: return launch_chain_typed<bgk::W_Q8_0>(op, p, st, img);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_lnq(const biogpt_hip_ctx *c, const float *x, int N, size_t ln_w, size_t ln_b, int q81, hipStream_t st);

// ---- float weights, one column, BioGPT-base shapes (kernels_fdecode.hip.h): the whole matrix requested at t = 0 --------------------
template <int WT, int PRO, int EPI>
bool try_launch_fdec(const bgk::MatvecParams &p, hipStream_t st, hipError_t &err) {
    if constexpr (EPI == bgk::EPI_LOGITS || EPI == bgk::EPI_GELU_Q8 || PRO == bgk::PRO_Q8IN) {
        return false;
    } else {
        if (p.N != 1 || opt().no_fast || opt().no_fdec || p.seq != nullptr || (p.dbg & 0xff) != 0) return false;
        const int K = p.W.K, M = p.W.M;
        if (EPI == bgk::EPI_QKV && (p.D != K || p.dk <= 0)) return false;
        if (K == 1024 && M == 3072 && PRO == bgk::PRO_LN && EPI == bgk::EPI_QKV) hipLaunchKernelGGL((bgk::fdec_kernel<WT, PRO, EPI, 1024, 3>), dim3(256), dim3(256), 0, st, p);
        else if (K == 1024 && M == 4096 && PRO == bgk::PRO_LN && EPI == bgk::EPI_GELU) hipLaunchKernelGGL((bgk::fdec_kernel<WT, PRO, EPI, 1024, 4>), dim3(256), dim3(256), 0, st, p);
        else if (K == 1024 && M == 1024 && PRO == bgk::PRO_PLAIN && EPI == bgk::EPI_RESID) hipLaunchKernelGGL((bgk::fdec_kernel<WT, PRO, EPI, 1024, 1>), dim3(256), dim3(256), 0, st, p);
        else if (K == 4096 && M == 1024 && PRO == bgk::PRO_PLAIN && EPI == bgk::EPI_RESID) {
            if constexpr (PRO == bgk::PRO_PLAIN) hipLaunchKernelGGL((bgk::fdec_kernel<WT, PRO, EPI, 4096, 1>), dim3(256), dim3(256), 0, st, p);
        } else return false;
        err = hipGetLastError();
        return true;
    }
}

template <int WT, int PRO, int EPI>
hipError_t launch_mv_typed(const bgk::MatvecParams &p, const MvShape &s, hipStream_t st, int *grid_out) {
    if (grid_out) *grid_out = s.grid;
    if constexpr (bgk::TypeInfo<WT>::quant) {
        hipError_t err = hipSuccess;
        if (try_launch_fast<WT, PRO, EPI>(p, st, err, grid_out)) return err;
    } else {
        hipError_t err = hipSuccess;
        if (try_launch_fdec<WT, PRO, EPI>(p, st, err)) return err;
    }
    return launch_mv_nc<WT, PRO, EPI>(p, s, st);
}

template <int PRO, int EPI>
hipError_t launch_mv(const bgk::MatvecParams &p, const MvShape &s, hipStream_t st, int *grid_out = nullptr) {
    switch (p.W.type) {
        case T_F32: return launch_mv_typed<bgk::W_F32, PRO, EPI>(p, s, st, grid_out);
        case T_F16: return launch_mv_typed<bgk::W_F16, PRO, EPI>(p, s, st, grid_out);
        case T_Q4_0: return launch_mv_typed<bgk::W_Q4_0, PRO, EPI>(p, s, st, grid_out);
        case T_Q4_1: return launch_mv_typed<bgk::W_Q4_1, PRO, EPI>(p, s, st, grid_out);
        case T_Q5_0: return launch_mv_typed<bgk::W_Q5_0, PRO, EPI>(p, s, st, grid_out);
        case T_Q5_1: return launch_mv_typed<bgk::W_Q5_1, PRO, EPI>(p, s, st, grid_out);
        case T_Q8_0: return launch_mv_typed<bgk::W_Q8_0, PRO, EPI>(p, s, st, grid_out);
        default: return hipErrorInvalidValue;
    }
}

bgk::MatvecParams mv_base(const biogpt_hip_ctx *c, const MatSlot &m, const MvShape &s) {
    bgk::MatvecParams p{};
    p.W = dev_matrix(c, m);
    p.upr = s.upr; p.lpr_log2 = s.lpr_log2; p.nit = s.nit; p.rpw = s.rpw;
    p.eps = 1e-5f;  // NORM_EPS biogpt.cpp:24
    p.D = c->hp.d_model;
    p.dk = c->hp.d_model / c->hp.n_head;
    p.dk_log2 = ilog2(p.dk);
    p.P = c->hp.n_positions;
    p.st = c->state;
    p.gelu_tab = reinterpret_cast<const uint16_t *>(c->arena + c->plan.gelu_tab);
    p.dbg = c->opt.dbg | (c->launch_parity << 8);
    p.tstamp = c->tstamp;
    p.inv_k = 1.0 / (double)m.K;
    p.k_pow2 = (m.K & (m.K - 1)) == 0;
    return p;
}

int target_wgs() { return opt().target_wgs; }

hipError_t launch_lnq(const biogpt_hip_ctx *c, const float *x, int N, size_t ln_w, size_t ln_b, int q81, hipStream_t st) {
    const double inv_k = 1.0 / 1024.0;
    if (q81) hipLaunchKernelGGL((bgk::lnq_kernel<1024, true>), dim3(N), dim3(256), 0, st, x, 1024, dev_vec(c, ln_w), dev_vec(c, ln_b), 1e-5f, inv_k, c->aq_q[2], c->aq_d[2], c->aq_s[2]);
    else hipLaunchKernelGGL((bgk::lnq_kernel<1024, false>), dim3(N), dim3(256), 0, st, x, 1024, dev_vec(c, ln_w), dev_vec(c, ln_b), 1e-5f, inv_k, c->aq_q[2], c->aq_d[2], c->aq_s[2]);
    return hipGetLastError();
}

// The fixed launch sequence for N tokens at the device-resident n_past (biogpt_graph's op order).
// lm_rows: 0 = last row only into c->logits (+ arg-max partials), else all N rows into logits_all.
// ---- row-tiled, expanded weight image for the MFMA kernels (kernels_mfma.hip.h) ---------------------------------
// Per block 32 int8 weights + the scale as f32 ({d, m} for Q4_1 / Q5_1): laid out matrix by matrix the first time a pass has enough columns
// (retile_kernel, from the SoA arena); 36 / 40 bytes per block -- 340 MB for BioGPT-base Q4_0, next to 288 GB of HBM.
bgk::DevMatrix tile_matrix(const biogpt_hip_ctx *c, const MatSlot &m) {
    bgk::DevMatrix d;
    d.qs = c->tile_img + m.xq;
    d.sc = c->tile_img + m.xs;
    d.qh = nullptr;
    d.type = m.type; d.M = (int32_t)m.M; d.K = (int32_t)m.K;
    return d;
}
bool ensure_tile_images(biogpt_hip_ctx *c) {
    if (c->tile_img || c->tile_img_failed) return true;
    size_t off = 0;
    auto place = [&](MatSlot &m) {
        if (!is_quantized(m.type)) return;
        const size_t nblk = (size_t)m.M * (size_t)(m.K / QK);
        const bool q81 = m.type == T_Q4_1 || m.type == T_Q5_1;
        m.xq = off; off = (off + nblk * 32 + 255) & ~(size_t)255;
        m.xs = off; off = (off + nblk * (q81 ? 8 : 4) + 255) & ~(size_t)255;
    };
    for (auto &L : c->plan.layers) { place(L.qkv); place(L.o); place(L.fc1); place(L.fc2); }
    place(c->plan.lm_head);
    if (hipMalloc(&c->tile_img, std::max<size_t>(off, 256)) != hipSuccess) {
        // no room for the image (several contexts / replicas on one device): the passes stay on the VALU chain, which needs none
        (void)hipGetLastError();
        c->tile_img = nullptr; c->tile_img_failed = true;
        if (c->opt.verbose) fprintf(stderr, "biogpt_hip: no memory for the %zu-byte row-tiled weight image; many-column passes stay on the 8-column kernels\n", off);
        return true;
    }
    bool retile_ok = true;
    auto one = [&](const MatSlot &m) {
        if (!is_quantized(m.type)) return;
        const bgk::DevMatrix src = dev_matrix(c, m);
        if (bg_mfma_retile(m.type, &src, sizeof(src), c->tile_img + m.xq, c->tile_img + m.xs, c->stream) != (int)hipSuccess) retile_ok = false;
    };
    for (const auto &L : c->plan.layers) { one(L.qkv); one(L.o); one(L.fc1); one(L.fc2); }
    one(c->plan.lm_head);
    if (!retile_ok || hipGetLastError() != hipSuccess) {
        // a partially built image must never be used (the next call would find tile_img non-null and run the matrix-core chain on it): drop it, the passes stay on the VALU chain
        (void)hipStreamSynchronize(c->stream); (void)hipGetLastError();
        (void)hipFree(c->tile_img);
        c->tile_img = nullptr; c->tile_img_failed = true;
        if (c->opt.verbose) fprintf(stderr, "biogpt_hip: building the row-tiled weight image failed; many-column passes stay on the 8-column kernels\n");
        return true;
    }
    return true;
}

#include "engine_xpipe.inc"

// ---- fused single-token decode step (kernels_decode.hip.h): 3 launches per layer + lm_head ---------------------
// tok_src 1: the token is in the device state (an eval call); 2: arg-max of the previous step's lm_head partials.
// advance: the lm_head kernel moves the device-side position on by one when the step is done.
bool fused_decode_ok(const biogpt_hip_ctx *c, int t_max) {
    const auto &hp = c->hp;
    return is_quantized(ftype_to_type(hp.ftype)) && hp.d_model == 1024 && hp.d_ff == 4096 && hp.n_head == 16 && t_max <= 1024 &&
           hp.n_positions >= 64 && !c->opt.no_fast && !c->opt.no_chain && !c->opt.no_fused_decode;
}

// only: -1 = the five kernels of the layer in order; 0..4 = just that kernel (biogpt_hip_bench_matvec)
template <int WT>
hipError_t launch_decode_layer(biogpt_hip_ctx *c, const bgk::DecQkvParams &a, const bgk::DecAttnParams &at, const bgk::DecOprojParams &op,
                               const bgk::DecFc1Params &f1, const bgk::DecFc2Params &f2, int only = -1) {
    hipStream_t st = c->stream;
    if (only < 0 || only == 0) {
        if (c->opt.qkv_waves == 8) hipLaunchKernelGGL((bgk::dec_qkv_kernel<WT, 8>), dim3(192), dim3(512), bgk::dec_qkv_smem_bytes(), st, a);
        else if (c->opt.qkv_waves == 4) hipLaunchKernelGGL((bgk::dec_qkv_kernel<WT, 4>), dim3(384), dim3(256), bgk::dec_qkv_smem_bytes(), st, a);
        else hipLaunchKernelGGL((bgk::dec_qkv_kernel<WT, 16>), dim3(96), dim3(1024), bgk::dec_qkv_smem_bytes(), st, a);
    }
    if ((only < 0 || only == 1) && at.t_cap > 256) {
        // beyond 256 keys one workgroup per head would pull up to 512 KB of K / V through one compute unit: the keys of a head
        // are spread over H x T/64 workgroups in three dependent launches (attn_split_*_kernel, kernels_fast.hip.h)
        bgk::AttnParams sa{};
        sa.q = at.q; sa.kcache = at.kcache; sa.vcache = at.vcache; sa.out = c->att; sa.st = at.st; sa.exp_tab = at.exp_tab;
        sa.N = 1; sa.D = c->hp.d_model; sa.dk = 64; sa.P = at.P; sa.t_cap = at.t_cap;
        sa.oq_q = at.oq_q; sa.oq_d = at.oq_d; sa.oq_s = at.oq_s; sa.q81 = at.q81;
        sa.sp_scores = c->sp_scores; sa.sp_max = c->sp_max; sa.sp_pv = c->sp_pv;
        sa.n_split = (sa.t_cap + bgk::SPLIT_KEYS - 1) / bgk::SPLIT_KEYS;
        if (sa.n_split > bgk::SPLIT_MAX) BG_FAIL(hipErrorInvalidValue, "internal: %d key ranges exceed the %d the split attention kernels hold", sa.n_split, bgk::SPLIT_MAX);
        hipLaunchKernelGGL(bgk::attn_split_scores_kernel, dim3(16, sa.n_split), dim3(256), 0, st, sa);
        hipLaunchKernelGGL(bgk::attn_split_pv_kernel, dim3(16, sa.n_split), dim3(256), 0, st, sa);
        hipLaunchKernelGGL(bgk::attn_split_combine_kernel, dim3(16), dim3(64), 0, st, sa);
    } else if (only < 0 || only == 1) {
        if (c->opt.attn_waves == 8 && at.t_cap <= 64) hipLaunchKernelGGL((bgk::dec_attn_kernel<8, 8>), dim3(16), dim3(512), 0, st, at);
        else if (c->opt.attn_waves == 8 && at.t_cap <= 128) hipLaunchKernelGGL((bgk::dec_attn_kernel<4, 8>), dim3(16), dim3(512), 0, st, at);
        else if (at.t_cap <= 64) hipLaunchKernelGGL((bgk::dec_attn_kernel<16, 16>), dim3(16), dim3(1024), 0, st, at);
        else if (at.t_cap <= 128) hipLaunchKernelGGL((bgk::dec_attn_kernel<8, 16>), dim3(16), dim3(1024), 0, st, at);
        else hipLaunchKernelGGL((bgk::dec_attn_kernel<4, 16>), dim3(16), dim3(1024), 0, st, at);
    }
    if (only < 0 || only == 2) {
        if (c->opt.oproj_waves == 4) hipLaunchKernelGGL((bgk::dec_oproj_kernel<WT, 4>), dim3(128), dim3(256), bgk::dec_oproj_smem_bytes(4), st, op);
        else if (c->opt.oproj_waves == 8) hipLaunchKernelGGL((bgk::dec_oproj_kernel<WT, 8>), dim3(64), dim3(512), bgk::dec_oproj_smem_bytes(8), st, op);
        else hipLaunchKernelGGL((bgk::dec_oproj_kernel<WT, 16>), dim3(32), dim3(1024), bgk::dec_oproj_smem_bytes(16), st, op);
    }
    if (only < 0 || only == 3) {
        if (c->opt.fc1_blocks == 2) hipLaunchKernelGGL((bgk::dec_fc1_kernel<WT, 2, 16>), dim3(64), dim3(1024), bgk::dec_fc1_smem_bytes<2>(), st, f1);
        else if (c->opt.fc1_waves == 8) hipLaunchKernelGGL((bgk::dec_fc1_kernel<WT, 1, 8>), dim3(128), dim3(512), bgk::dec_fc1_smem_bytes<1>(), st, f1);
        else if (c->opt.fc1_waves == 4) hipLaunchKernelGGL((bgk::dec_fc1_kernel<WT, 1, 4>), dim3(128), dim3(256), bgk::dec_fc1_smem_bytes<1>(), st, f1);
        else hipLaunchKernelGGL((bgk::dec_fc1_kernel<WT, 1, 16>), dim3(128), dim3(1024), bgk::dec_fc1_smem_bytes<1>(), st, f1);
    }
    if (only < 0 || only == 4) {
        if (c->opt.fc2_waves == 4) hipLaunchKernelGGL((bgk::dec_fc2_kernel<WT, 4>), dim3(256), dim3(256), bgk::dec_fc2_smem_bytes(4), st, f2);
        else if (c->opt.fc2_waves == 8) hipLaunchKernelGGL((bgk::dec_fc2_kernel<WT, 8>), dim3(128), dim3(512), bgk::dec_fc2_smem_bytes(8), st, f2);
        else hipLaunchKernelGGL((bgk::dec_fc2_kernel<WT, 16>), dim3(64), dim3(1024), bgk::dec_fc2_smem_bytes(16), st, f2);
    }
    return hipGetLastError();
}

// grid of the single-token lm_head launch (launch_fast_k, K = 1024: 2 rows per wave step, 4 waves)
int fast_lm_grid(const biogpt_hip_ctx *c) {
    const int M = c->hp.n_vocab;
    int steps = c->opt.lm_steps;
    while (steps > 1 && (M + 8 * steps - 1) / (8 * steps) < 128) steps >>= 1;
    return (M + 8 * steps - 1) / (8 * steps);
}

// l0 / l1 / only: biogpt_hip_bench_matvec launches one kernel of one layer; the decode step is all layers + lm_head
// host_row (optional): pinned host buffer that also receives the logits row; *host_row_done tells whether the launch wrote it itself
// pl: -1 = go through the XCD pipeline if this context can take the device's slot now; 0 / 1 = the caller decided (graph capture: the
//     graph is replayed only in the matching state)
struct ResidentArgs { int32_t tok0, n_past0; uint32_t seq0; int32_t spec0; };
bool enqueue_decode_fused(biogpt_hip_ctx *c, int t_max, int tok_src, int advance, int l0 = 0, int l1 = -1, int only = -1, int n_tok = 1, float *host_row = nullptr,
                          bool *host_row_done = nullptr, int pl = -1, const ResidentArgs *ra = nullptr) {
    t_ctx = c;
    (void)hipGetLastError();
    const auto &hp = c->hp;
    const int D = hp.d_model, V = hp.n_vocab, P = hp.n_positions;
    const int lm_parts = fast_lm_grid(c);   // partials the previous step's lm_head left (same launch shape every step)
    if (lm_parts > c->pmax_cap) BG_FAIL(false, "internal: arg-max partial buffer too small (%d > %d)", lm_parts, c->pmax_cap);
    const int32_t wt = ftype_to_type(hp.ftype);
    const int q81 = (wt == T_Q4_1 || wt == T_Q5_1) ? 1 : 0;
    unsigned long long *const ts = (c->opt.dbg & 96) ? c->tstamp : nullptr;
    unsigned long long *const wall = (c->opt.dbg & 64) ? c->tstamp + 128 : nullptr;
    if (l1 < 0) l1 = hp.n_layer;
    const bool pipelined = only < 0 && l0 == 0 && l1 == hp.n_layer && tok_src != 0 && (pl < 0 ? xpipe_usable(c, t_max) : (pl == 1 && xpipe_bucket_ok(c, t_max)));
    bool lm_in_kernel = false;
    if (n_tok > 1 && !pipelined) BG_FAIL(false, "internal: multi-token launches exist on the XCD pipeline only");
    if (only == -2 && !pipelined) BG_FAIL(false, "the XCD-pipelined decode step is not available for this context");
    if (pipelined) {
        bgk::XpParams xp{};
        xp.layers = c->xp_layers; xp.n_layer = hp.n_layer; xp.gran = c->xp_gran; xp.gran_l = c->xp_gran_l; xp.ctl = c->xp_ctl; xp.err_host = c->xp_err_host;
        xp.st = c->state;
        xp.tok_emb = dev_matrix(c, c->plan.embed_tokens); xp.pos_emb = dev_matrix(c, c->plan.embed_pos);
        xp.embed_scale = sqrtf((float)D);
        xp.tok_src = tok_src; xp.as_res = c->opt.xpipe_as_res;
        xp.pmax_val = c->pmax_val; xp.pmax_idx = c->pmax_idx; xp.nparts = lm_parts;
        xp.n_positions = P; xp.n_vocab = V;
        xp.eps = 1e-5f; xp.q_scale = 1.0f / sqrtf(64.0f);
        xp.P = P; xp.t_cap = std::min(P, (t_max + 63) & ~63); xp.dual = (c->opt.xpipe_dual && c->xp_gran_l != nullptr) ? 1 : 0;
        xp.exp_tab = reinterpret_cast<const uint16_t *>(c->arena + c->plan.exp_tab);
        xp.gelu_tab = reinterpret_cast<const uint16_t *>(c->arena + c->plan.gelu_tab);
        xp.gelu_p = c->xp_gelu_p; xp.gelu_n = c->xp_gelu_n; xp.gelu_z = c->xp_gelu_z; xp.exp_n = c->xp_exp_n;
        xp.x_final = c->x;
        {   // lm_head inside the launch: its 64-row blocks (= the stand-alone launch's workgroups) three per workgroup of 7 XCDs
            const MatSlot &m = c->plan.lm_head;
            const bool fold = xpipe_lm_folds(c);
            xp.lm = fold ? 1 : 0;
            xp.lm_blocks = lm_parts; xp.adv = fold ? advance : 0; xp.n_tok = n_tok; xp.samp = c->xp_samp;   // no lm_head in here: the lm_head launch moves the position
            xp.Wlm = dev_matrix(c, m);
            xp.lm_ln_w = dev_vec(c, c->plan.ln_w); xp.lm_ln_b = dev_vec(c, c->plan.ln_b);
            xp.logits = c->logits; xp.logits_host = fold ? host_row : nullptr; xp.pmax_out_val = c->pmax_val; xp.pmax_out_idx = c->pmax_idx;
            lm_in_kernel = fold;
            if (host_row_done) *host_row_done = fold && host_row != nullptr;
            if (ra) {     // resident launch (biogpt_hip_eval): token 0 and its position travel in the parameter block, the following ones through the mailbox
                if (!(fold && host_row && advance == 0 && tok_src == 1 && xp.t_cap <= 1024 && (xp.t_cap <= 256 || xp.gran_l != nullptr))) BG_FAIL(false, "internal: a resident launch needs the lm_head inside the pipelined launch");
                xp.resident = 1; xp.mbox = c->res_mbox; xp.mbox_seq0 = ra->seq0; xp.done_host = c->res_done;
                xp.idle_ticks = (uint32_t)std::min(10000000, std::max(1, c->opt.resident_us)) * 100u;      // 100 MHz ticks; clamped to 10 s (the product must fit 32 bits)
                xp.res_tok0 = ra->tok0; xp.res_n_past0 = ra->n_past0; xp.res_dbg = c->opt.res_dbg;
                xp.res_spec0 = ra->spec0; xp.spec_rec = c->res_spec;
                xp.logits_alt = c->logits_alt; xp.logits_host_alt = c->logits_host_alt; xp.pmax_alt_val = c->pmax_val_alt; xp.pmax_alt_idx = c->pmax_idx_alt;
                if (!xp.spec_rec || !xp.logits_alt || !xp.logits_host_alt || !xp.pmax_alt_val || !xp.pmax_alt_idx) BG_FAIL(false, "internal: a resident launch without its alternate buffers");
            } else if (n_tok > 1 && !(fold && advance == 1 && tok_src == 2)) BG_FAIL(false, "internal: a multi-token launch needs the lm_head inside the pipeline");
        }
        xp.wall = ((c->opt.dbg & 128) || (ra && (c->opt.res_dbg & 32))) ? c->tstamp : nullptr;
        const hipError_t e = (hipError_t)bg_xpipe_launch(wt, xp.t_cap, bgk::xpipe_smem_bytes(xp.gelu_p + xp.gelu_n), c->stream, &xp, sizeof(xp));
        HIP_TRY(false, e);
    }
    for (int l = l0; l < l1 && !pipelined; l++) {
        const LayerSlots &L = c->plan.layers[(size_t)l];
        bgk::DecQkvParams a{};
        a.x = c->x; a.x_out = c->x;
        a.tok_emb = dev_matrix(c, c->plan.embed_tokens); a.pos_emb = dev_matrix(c, c->plan.embed_pos);
        a.embed_scale = sqrtf((float)D);
        a.tok_src = (l == 0) ? tok_src : 0;
        a.pmax_val = c->pmax_val; a.pmax_idx = c->pmax_idx; a.nparts = lm_parts;
        a.st = c->state; a.n_positions = P; a.n_vocab = V;
        a.ln_w = dev_vec(c, L.ln0_w); a.ln_b = dev_vec(c, L.ln0_b); a.eps = 1e-5f;
        a.Wqkv = dev_matrix(c, L.qkv); a.bqkv = dev_vec(c, L.qkv_b); a.q_scale = 1.0f / sqrtf(64.0f);
        a.q_out = c->q;
        a.kcache = c->memory_k + (size_t)l * P * D; a.vcache = c->memory_v + (size_t)l * P * D; a.P = P;
        a.tstamp = ts; a.wall = wall; a.wall_slot = 5 * l;
        bgk::DecAttnParams at{};
        at.q = c->q; at.kcache = a.kcache; at.vcache = a.vcache; at.st = c->state; at.P = P;
        at.t_cap = std::min(P, (t_max + 63) & ~63);
        at.exp_tab = reinterpret_cast<const uint16_t *>(c->arena + c->plan.exp_tab);
        at.oq_q = c->aq_q[0]; at.oq_d = c->aq_d[0]; at.oq_s = c->aq_s[0]; at.att_out = nullptr; at.q81 = q81;
        at.tstamp = ts ? ts + 16 : nullptr; at.wall = wall; at.wall_slot = 5 * l + 1;
        bgk::DecOprojParams op{};
        op.Wo = dev_matrix(c, L.o); op.aq_q = c->aq_q[0]; op.aq_d = c->aq_d[0]; op.aq_s = c->aq_s[0];
        op.bias = dev_vec(c, L.o_b); op.resid = c->x; op.out = c->x1;
        op.tstamp = ts ? ts + 32 : nullptr; op.wall = wall; op.wall_slot = 5 * l + 2;
        bgk::DecFc1Params f1{};
        f1.x1 = c->x1;
        f1.ln_w = dev_vec(c, L.ln1_w); f1.ln_b = dev_vec(c, L.ln1_b); f1.eps = 1e-5f;
        f1.W1 = dev_matrix(c, L.fc1); f1.b1 = dev_vec(c, L.fc1_b);
        f1.gelu_tab = reinterpret_cast<const uint16_t *>(c->arena + c->plan.gelu_tab);
        f1.oq_q = c->aq_q[1]; f1.oq_d = c->aq_d[1]; f1.oq_s = c->aq_s[1]; f1.q81 = q81;
        f1.tstamp = ts ? ts + 48 : nullptr; f1.wall = wall; f1.wall_slot = 5 * l + 3;
        bgk::DecFc2Params f2{};
        f2.W2 = dev_matrix(c, L.fc2); f2.aq_q = c->aq_q[1]; f2.aq_d = c->aq_d[1]; f2.aq_s = c->aq_s[1];
        f2.bias = dev_vec(c, L.fc2_b); f2.resid = c->x1; f2.out = c->x;
        f2.tstamp = ts ? ts + 64 : nullptr; f2.wall = wall; f2.wall_slot = 5 * l + 4;
        f2.seq = (l == hp.n_layer - 1) ? c->seq_dev : nullptr;
        hipError_t e = hipErrorInvalidValue;
        switch (wt) {
            case T_Q4_0: e = launch_decode_layer<bgk::W_Q4_0>(c, a, at, op, f1, f2, only); break;
            case T_Q4_1: e = launch_decode_layer<bgk::W_Q4_1>(c, a, at, op, f1, f2, only); break;
            case T_Q5_0: e = launch_decode_layer<bgk::W_Q5_0>(c, a, at, op, f1, f2, only); break;
            case T_Q5_1: e = launch_decode_layer<bgk::W_Q5_1>(c, a, at, op, f1, f2, only); break;
            case T_Q8_0: e = launch_decode_layer<bgk::W_Q8_0>(c, a, at, op, f1, f2, only); break;
            default: break;
        }
        HIP_TRY(false, e);
    }
    if (lm_in_kernel) { c->lm_blocks = lm_parts; return true; }   // the pipelined launch wrote the logits, the partials and moved the position
    if (only >= 0 || only == -2 || l1 < hp.n_layer) return true;   // one kernel (bench) or a leading segment of the step: no lm_head
    {  // final LayerNorm + lm_head (last row only, F8) + per-workgroup arg-max partials; block 0 advances the position
        const MatSlot &m = c->plan.lm_head;
        const MvShape s = mv_shape(m.type, m.M, m.K, target_wgs(), 1);
        bgk::MatvecParams p = mv_base(c, m, s);
        p.ln_w = dev_vec(c, c->plan.ln_w); p.ln_b = dev_vec(c, c->plan.ln_b);
        p.ldx = D; p.ldo = V; p.x = c->x; p.N = 1; p.out = c->logits;
        p.pmax_val = c->pmax_val; p.pmax_idx = c->pmax_idx;
        p.st_adv = c->state; p.adv = advance;
        p.lineage = c->seq_dev;
        int lm_grid = 0;
        HIP_TRY(false, (launch_mv<bgk::PRO_LN, bgk::EPI_LOGITS>(p, s, c->stream, &lm_grid)));
        if (lm_grid != lm_parts) BG_FAIL(false, "internal: lm_head grid %d != expected %d", lm_grid, lm_parts);
        c->lm_blocks = lm_grid;
    }
    return true;
}

// ---- single-token steps of FLOAT weight files as ONE persistent launch for all layers (kernels_fpipe.hip.h, fpipe_tu.hip) ----
extern "C" int bg_fpipe_launch(int wt, hipStream_t st, const void *params, size_t params_bytes);
extern "C" int bg_fpipe_set_lds(void);
bool fpipe_prepare(biogpt_hip_ctx *c) {
    if (c->fp_state != 0) return c->fp_state == 1;
    c->fp_state = -1;
    const auto &hp = c->hp;
    if (!c->opt.fpipe || c->opt.no_fast || hp.d_model != 1024 || hp.d_ff != 4096 || hp.n_head != 16 || hp.n_layer < 1 || c->device < 0 || c->device >= 64) return false;
    int32_t wt = -1;
    for (const auto &L : c->plan.layers) {
        for (const MatSlot *m : {&L.qkv, &L.o, &L.fc1, &L.fc2}) {
            if (m->type != T_F32 && m->type != T_F16) return false;
            if (wt >= 0 && m->type != wt) return false;
            wt = m->type;
        }
        if (L.qkv.M != 3072 || L.qkv.K != 1024 || L.o.M != 1024 || L.o.K != 1024 || L.fc1.M != 4096 || L.fc1.K != 1024 || L.fc2.M != 1024 || L.fc2.K != 4096) return false;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); c->fp_state = 0; return false; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess || prop.multiProcessorCount != 256 || (size_t)prop.sharedMemPerBlockOptin < bgk::fpipe_smem_bytes()) { (void)hipGetLastError(); return false; }
    if (!xpipe_process_lock(c)) return false;      // a persistent launch: one process per device drives them (engine_xpipe.inc)
    const size_t P = (size_t)hp.n_positions, D = (size_t)hp.d_model;
    std::vector<bgk::FpLayer> tab((size_t)hp.n_layer);
    for (int l = 0; l < hp.n_layer; l++) {
        const LayerSlots &L = c->plan.layers[(size_t)l];
        bgk::FpLayer &y = tab[(size_t)l];
        y.ln0_w = dev_vec(c, L.ln0_w); y.ln0_b = dev_vec(c, L.ln0_b); y.ln1_w = dev_vec(c, L.ln1_w); y.ln1_b = dev_vec(c, L.ln1_b);
        y.bqkv = dev_vec(c, L.qkv_b); y.bo = dev_vec(c, L.o_b); y.b1 = dev_vec(c, L.fc1_b); y.b2 = dev_vec(c, L.fc2_b);
        y.Wqkv = dev_matrix(c, L.qkv).qs; y.Wo = dev_matrix(c, L.o).qs; y.W1 = dev_matrix(c, L.fc1).qs; y.W2 = dev_matrix(c, L.fc2).qs;
        y.kcache = c->memory_k + (size_t)l * P * D; y.vcache = c->memory_v + (size_t)l * P * D;
    }
    const size_t gbytes = (size_t)(1024 + 3072 + 1024 + 1024 + 4096) * 8;
    const uint32_t ctl0[2] = {1u, 0u};
    bool ok = hipMalloc(&c->fp_layers, tab.size() * sizeof(bgk::FpLayer)) == hipSuccess && hipMalloc(&c->fp_gran, gbytes) == hipSuccess && hipMalloc(&c->fp_ctl, 64 + 3 * 1024 * 8) == hipSuccess &&
              hipMemcpy(c->fp_layers, tab.data(), tab.size() * sizeof(bgk::FpLayer), hipMemcpyHostToDevice) == hipSuccess && hipMemset(c->fp_gran, 0, gbytes) == hipSuccess &&
              hipMemset(c->fp_ctl, 0, 64 + 3 * 1024 * 8) == hipSuccess && hipMemcpy(c->fp_ctl, ctl0, 8, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && !c->xp_err_host) { ok = hipHostMalloc(reinterpret_cast<void **>(&c->xp_err_host), 64, hipHostMallocDefault) == hipSuccess; if (ok) *c->xp_err_host = 0u; }
    ok = ok && bg_fpipe_set_lds() == (int)hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (void *q : {(void *)c->fp_layers, (void *)c->fp_gran, (void *)c->fp_ctl}) if (q) (void)hipFree(q);
        c->fp_layers = nullptr; c->fp_gran = nullptr; c->fp_ctl = nullptr;
        return false;
    }
    c->fp_state = 1;
    return true;
}
bool fpipe_usable(biogpt_hip_ctx *c, int t_max) {
    if (t_max > bgk::FP_TMAX || !fpipe_prepare(c)) return false;
    return pipeline_slot_take(c);
}
// which form of a single-token step may be captured / replayed for this context bucket: 1 a persistent launch (the XCD pipeline; float files: kernels_fpipe.hip.h) -- the
// device's slot is then this context's until its next synchronisation --, 0 five launches per layer
int pipeline_pl(biogpt_hip_ctx *c, int t_max) {
    if (xpipe_usable(c, t_max)) return 1;
    return (c->fp_state >= 0 && !fused_decode_ok(c, t_max) && fpipe_usable(c, t_max)) ? 1 : 0;
}
// the token of the device state, embedded into c->x by the launch in front, through all layers; the final LayerNorm + lm_head launch follows
bool enqueue_fpipe(biogpt_hip_ctx *c) {
    const auto &hp = c->hp;
    bgk::FpParams fp{};
    fp.layers = c->fp_layers; fp.n_layer = hp.n_layer;
    fp.g_x = c->fp_gran; fp.g_qkv = fp.g_x + 1024; fp.g_att = fp.g_qkv + 3072; fp.g_x1 = fp.g_att + 1024; fp.g_h = fp.g_x1 + 1024;
    fp.ctl = c->fp_ctl; fp.err_host = c->xp_err_host;
    fp.st = c->state; fp.x_in = c->x; fp.x_out = c->x;
    fp.eps = 1e-5f; fp.q_scale = 1.0f / sqrtf(64.0f); fp.P = hp.n_positions;
    fp.exp_tab = reinterpret_cast<const uint16_t *>(c->arena + c->plan.exp_tab);
    fp.gelu_tab = reinterpret_cast<const uint16_t *>(c->arena + c->plan.gelu_tab);
    if (c->unsynced_from < 0) c->unsynced_from = c->state_n_past;
    fp.fault = (c->opt.fpipe_fault && c->fp_launches == 0) ? 1 : 0;
    fp.lead = c->opt.fpipe_lead >= 0 ? c->opt.fpipe_lead : (c->plan.layers[0].qkv.type == T_F32 ? 26 : 16);
    fp.stamps = c->opt.fpipe_stamps ? reinterpret_cast<unsigned long long *>(c->fp_ctl + 16) : nullptr;
    c->fp_launches++;
    HIP_TRY(false, (hipError_t)bg_fpipe_launch(c->plan.layers[0].qkv.type == T_F32 ? 0 : 1, c->stream, &fp, sizeof(fp)));
    return true;
}

// ---- biogpt_eval with 2 .. 8 tokens (the reference's prompt chunks, main.cpp:129-137) as ONE persistent launch: one column per XCD (kernels_xcols.hip.h) ----
extern "C" int bg_xcols_launch(int wt, int t_cap, size_t smem_bytes, hipStream_t st, const void *params, size_t params_bytes);
extern "C" int bg_xcols_set_lds(int wt, size_t smem_bytes);

// May this pass go through the chunk launch ?  It is one more launch of the context's pipeline (same control words and tag counter, all 256 compute units held):
// same conditions and the device's pipeline slot (taken here).  Not under the opt-in causal mask, not for columns that are several reference chunks (DevState::chunk).
// model / device / options fit, and the launch's buffers exist (allocated here: never inside a stream capture)
bool xcols_prepare(biogpt_hip_ctx *c, int N, int t_max) {
    const int32_t wt = ftype_to_type(c->hp.ftype);
    if (!c->opt.xcols || c->opt.causal || N < 2 || N > 8 || t_max > 512 || c->xc_lds < 0) return false;      // (257 .. 512 keys: the variant that requests the second half of a head's old rows inside the attention stage)
    if (!(wt == T_Q4_0 || wt == T_Q4_1 || wt == T_Q5_0 || wt == T_Q5_1 || wt == T_Q8_0)) return false;
    if (!fused_decode_ok(c, t_max) || c->xp_state != 1) return false;
    if (c->xc_gran && c->xc_lds == 1) return true;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
    if (!c->xc_gran) {
        const size_t bytes = (size_t)8 * c->hp.n_layer * bgk::XP_G_LAYER * 8;
        if (hipMalloc(&c->xc_gran, bytes) != hipSuccess || hipMemsetAsync(c->xc_gran, 0, bytes, c->stream) != hipSuccess) {
            (void)hipGetLastError();
            if (c->xc_gran) (void)hipFree(c->xc_gran);
            c->xc_gran = nullptr; c->xc_lds = -1;
            return false;
        }
    }
    if (c->xc_lds == 0) {
        const size_t sm = bgk::xpipe_smem_bytes(c->xp_gelu_p + c->xp_gelu_n);
        c->xc_lds = (sm <= 64 * 1024 || bg_xcols_set_lds(wt, sm) == (int)hipSuccess) ? 1 : -1;
        (void)hipGetLastError();
        if (c->xc_lds < 0) return false;
    }
    return true;
}
bool xcols_usable(biogpt_hip_ctx *c, int N, int t_max) {
    if (c->state_chunk != 0 || !xcols_prepare(c, N, t_max)) return false;
    return xpipe_usable(c, std::min(t_max, 256));      // the device's pipeline slot (a chunk launch needs none of the long-context launch's regions)
}

// the N columns of the device state (upload_state) through all layers in one launch, then the ordinary final LayerNorm + lm_head launch on the LAST column (F8)
bool enqueue_xcols(biogpt_hip_ctx *c, int N, int t_max, bool streams = false) {
    t_ctx = c;
    (void)hipGetLastError();
    const auto &hp = c->hp;
    const int D = hp.d_model, V = hp.n_vocab, P = hp.n_positions;
    const int32_t wt = ftype_to_type(hp.ftype);
    bgk::XcParams xc{};
    xc.layers = c->xp_layers; xc.n_layer = hp.n_layer; xc.gran = c->xc_gran; xc.ctl = c->xp_ctl; xc.err_host = c->xp_err_host;
    xc.st = c->state;
    xc.tok_emb = dev_matrix(c, c->plan.embed_tokens); xc.pos_emb = dev_matrix(c, c->plan.embed_pos);
    xc.embed_scale = sqrtf((float)D);
    xc.n_positions = P; xc.n_vocab = V;
    xc.eps = 1e-5f; xc.q_scale = 1.0f / sqrtf(64.0f);
    xc.P = P; xc.t_cap = std::min(P, (t_max + 63) & ~63);
    xc.exp_tab = reinterpret_cast<const uint16_t *>(c->arena + c->plan.exp_tab);
    xc.gelu_tab = reinterpret_cast<const uint16_t *>(c->arena + c->plan.gelu_tab);
    xc.gelu_p = c->xp_gelu_p; xc.gelu_n = c->xp_gelu_n; xc.gelu_z = c->xp_gelu_z;
    xc.n_cols = N; xc.x_out = c->x;
    if (streams) { xc.seq = c->seq; xc.kroot = c->bk; xc.vroot = c->bv; xc.seq_stride = (int64_t)hp.n_layer * P * D; }
    if ((c->opt.dbg & 128) && !c->tstamp) {      // profiling builds: stage stamps (tools/xcols_timeline.py)
        HIP_TRY(false, hipMalloc(&c->tstamp, (size_t)4 << 20));
        HIP_TRY(false, hipMemset(c->tstamp, 0, (size_t)4 << 20));
    }
    xc.wall = (c->opt.dbg & 128) ? c->tstamp : nullptr;
    if (c->unsynced_from < 0) c->unsynced_from = streams ? 0 : c->state_n_past;      // what a disturbed launch would spoil: this chunk's K / V rows (and every later eval's until the next synchronisation)
    c->xc_launches++;
    HIP_TRY(false, (hipError_t)bg_xcols_launch(wt, xc.t_cap, bgk::xpipe_smem_bytes(xc.gelu_p + xc.gelu_n), c->stream, &xc, sizeof(xc)));
    if (streams) return true;      // every sequence's row: the caller's 8-column lm_head chain on c->x
    {  // final LayerNorm + lm_head of the last column + per-workgroup arg-max partials
        const MatSlot &m = c->plan.lm_head;
        const MvShape s = mv_shape(m.type, m.M, m.K, target_wgs(), 1);
        bgk::MatvecParams p = mv_base(c, m, s);
        p.ln_w = dev_vec(c, c->plan.ln_w); p.ln_b = dev_vec(c, c->plan.ln_b);
        p.ldx = D; p.ldo = V;
        p.x = c->x + (size_t)(N - 1) * D; p.N = 1; p.out = c->logits;
        if (s.grid > c->pmax_cap) BG_FAIL(false, "internal: arg-max partial buffer too small (%d > %d)", s.grid, c->pmax_cap);
        p.pmax_val = c->pmax_val; p.pmax_idx = c->pmax_idx;
        int lm_grid = 0;
        HIP_TRY(false, (launch_mv<bgk::PRO_LN, bgk::EPI_LOGITS>(p, s, c->stream, &lm_grid)));
        c->lm_blocks = lm_grid;
    }
    return true;
}

// batch: one column per sequence (decode step).  cols != null: the columns are prompt tokens of several sequences
// (column states with seq_id / t_vis), no lm_head -- the caller gets the logits from the following decode step.
bool enqueue_forward(biogpt_hip_ctx *c, int N, bool all_rows, int t_max, bool batch = false, const bgk::SeqState *cols = nullptr) {
    t_ctx = c;
    (void)hipGetLastError();   // a failed call of some OTHER context / API leaves its code behind; the checks below are about these launches
    if (N < 1 || N > c->hp.n_positions) BG_FAIL(false, "internal: a pass of %d columns exceeds the %d-column activation scratch", N, c->hp.n_positions);
    if (N == 1 && !batch && !all_rows && fused_decode_ok(c, t_max)) return enqueue_decode_fused(c, t_max, 1, 0);
    if (N >= 2 && N <= 8 && !batch && !all_rows && xcols_usable(c, N, t_max)) return enqueue_xcols(c, N, t_max);
    const auto &hp = c->hp;
    const int D = hp.d_model, F = hp.d_ff, V = hp.n_vocab, H = hp.n_head, P = hp.n_positions;
    const int dk = D / H;
    hipStream_t st = c->stream;
    const int tw = target_wgs();
    // attention workgroup size: a thread owns up to ATTN_MAXK whole keys, so T <= 4 * threads
    int attn_threads = 256;
    while (attn_threads < 1024 && t_max > attn_threads) attn_threads <<= 1;  // ~1 key per thread when possible
    if (attn_threads % dk != 0 || t_max > bgk::ATTN_MAXK * attn_threads)
        BG_FAIL(false, "context of %d tokens / head size %d not supported by the attention kernel", t_max, dk);

    // single-token fast chain: BioGPT-base shapes, block-quantized weights -> producer-side Q8 hand-offs
    const int32_t wt = ftype_to_type(hp.ftype);
    const bool chain = is_quantized(wt) && D == 1024 && F == 4096 && dk == 64 && t_max <= 1024 &&
                       !opt().no_fast && !opt().no_chain;
    const bool pchain = chain && (N > 1 || batch);   // several columns: LayerNorm+Q8 once per site (lnq_kernel), 8 columns per workgroup
    // enough columns to fill 16-wide MFMA tiles: the chain runs on the int8 matrix cores from the row-tiled weight image
    // (measured cross-overs: decode steps of S sequences 48; prompt passes 64 columns)
    const bool mfma = pchain && N >= opt().mfma_min((batch && !cols) ? 48 : 64) && c->tile_img != nullptr;
    bgk::DevMatrix img;
    auto tile = [&](const MatSlot &m) -> const bgk::DevMatrix * { if (!mfma) return nullptr; img = tile_matrix(c, m); return &img; };
    if (batch && !chain) BG_FAIL(false, "batched decode needs the BioGPT-base fast chain (block-quantized weights, d_model 1024, d_ff 4096, head size 64)");
    const int64_t seq_stride = (int64_t)hp.n_layer * P * D;
    float *const kroot = batch ? c->bk : c->memory_k;
    float *const vroot = batch ? c->bv : c->memory_v;
    const int q81 = (wt == T_Q4_1 || wt == T_Q5_1) ? 1 : 0;

    // a decode step of 2 .. 8 sequences while the caller (biogpt_hip_generate_greedy_batch) holds the device's pipeline slot: embedding + all layers as ONE launch,
    // one sequence per XCD (kernels_xcols.hip.h, streams mode); the rows below on its output
    const bool xc_streams = batch && !cols && c->xc_batch != 0 && t_max <= 256 && xcols_prepare(c, N, t_max);
    if (xc_streams) { if (!enqueue_xcols(c, N, t_max, true)) return false; }
    else
    hipLaunchKernelGGL(bgk::embed_kernel, dim3((D + 255) / 256, N), dim3(256), 0, st,
                       dev_matrix(c, c->plan.embed_tokens), dev_matrix(c, c->plan.embed_pos), c->state,
                       sqrtf((float)D), c->x, D, batch ? (cols ? cols : c->seq) : nullptr);
    // a single token of a float-weight file: all layers as ONE persistent launch (kernels_fpipe.hip.h) on the embedding the launch above left in c->x
    const bool fp_one = N == 1 && !batch && !all_rows && !chain &&
                        (c->fp_force >= 0 ? (c->fp_force == 1 && c->fp_state == 1 && t_max <= bgk::FP_TMAX) : fpipe_usable(c, t_max));
    if (fp_one && !enqueue_fpipe(c)) return false;
    for (int l = 0; l < hp.n_layer && !xc_streams && !fp_one; l++) {
        const LayerSlots &L = c->plan.layers[(size_t)l];
        {  // LN0 + fused q/k/v projection + bias + Q scale + KV append
            const MvShape s = mv_shape(L.qkv.type, L.qkv.M, L.qkv.K, tw, N);
            bgk::MatvecParams p = mv_base(c, L.qkv, s);
            p.x = c->x; p.ldx = D; p.N = N;
            p.ln_w = dev_vec(c, L.ln0_w); p.ln_b = dev_vec(c, L.ln0_b);
            p.bias = dev_vec(c, L.qkv_b);
            p.q_out = c->q;
            p.kcache = kroot + (size_t)l * P * D;
            p.vcache = vroot + (size_t)l * P * D;
            if (batch) { p.seq = cols ? cols : c->seq; p.col_mode = cols ? 1 : 0; p.kv_seq_stride = seq_stride; }
            p.q_scale = 1.0f / sqrtf((float)dk);
            if (pchain) {
                HIP_TRY(false, launch_lnq(c, c->x, N, L.ln0_w, L.ln0_b, q81, st));
                p.aq_q = c->aq_q[2]; p.aq_d = c->aq_d[2]; p.aq_s = c->aq_s[2];
                HIP_TRY(false, launch_chain(CHAIN_QKV_Q8, p, st, tile(L.qkv)));
            } else {
                HIP_TRY(false, (launch_mv<bgk::PRO_LN, bgk::EPI_QKV>(p, s, st)));
            }
        }
        {  // attention
            bgk::AttnParams a{};
            a.q = c->q; a.kcache = kroot + (size_t)l * P * D; a.vcache = vroot + (size_t)l * P * D;
            if (batch) { a.seq = cols ? cols : c->seq; a.col_mode = cols ? 1 : 0; a.kv_seq_stride = seq_stride; }
            a.out = c->att; a.st = c->state;
            a.exp_tab = reinterpret_cast<const uint16_t *>(c->arena + c->plan.exp_tab);
            a.N = N; a.D = D; a.dk = dk; a.P = P;
            a.dbg = c->opt.dbg; a.tstamp = c->tstamp;
            a.q81 = q81;
            if (chain) { a.oq_q = c->aq_q[0]; a.oq_d = c->aq_d[0]; a.oq_s = c->aq_s[0]; }
            if (!batch && dk == 64 && t_max <= 1024 && N >= opt().attn_group_min && !opt().no_fast) {
                // a pass of many query columns: register-tiled kernel, 16 queries per workgroup share every K / V row they load
                // (BIOGPT_HIP_ATTN_TILE=0: the first grouped kernel, 8 queries per workgroup, kept as the A/B arm of the equivalence test)
                a.t_cap = std::min(P, t_max);
                // (the tile kernel addresses a thread's four consecutive key rows from ONE base clamped to P - 4: that is only their own rows when 4 | P and P >= 4;
                //  any other table size takes the grouped kernel, which clamps row by row)
                if (opt().attn_tile && (P & 3) == 0 && P >= 4) {
                    // up to 640 keys the K / V rows of the two MAC loops travel through a ring in LDS, two steps ahead (global_load_lds); beyond, the ring has no room
                    // beside the scores in a 2-workgroups-per-compute-unit footprint: loads at the top of each step
                    const bool dma = bgk::attn_tile_dma_ok(a.t_cap) && opt().attn_tile != 2;      // (BIOGPT_HIP_ATTN_TILE=2: the A/B arm without the ring)
                    const size_t smb = bgk::attn_tile_smem_bytes<16>(a.t_cap);
                    const void *fn = dma ? reinterpret_cast<const void *>(bgk::attn_tile_kernel<16, true>) : reinterpret_cast<const void *>(bgk::attn_tile_kernel<16, false>);
                    if (smb > 64 * 1024 && !c->lds_attr_done.count(fn)) {
                        HIP_TRY(false, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bgk::attn_tile_smem_bytes<16>(P)));
                        c->lds_attr_done.insert(fn);
                    }
                    if (dma) hipLaunchKernelGGL((bgk::attn_tile_kernel<16, true>), dim3(H, (N + 15) / 16), dim3(512), smb, st, a);
                    else hipLaunchKernelGGL((bgk::attn_tile_kernel<16, false>), dim3(H, (N + 15) / 16), dim3(512), smb, st, a);
                } else {
                    hipLaunchKernelGGL((bgk::attn_group_kernel<8>), dim3(H, (N + 7) / 8), dim3(512), bgk::attn_group_smem_bytes(a.t_cap), st, a);
                }
            } else if (dk == 64 && t_max <= 1024 && !opt().no_fast) {
                // loads are bounded by t_cap (= P when the table is not a multiple of 64; the workgroup stays whole
                // waves); 4 lanes per key, 16 prefetched V rows per lane
                a.t_cap = std::min(P, (t_max + 63) & ~63);
                const int split_min = opt().split_min;
                if (N == 1 && !batch && a.t_cap > split_min) {
                    // long context, one query: spread the head's keys over the chip (three dependent launches)
                    a.sp_scores = c->sp_scores; a.sp_max = c->sp_max; a.sp_pv = c->sp_pv;
                    a.n_split = (a.t_cap + bgk::SPLIT_KEYS - 1) / bgk::SPLIT_KEYS;
                    if (a.n_split > bgk::SPLIT_MAX) BG_FAIL(false, "internal: %d key ranges exceed the %d the split attention kernels hold", a.n_split, bgk::SPLIT_MAX);
                    hipLaunchKernelGGL(bgk::attn_split_scores_kernel, dim3(H, a.n_split), dim3(256), 0, st, a);
                    hipLaunchKernelGGL(bgk::attn_split_pv_kernel, dim3(H, a.n_split), dim3(256), 0, st, a);
                    hipLaunchKernelGGL(bgk::attn_split_combine_kernel, dim3(H), dim3(64), 0, st, a);
                } else if (batch && N >= opt().attn_slim_min) {
                    // many (sequence, head) workgroups: throughput over latency -- one lane quad per 4 keys (4 key passes), a
                    // quarter of the threads, four times as many workgroups resident per compute unit
                    const int t64 = (a.t_cap + 63) & ~63;
                    hipLaunchKernelGGL((bgk::attn_fast_kernel<4, false>), dim3(H, N), dim3(std::max(256, t64)), 0, st, a);
                } else if (a.t_cap <= 256) {
                    hipLaunchKernelGGL((bgk::attn_fast_kernel<1, true>), dim3(H, N), dim3(4 * ((a.t_cap + 63) & ~63)), 0, st, a);
                } else if (a.t_cap <= 512) {
                    hipLaunchKernelGGL((bgk::attn_fast_kernel<2, false>), dim3(H, N), dim3(1024), 0, st, a);
                } else {
                    hipLaunchKernelGGL((bgk::attn_fast_kernel<4, false>), dim3(H, N), dim3(1024), 0, st, a);
                }
            } else {
                hipLaunchKernelGGL(bgk::attn_kernel, dim3(H, N), dim3(attn_threads), bgk::attn_smem_bytes(P, dk, attn_threads), st, a);
            }
        }
        {  // out_proj + bias + residual
            const MvShape s = mv_shape(L.o.type, L.o.M, L.o.K, tw, N);
            bgk::MatvecParams p = mv_base(c, L.o, s);
            p.x = c->att; p.ldx = D; p.N = N;
            p.bias = dev_vec(c, L.o_b);
            p.resid = c->x; p.ldr = D; p.out = c->x1; p.ldo = D;
            if (chain) {
                p.aq_q = c->aq_q[0]; p.aq_d = c->aq_d[0]; p.aq_s = c->aq_s[0];
                HIP_TRY(false, launch_chain(CHAIN_OPROJ, p, st, tile(L.o)));
            } else {
                HIP_TRY(false, (launch_mv<bgk::PRO_PLAIN, bgk::EPI_RESID>(p, s, st)));
            }
        }
        {  // LN1 + fc1 + bias + GELU
            const MvShape s = mv_shape(L.fc1.type, L.fc1.M, L.fc1.K, tw, N);
            bgk::MatvecParams p = mv_base(c, L.fc1, s);
            p.x = c->x1; p.ldx = D; p.N = N;
            p.ln_w = dev_vec(c, L.ln1_w); p.ln_b = dev_vec(c, L.ln1_b);
            p.bias = dev_vec(c, L.fc1_b);
            p.out = c->h; p.ldo = F;
            if (pchain) {
                HIP_TRY(false, launch_lnq(c, c->x1, N, L.ln1_w, L.ln1_b, q81, st));
                p.aq_q = c->aq_q[2]; p.aq_d = c->aq_d[2]; p.aq_s = c->aq_s[2];
                p.oq_q = c->aq_q[1]; p.oq_d = c->aq_d[1]; p.oq_s = c->aq_s[1];
                HIP_TRY(false, launch_chain(CHAIN_FC1_Q8, p, st, tile(L.fc1)));
            } else if (chain) {
                p.oq_q = c->aq_q[1]; p.oq_d = c->aq_d[1]; p.oq_s = c->aq_s[1];
                HIP_TRY(false, launch_chain(CHAIN_FC1, p, st));
            } else {
                HIP_TRY(false, (launch_mv<bgk::PRO_LN, bgk::EPI_GELU>(p, s, st)));
            }
        }
        {  // fc2 + bias + residual
            const MvShape s = mv_shape(L.fc2.type, L.fc2.M, L.fc2.K, tw, N);
            bgk::MatvecParams p = mv_base(c, L.fc2, s);
            p.x = c->h; p.ldx = F; p.N = N;
            p.bias = dev_vec(c, L.fc2_b);
            p.resid = c->x1; p.ldr = D; p.out = c->x; p.ldo = D;
            if (chain) {
                p.aq_q = c->aq_q[1]; p.aq_d = c->aq_d[1]; p.aq_s = c->aq_s[1];
                HIP_TRY(false, launch_chain(CHAIN_FC2, p, st, tile(L.fc2)));
            } else {
                HIP_TRY(false, (launch_mv<bgk::PRO_PLAIN, bgk::EPI_RESID>(p, s, st)));
            }
        }
    }
    if (batch && cols) return true;   // prompt columns: only the KV rows matter
    if (batch) {  // every sequence needs its logits row: LayerNorm+Q8 once, then the 8-column mat-vec
        const MatSlot &m = c->plan.lm_head;
        const MvShape s = mv_shape(m.type, m.M, m.K, tw, N);
        bgk::MatvecParams p = mv_base(c, m, s);
        HIP_TRY(false, launch_lnq(c, c->x, N, c->plan.ln_w, c->plan.ln_b, q81, st));
        p.aq_q = c->aq_q[2]; p.aq_d = c->aq_d[2]; p.aq_s = c->aq_s[2];
        p.N = N; p.out = c->logits_all; p.ldo = V;
        HIP_TRY(false, launch_chain(CHAIN_LMHEAD_Q8, p, st, tile(c->plan.lm_head)));
        return true;
    }
    {  // final LayerNorm + lm_head; only the rows that are returned (F8)
        const MatSlot &m = c->plan.lm_head;
        const MvShape s = mv_shape(m.type, m.M, m.K, tw, all_rows ? N : 1);
        bgk::MatvecParams p = mv_base(c, m, s);
        p.ln_w = dev_vec(c, c->plan.ln_w); p.ln_b = dev_vec(c, c->plan.ln_b);
        p.ldx = D; p.ldo = V;
        if (all_rows) {
            p.x = c->x; p.N = N; p.out = c->logits_all;
        } else {
            p.x = c->x + (size_t)(N - 1) * D; p.N = 1; p.out = c->logits;
            if (s.grid > c->pmax_cap) BG_FAIL(false, "internal: arg-max partial buffer too small (%d > %d)", s.grid, c->pmax_cap);
            p.pmax_val = c->pmax_val; p.pmax_idx = c->pmax_idx;
        }
        int lm_grid = 0;
        HIP_TRY(false, (launch_mv<bgk::PRO_LN, bgk::EPI_LOGITS>(p, s, st, &lm_grid)));
        if (!all_rows) c->lm_blocks = lm_grid;
    }
    return true;
}

bool enqueue_argmax(biogpt_hip_ctx *c, int n_eval) {
    hipLaunchKernelGGL(bgk::argmax_kernel, dim3(1), dim3(256), 0, c->stream, c->pmax_val, c->pmax_idx, c->lm_blocks,
                       c->state, n_eval, c->hp.n_positions);
    HIP_TRY(false, hipGetLastError());
    return true;
}

constexpr int STATE_SLOTS = 64;

// Async upload of {n_past, tokens} through a ring of pinned slots (a slot is only reused after the
// stream has drained, so an in-flight copy never sees a half-written slot).
bool upload_state(biogpt_hip_ctx *c, const int32_t *tokens, int n, int n_past, int chunk = 0) {
    if (c->slot_idx == STATE_SLOTS) {
        HIP_TRY(false, hipStreamSynchronize(c->stream));
        c->slot_idx = 0;
    }
    uint8_t *slot = c->state_host + (size_t)c->slot_idx++ * c->slot_bytes;
    auto *hs = reinterpret_cast<bgk::DevState *>(slot);
    hs->n_past = n_past;
    hs->n_gen = 0;
    hs->causal = c->opt.causal;
    hs->chunk = chunk;
    c->state_n_past = n_past; c->state_chunk = chunk;
    std::memcpy(slot + sizeof(bgk::DevState), tokens, (size_t)n * 4);
    HIP_TRY(false, hipMemcpyAsync(c->state, slot, sizeof(bgk::DevState) + (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    return true;
}

bool check_eval_args(const biogpt_hip_ctx *c, const int32_t *tokens, int n, int n_past) {
    if (!c) BG_FAIL(false, "null context");
    if (!c->ready) BG_FAIL(false, "model has no tensors loaded (empty model): cannot evaluate");
    if (!tokens || n < 1) BG_FAIL(false, "no tokens to evaluate");
    if (n_past < 0 || n_past + n > c->hp.n_positions)
        BG_FAIL(false, "n_past (%d) + n_tokens (%d) exceeds n_positions (%d)", n_past, n, c->hp.n_positions);
    for (int i = 0; i < n; i++)
        if (tokens[i] < 0 || tokens[i] >= c->hp.n_vocab) BG_FAIL(false, "token id %d out of range [0, %d)", tokens[i], c->hp.n_vocab);
    return true;
}

bool alloc_runtime(biogpt_hip_ctx *c) {
    const auto &hp = c->hp;
    const size_t P = (size_t)hp.n_positions, D = (size_t)hp.d_model, F = (size_t)hp.d_ff, V = (size_t)hp.n_vocab;
    const size_t kv = (size_t)hp.n_layer * P * D * 4;
    HIP_TRY(false, hipMalloc(&c->memory_k, kv));
    HIP_TRY(false, hipMalloc(&c->memory_v, kv));
    HIP_TRY(false, hipMemset(c->memory_k, 0, kv));
    HIP_TRY(false, hipMemset(c->memory_v, 0, kv));
    HIP_TRY(false, hipMalloc(&c->x, P * D * 4));
    HIP_TRY(false, hipMalloc(&c->x1, P * D * 4));
    HIP_TRY(false, hipMalloc(&c->q, P * D * 4));
    HIP_TRY(false, hipMalloc(&c->att, P * D * 4));
    HIP_TRY(false, hipMalloc(&c->h, P * F * 4));
    HIP_TRY(false, hipMalloc(&c->logits, V * 4));
    for (int k = 0; k < 3; k++) {
        const size_t n = (k == 1 ? F : D) * P;
        HIP_TRY(false, hipMalloc(&c->aq_q[k], n));
        HIP_TRY(false, hipMalloc(&c->aq_d[k], n / 32 * 4 + 16));
        HIP_TRY(false, hipMalloc(&c->aq_s[k], n / 32 * 4 + 16));
    }
    c->pmax_cap = 4096;
    HIP_TRY(false, hipMalloc(&c->pmax_val, (size_t)c->pmax_cap * 4));
    HIP_TRY(false, hipMalloc(&c->sp_scores, (size_t)hp.n_head * hp.n_positions * 4));
    HIP_TRY(false, hipMalloc(&c->sp_max, (size_t)hp.n_head * bgk::SPLIT_MAX * 4));
    HIP_TRY(false, hipMalloc(&c->sp_pv, (size_t)hp.n_head * bgk::SPLIT_MAX * 64 * 8));
    HIP_TRY(false, hipMalloc(&c->pmax_idx, (size_t)c->pmax_cap * 4));
    HIP_TRY(false, hipMemset(c->pmax_val, 0, (size_t)c->pmax_cap * 4));
    HIP_TRY(false, hipMemset(c->pmax_idx, 0, (size_t)c->pmax_cap * 4));
    c->state_bytes = sizeof(bgk::DevState) + 2 * P * 4;
    HIP_TRY(false, hipMalloc(&c->state, c->state_bytes));
    HIP_TRY(false, hipMemset(c->state, 0, c->state_bytes));
    c->slot_bytes = (sizeof(bgk::DevState) + P * 4 + 63) & ~(size_t)63;
    HIP_TRY(false, hipHostMalloc(reinterpret_cast<void **>(&c->state_host), c->slot_bytes * STATE_SLOTS, hipHostMallocDefault));
    std::memset(c->state_host, 0, c->slot_bytes * STATE_SLOTS);
    HIP_TRY(false, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(false, hipEventCreate(&c->ev0));
    HIP_TRY(false, hipEventCreate(&c->ev1));
    return true;
}

bool select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) BG_FAIL(false, "no HIP device available (%s): this engine has no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= n) BG_FAIL(false, "HIP device %d out of range (found %d)", device, n);
    HIP_TRY(false, hipSetDevice(device));
    return true;
}

// read every tensor, validate against the expected directory (biogpt.cpp:394-417), repack, upload
bool upload_weights(biogpt_hip_ctx *c, const ModelFile &mf) {
    const auto &hp = c->hp;
    const int32_t wt = ftype_to_type(hp.ftype);
    const int64_t D = hp.d_model;
    const auto expected = expected_tensors(hp);
    // unknown tensors are an error (biogpt.cpp:394-397)
    for (const auto &t : mf.tensors) {
        bool known = false;
        for (const auto &e : expected) if (e.name == t.name) { known = true; break; }
        if (!known) BG_FAIL(false, "unknown tensor '%s' in model file", t.name.c_str());
    }
    if (mf.tensors.size() != expected.size())
        BG_FAIL(false, "ERROR not all tensors loaded from model file - expected %zu, got %zu", expected.size(), mf.tensors.size());

    std::vector<uint8_t> raw, qs, sc, qh;
    auto put_matrix = [&](const std::string &name, const MatSlot &slot, int64_t row0, int64_t rows, int64_t K, bool more_rows_ok = false) -> bool {
        const TensorEntry *t = mf.find(name);
        if (!t) BG_FAIL(false, "ERROR not all tensors loaded from model file - missing '%s'", name.c_str());
        if (t->ne0 != K || (more_rows_ok ? t->ne1 < rows : t->ne1 != rows))
            BG_FAIL(false, "tensor '%s' has wrong shape in model file: got [%lld, %lld], expected [%lld, %lld]", name.c_str(),
                    (long long)t->ne0, (long long)t->ne1, (long long)K, (long long)rows);
        if (t->type != wt) BG_FAIL(false, "tensor '%s' has wrong size in model file: type %s, expected %s", name.c_str(), type_name(t->type), type_name(wt));
        raw.resize(t->nbytes);
        if (!mf.read_payload(*t, raw.data())) return false;
        const size_t qb = unit_bytes_qs(wt, K) * (size_t)rows, sb = unit_bytes_sc(wt, K) * (size_t)rows, hb = unit_bytes_qh(wt, K) * (size_t)rows;
        qs.resize(qb); sc.resize(sb + 4); qh.resize(hb + 4);
        repack_rows(wt, raw.data(), rows, K, qs.data(), sc.data(), qh.data());
        HIP_TRY(false, hipMemcpy(c->arena + slot.qs + unit_bytes_qs(wt, K) * (size_t)row0, qs.data(), qb, hipMemcpyHostToDevice));
        if (sb) HIP_TRY(false, hipMemcpy(c->arena + slot.sc + unit_bytes_sc(wt, K) * (size_t)row0, sc.data(), sb, hipMemcpyHostToDevice));
        if (hb) HIP_TRY(false, hipMemcpy(c->arena + slot.qh + unit_bytes_qh(wt, K) * (size_t)row0, qh.data(), hb, hipMemcpyHostToDevice));
        return true;
    };
    auto put_vec = [&](const std::string &name, size_t off, int64_t elem0, int64_t n) -> bool {
        const TensorEntry *t = mf.find(name);
        if (!t) BG_FAIL(false, "ERROR not all tensors loaded from model file - missing '%s'", name.c_str());
        if (t->ne0 != n || t->ne1 != 1)
            BG_FAIL(false, "tensor '%s' has wrong shape in model file: got [%lld, %lld], expected [%lld, 1]", name.c_str(),
                    (long long)t->ne0, (long long)t->ne1, (long long)n);
        if (t->type != T_F32) BG_FAIL(false, "tensor '%s' has wrong size in model file: type %s, expected f32", name.c_str(), type_name(t->type));
        raw.resize(t->nbytes);
        if (!mf.read_payload(*t, raw.data())) return false;
        HIP_TRY(false, hipMemcpy(c->arena + off + (size_t)elem0 * 4, raw.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        return true;
    };

    const ArenaPlan &pl = c->plan;
    if (!put_matrix("biogpt.embed_tokens.weight", pl.embed_tokens, 0, hp.n_vocab, D)) return false;
    if (!put_matrix("biogpt.embed_positions.weight", pl.embed_pos, 0, c->pos_rows, D, true)) return false;   // first n_positions + 2 rows
    for (int l = 0; l < hp.n_layer; l++) {
        const LayerSlots &L = pl.layers[(size_t)l];
        const std::string p = "biogpt.layers." + std::to_string(l) + ".";
        const char *proj[3] = {"q_proj", "k_proj", "v_proj"};
        for (int k = 0; k < 3; k++) {
            if (!put_matrix(p + "self_attn." + proj[k] + ".weight", L.qkv, (int64_t)k * D, D, D)) return false;
            if (!put_vec(p + "self_attn." + proj[k] + ".bias", L.qkv_b, (int64_t)k * D, D)) return false;
        }
        if (!put_matrix(p + "self_attn.out_proj.weight", L.o, 0, D, D)) return false;
        if (!put_vec(p + "self_attn.out_proj.bias", L.o_b, 0, D)) return false;
        if (!put_vec(p + "self_attn_layer_norm.weight", L.ln0_w, 0, D)) return false;
        if (!put_vec(p + "self_attn_layer_norm.bias", L.ln0_b, 0, D)) return false;
        if (!put_vec(p + "final_layer_norm.weight", L.ln1_w, 0, D)) return false;
        if (!put_vec(p + "final_layer_norm.bias", L.ln1_b, 0, D)) return false;
        if (!put_matrix(p + "fc1.weight", L.fc1, 0, hp.d_ff, D)) return false;
        if (!put_vec(p + "fc1.bias", L.fc1_b, 0, hp.d_ff)) return false;
        if (!put_matrix(p + "fc2.weight", L.fc2, 0, D, hp.d_ff)) return false;
        if (!put_vec(p + "fc2.bias", L.fc2_b, 0, D)) return false;
    }
    if (!put_vec("biogpt.layer_norm.weight", pl.ln_w, 0, D)) return false;
    if (!put_vec("biogpt.layer_norm.bias", pl.ln_b, 0, D)) return false;
    if (!put_matrix("output_projection.weight", pl.lm_head, 0, hp.n_vocab, D)) return false;

    // fp16 tables, built the way ggml_init builds them [SURVEY A.5]
    std::vector<uint16_t> tg(65536), te(65536);
    for (uint32_t i = 0; i < 65536; i++) {
        const float f = f16_to_f32((uint16_t)i);
        tg[i] = f32_to_f16(gelu_tanh_f32(f));
        te[i] = f32_to_f16(expf(f));
    }
    HIP_TRY(false, hipMemcpy(c->arena + pl.gelu_tab, tg.data(), 65536 * 2, hipMemcpyHostToDevice));
    HIP_TRY(false, hipMemcpy(c->arena + pl.exp_tab, te.data(), 65536 * 2, hipMemcpyHostToDevice));
    return true;
}

void destroy(biogpt_hip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)resident_stop(c);
    if ((c->opt.res_dbg & 32) && c->tstamp && c->res_seq > 8) {      // device-side stamps: token seen by XCD 0's poller [s][0], lm_head workgroup 0's completion word [s][1]
        std::vector<unsigned long long> w(8192);
        if (hipMemcpy(w.data(), c->tstamp, w.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            double proc = 0.0, turn = 0.0; int np = 0, nt = 0;
            const uint32_t hi = c->res_seq < 4000u ? c->res_seq : 4000u;
            for (uint32_t q = 2; q + 1 < hi; q++) {
                const unsigned long long a = w[2 * q], b = w[2 * q + 1], a2 = w[2 * q + 2];
                if (a && b && b > a && b - a < 100000ull) { proc += (double)(b - a) * 0.01; np++; }
                if (b && a2 && a2 > b && a2 - b < 100000ull) { turn += (double)(a2 - b) * 0.01; nt++; }
            }
            fprintf(stderr, "resident launch, device clock: token seen -> completion word %.2f us (%d), completion word -> next token seen %.2f us (%d)\n", np ? proc / np : 0.0, np, nt ? turn / nt : 0.0, nt);
            std::vector<unsigned long long> e(4096 * 4);
            if (hipMemcpy(e.data(), c->tstamp + 32768, e.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                // [q][0] XCD 0's workgroup 0 is ready for the next token, [1] it has the arg-max of token q - 1, [2] token out (and the post of q - 1 checked), [3] lm workgroup 0 published its partials of token q
                double s01 = 0, s12 = 0, sp = 0, sdone = 0, sready = 0; int n = 0;
                for (uint32_t q = 3; q + 1 < hi; q++) {
                    const unsigned long long t0 = e[4 * q], t1 = e[4 * q + 1], t2 = e[4 * q + 2], pp = e[4 * (q - 1) + 3], dn = w[2 * (q - 1) + 1];
                    if (!t0 || !t1 || !t2 || !pp || !dn || t2 < t0 || t2 - t0 > 100000ull) continue;
                    s01 += (double)(long long)(t1 - t0) * 0.01; s12 += (double)(long long)(t2 - t1) * 0.01; sp += (double)(long long)(t1 - pp) * 0.01;
                    sdone += (double)(long long)(pp - dn) * 0.01; sready += (double)(long long)(t0 - dn) * 0.01; n++;
                }
                if (n) fprintf(stderr, "   XCD 0, workgroup 0: ready for the next token %.2f us after lm workgroup 0's completion word (its partials went out %.2f us after that word), arg-max %.2f us after ready = %.2f us after those partials, token out + post checked %.2f us later (%d)\n",
                               sready / n, sdone / n, s01 / n, sp / n, s12 / n, n);
            }
            std::vector<unsigned long long> d(64 * 256);
            if (hipMemcpy(d.data(), c->tstamp + 8192, d.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                double spread = 0.0, last_after_first = 0.0; int ns = 0; int late[8] = {};
                for (int q = 0; q < 64; q++) {
                    unsigned long long lo = ~0ull, hi2 = 0; int arg = -1;
                    for (int k = 0; k < c->res_nw && k < 256; k++) { const unsigned long long t = d[(size_t)q * 256 + k]; if (!t) continue; if (t < lo) lo = t; if (t > hi2) { hi2 = t; arg = k; } }
                    if (hi2 > lo && hi2 - lo < 100000ull) { spread += (double)(hi2 - lo) * 0.01; ns++; if (arg >= 0) late[(arg / 32) & 7]++; last_after_first += (double)(hi2 - d[(size_t)q * 256]) * 0.01; }
                }
                fprintf(stderr, "   completion words of one token: first -> last %.2f us, workgroup 0 -> last %.2f us (%d tokens); XCD group of the last one: %d %d %d %d %d %d\n",
                        ns ? spread / ns : 0.0, ns ? last_after_first / ns : 0.0, ns, late[0], late[1], late[2], late[3], late[4], late[5]);
            }
        }
    }
    if ((c->opt.res_dbg & 8) && c->res_calls > 0)
        fprintf(stderr, "resident evals: %ld calls, %.2f us waiting for the completion words, %.2f us between a return and the next post\n", c->res_calls, c->res_t_wait / c->res_calls * 1e6, c->res_t_call / c->res_calls * 1e6);
    for (auto &pl : c->graph_step) for (auto &row : pl) for (auto &g : row) if (g) (void)hipGraphExecDestroy(g);
    for (auto &pl : c->graph_eval) for (auto &f : pl) for (auto &row : f) for (auto &g : row) if (g) (void)hipGraphExecDestroy(g);
    for (auto &g : c->graph_batch) if (g) (void)hipGraphExecDestroy(g);
    xpipe_release(c);
    if (c->topk_host) (void)hipHostFree(c->topk_host);
    if (c->mbox_host) (void)hipHostFree(c->mbox_host);
    if (c->mbox_ctr) (void)hipFree(c->mbox_ctr);
    if (c->seq_dev) (void)hipFree(c->seq_dev);
    plain_graph_end(c);
    for (void *p : {(void *)c->bk, (void *)c->bv, (void *)c->seq, (void *)c->seq_gen, (void *)c->cols}) if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->owns_arena && c->arena) (void)hipFree(c->arena);
    for (void *p : {(void *)c->memory_k, (void *)c->memory_v, (void *)c->x, (void *)c->x1, (void *)c->q, (void *)c->att,
                    (void *)c->h, (void *)c->logits, (void *)c->aq_q[0], (void *)c->aq_q[1], (void *)c->aq_q[2], (void *)c->aq_d[0], (void *)c->aq_d[1], (void *)c->aq_d[2], (void *)c->aq_s[0], (void *)c->aq_s[1], (void *)c->aq_s[2], (void *)c->logits_all, (void *)c->pmax_val, (void *)c->pmax_idx, (void *)c->sp_scores, (void *)c->sp_max, (void *)c->sp_pv, (void *)c->tile_img, (void *)c->state})
        if (p) (void)hipFree(p);
    if (c->state_host) (void)hipHostFree(c->state_host);
    if (c->logits_host) (void)hipHostFree(c->logits_host);
    if (c->logits_host_alt) (void)hipHostFree(c->logits_host_alt);
    if (c->res_spec) (void)hipHostFree(c->res_spec);
    for (void *p : {(void *)c->logits_alt, (void *)c->pmax_val_alt, (void *)c->pmax_idx_alt}) if (p) (void)hipFree(p);
    if (c->tstamp) (void)hipFree(c->tstamp);
    if (c->tok_vocab) bg::drop_vocab(c->tok_vocab);
    delete c;
}

biogpt_hip_ctx *load_impl(const char *fname, int device, int verbosity, void *ext_arena, size_t ext_bytes) {
    clear_error();
    if (!fname) BG_FAIL(nullptr, "null file name");
    fprintf(stderr, "%s: loading model from '%s'\n", "biogpt_hip_load", fname);
    ModelFile mf;
    if (!mf.open(fname)) return nullptr;
    if (!select_device(device)) return nullptr;

    std::unique_ptr<biogpt_hip_ctx, void (*)(biogpt_hip_ctx *)> c(new biogpt_hip_ctx(), destroy);
    c->opt.load();
    c->hp = mf.hp;
    c->device = device;
    c->n_tensors = (int)mf.tensors.size();
    c->vocab = mf.vocab;
    c->merges = mf.merges;
    c->tok_vocab = bg::make_vocab(mf.vocab, mf.merges);
    const auto &hp = c->hp;
    if (verbosity > 0) {
        fprintf(stderr, "biogpt_hip_load: n_vocab = %d d_ff = %d d_model = %d n_positions = %d n_head = %d n_layer = %d ftype = %d n_merges = %d\n",
                hp.n_vocab, hp.d_ff, hp.d_model, hp.n_positions, hp.n_head, hp.n_layer, hp.ftype, hp.n_merges);
    }
    const int dk = hp.d_model / hp.n_head;
    if (hp.d_model % 32 != 0 || hp.d_ff % 32 != 0) BG_FAIL(nullptr, "d_model (%d) and d_ff (%d) must be multiples of 32", hp.d_model, hp.d_ff);
    if (dk % 4 != 0 || (dk & (dk - 1)) != 0 || dk > 256) BG_FAIL(nullptr, "head size %d unsupported (needs a power of two in [4, 256])", dk);

    // F5: the file's embed_positions must cover n_positions + 2 rows; a longer table is cut to that (rows past it are
    // never indexed), so that the arena layout depends on the 7 header ints only -- a replica that attaches to a
    // broadcast arena computes the same offsets without the file
    c->pos_rows = (int64_t)hp.n_positions + 2;
    if (const TensorEntry *pe = mf.find("biogpt.embed_positions.weight")) {
        if (pe->ne1 < c->pos_rows)
            BG_FAIL(nullptr, "tensor 'biogpt.embed_positions.weight' has wrong shape in model file: got [%lld, %lld], expected [%d, >=%lld]",
                    (long long)pe->ne0, (long long)pe->ne1, hp.d_model, (long long)c->pos_rows);
    }
    c->plan = plan_arena(hp, c->pos_rows);
    if (ext_arena) {
        if (ext_bytes < c->plan.total) BG_FAIL(nullptr, "external arena too small: %zu < %zu bytes", ext_bytes, c->plan.total);
        c->arena = static_cast<uint8_t *>(ext_arena);
        c->owns_arena = false;
    } else {
        HIP_TRY(nullptr, hipMalloc(reinterpret_cast<void **>(&c->arena), c->plan.total));
        c->owns_arena = true;
    }
    c->arena_bytes = c->plan.total;
    if (!alloc_runtime(c.get())) return nullptr;

    if (mf.tensors.empty()) {  // biogpt.cpp:442-443
        fprintf(stderr, "biogpt_hip_load: WARN no tensors loaded from model file - assuming empty model for testing\n");
        c->ready = false;
        return c.release();
    }
    if (!upload_weights(c.get(), mf)) return nullptr;
    HIP_TRY(nullptr, hipDeviceSynchronize());
    c->ready = true;
    xpipe_prepare(c.get());
    (void)fpipe_prepare(c.get());      // (float files: the persistent launch of kernels_fpipe.hip.h, prepared outside any capture)
    if (verbosity > 0)
        fprintf(stderr, "biogpt_hip_load: weight arena = %.2f MB, KV cache = %.2f MB, %d tensors\n", c->plan.total / 1048576.0,
                2.0 * hp.n_layer * hp.n_positions * hp.d_model * 4 / 1048576.0, c->n_tensors);
    return c.release();
}

// The single-token decode step is captured once per context bucket (the attention workgroup size is a
// launch parameter, everything else reads n_past / the token from HBM) and replayed per token.
// context buckets of the captured decode graphs: 64 / 128 / 192 / 256 / 512 / n_positions keys
int graph_bucket(int T) { return T <= 256 ? (T - 1) / 64 : (T <= 512 ? 4 : 5); }
int bucket_tmax(const biogpt_hip_ctx *c, int b) {
    const int t = b < 4 ? 64 * (b + 1) : (b == 4 ? 512 : c->hp.n_positions);
    return std::min(t, c->hp.n_positions);
}

// pl: 1 = the step as the XCD-pipelined launch (the caller holds the device's pipeline slot), 0 = the five-launch layer
bool ensure_graph(biogpt_hip_ctx *c, int advance, int bucket, int pl) {
    if (c->graph_step[pl][advance][bucket]) return true;
    hipGraph_t g = nullptr;
    HIP_TRY(false, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    // fused step (contexts up to 256 keys): the sampler of the previous token is the first kernel's prologue and the
    // lm_head kernel advances the position; otherwise embed ... lm_head + the arg-max kernel
    const int tmax = bucket_tmax(c, bucket);
    c->fp_force = pl;
    bool ok = fused_decode_ok(c, tmax) ? enqueue_decode_fused(c, tmax, 2, advance, 0, -1, -1, 1, nullptr, nullptr, pl)
                                       : (enqueue_forward(c, 1, false, tmax) && enqueue_argmax(c, advance));
    c->fp_force = -1;
    hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (!ok) { if (g) (void)hipGraphDestroy(g); return false; }
    HIP_TRY(false, e);
    HIP_TRY(false, hipGraphInstantiate(&c->graph_step[pl][advance][bucket], g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    return true;
}

}  // namespace

// ======================================== extern "C" ================================================
extern "C" {

const char *biogpt_hip_last_error(void) { return last_error(); }
const char *biogpt_hip_version(void) { return "biogpt-hip gfx950 r1"; }

biogpt_hip_ctx *biogpt_hip_load(const char *fname, int device, int verbosity) { return load_impl(fname, device, verbosity, nullptr, 0); }

biogpt_hip_ctx *biogpt_hip_load_into(const char *fname, int device, int verbosity, void *device_arena, size_t arena_bytes) {
    if (!device_arena) BG_FAIL(nullptr, "null arena");
    return load_impl(fname, device, verbosity, device_arena, arena_bytes);
}

size_t biogpt_hip_arena_bytes_for(const biogpt_hip_hparams *hp) {
    if (!hp || ftype_to_type(hp->ftype) == T_INVALID) return 0;
    return plan_arena(*hp, (int64_t)hp->n_positions + 2).total;
}

biogpt_hip_ctx *biogpt_hip_attach(const biogpt_hip_hparams *hp, int device, void *device_arena, size_t arena_bytes) {
    clear_error();
    if (!hp || !device_arena) BG_FAIL(nullptr, "null argument");
    if (ftype_to_type(hp->ftype) == T_INVALID) BG_FAIL(nullptr, "bad ftype value %d", hp->ftype);
    if (!select_device(device)) return nullptr;
    std::unique_ptr<biogpt_hip_ctx, void (*)(biogpt_hip_ctx *)> c(new biogpt_hip_ctx(), destroy);
    c->opt.load();
    c->hp = *hp;
    c->device = device;
    c->pos_rows = (int64_t)hp->n_positions + 2;
    c->plan = plan_arena(*hp, c->pos_rows);
    if (arena_bytes < c->plan.total) BG_FAIL(nullptr, "external arena too small: %zu < %zu bytes", arena_bytes, c->plan.total);
    c->arena = static_cast<uint8_t *>(device_arena);
    c->arena_bytes = c->plan.total;
    c->owns_arena = false;
    c->n_tensors = 5 + 16 * hp->n_layer;
    if (!alloc_runtime(c.get())) return nullptr;
    c->ready = true;
    xpipe_prepare(c.get());
    (void)fpipe_prepare(c.get());      // (float files: the persistent launch of kernels_fpipe.hip.h, prepared outside any capture)
    return c.release();
}

void *biogpt_hip_arena_ptr(biogpt_hip_ctx *ctx) { return ctx ? ctx->arena : nullptr; }
size_t biogpt_hip_arena_bytes(const biogpt_hip_ctx *ctx) { return ctx ? ctx->arena_bytes : 0; }

void biogpt_hip_free(biogpt_hip_ctx *ctx) { destroy(ctx); }

int biogpt_hip_share_vocab(biogpt_hip_ctx *dst, const biogpt_hip_ctx *src) {
    if (!dst || !src) BG_FAIL(-1, "null context");
    if (dst == src) return 0;
    if (dst->tok_vocab) { bg::drop_vocab(dst->tok_vocab); dst->tok_vocab = nullptr; }
    dst->vocab = src->vocab;
    dst->merges = src->merges;
    dst->tok_vocab = bg::make_vocab(dst->vocab, dst->merges);
    return 0;
}

int biogpt_hip_refresh_options(biogpt_hip_ctx *ctx) {
    if (!ctx) BG_FAIL(-1, "null context");
    if (!resident_stop(ctx)) return -2;
    ctx->opt.load();
    // captured graphs bake launch shapes chosen from the options
    for (auto &pl : ctx->graph_step) for (auto &row : pl) for (auto &g : row) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    for (auto &pl : ctx->graph_eval) for (auto &f : pl) for (auto &row : f) for (auto &g : row) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    for (auto &g : ctx->graph_batch) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    ctx->graph_batch_n = 0;
    // the pipelined path is rebuilt from the new options (GELU slice, fault hook, long-context buffers) -- which also re-arms a context that had abandoned the
    // path after a disturbed launch: an explicit call, not an automatic cool-down
    HIP_TRY(-2, hipSetDevice(ctx->device));
    HIP_TRY(-2, hipStreamSynchronize(ctx->stream));
    (void)xpipe_check(ctx);
    ctx->xp_tripped = false;
    xpipe_release(ctx);
    ctx->xp_state = 0; ctx->xp_gelu_p = 0; ctx->xp_gelu_n = 0; ctx->xp_gelu_z = 0; ctx->xp_exp_n = 0;
    xpipe_prepare(ctx);
    ctx->fp_state = 0;
    (void)fpipe_prepare(ctx);      // (float files: outside any capture)
    return 0;
}

int biogpt_hip_xpipe_state(const biogpt_hip_ctx *ctx) {
    if (!ctx) return -2;
    if (ctx->xp_state != 1 || !ctx->opt.xpipe) return ctx->xp_state == 1 ? 0 : ctx->xp_state;
    std::lock_guard<std::mutex> lk(g_xp_mu);
    const biogpt_hip_ctx *owner = (ctx->device >= 0 && ctx->device < 64) ? g_xp_owner[ctx->device] : nullptr;
    return (owner == nullptr || owner == ctx) ? 1 : 0;
}

int biogpt_hip_get_hparams(const biogpt_hip_ctx *ctx, biogpt_hip_hparams *out) {
    if (!ctx || !out) BG_FAIL(-1, "null argument");
    *out = ctx->hp;
    return 0;
}
int biogpt_hip_n_tensors(const biogpt_hip_ctx *ctx) { return ctx ? ctx->n_tensors : -1; }
const biogpt_hip_vocab *biogpt_hip_ctx_vocab(const biogpt_hip_ctx *ctx) { return ctx ? ctx->tok_vocab : nullptr; }

int biogpt_hip_vocab_token(const biogpt_hip_ctx *ctx, int32_t id, const char **bytes, int32_t *len) {
    if (!ctx || id < 0 || (size_t)id >= ctx->vocab.size()) return -1;
    if (bytes) *bytes = ctx->vocab[(size_t)id].data();
    if (len) *len = (int32_t)ctx->vocab[(size_t)id].size();
    return 0;
}
int biogpt_hip_merge(const biogpt_hip_ctx *ctx, int32_t rank, const char **bytes, int32_t *len) {
    if (!ctx || rank < 0 || (size_t)rank >= ctx->merges.size()) return -1;
    if (bytes) *bytes = ctx->merges[(size_t)rank].data();
    if (len) *len = (int32_t)ctx->merges[(size_t)rank].size();
    return 0;
}

#include "engine_resident.inc"

// low-latency wait for everything enqueued on the context's stream (the caller is blocked on this token anyway).  BIOGPT_HIP_EVAL_SYNC=1: hipStreamSynchronize instead
// (diagnostic)
static bool poll_stream(biogpt_hip_ctx *ctx) {
    if (ctx->opt.eval_sync == 1) { HIP_TRY(false, hipStreamSynchronize(ctx->stream)); return true; }
    if (ctx->opt.eval_sync == 2) { HIP_TRY(false, hipDeviceSynchronize()); return true; }
    if (ctx->opt.eval_sync == 3) { HIP_TRY(false, hipEventRecord(ctx->ev1, ctx->stream)); HIP_TRY(false, hipEventSynchronize(ctx->ev1)); return true; }
    for (;;) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) return true;
        if (q != hipErrorNotReady) HIP_TRY(false, q);
    }
}

// the k largest values of a row, descending, equal values: lower index first (topk_kernel's order) -- ONE pass: a block of 16 values is only looked at when its
// maximum beats the current k-th value.  Returns how many were found (< k only when the row holds NaNs)
static int host_topk(const float *row, int n, int k, float *vals, int32_t *ids) {
    int have = 0;
    float thr = -INFINITY;
    auto offer = [&](float v, int i) {
        if ((have == k && !(v > thr)) || v != v) return;
        int pos = have < k ? have : k - 1;
        while (pos > 0 && vals[pos - 1] < v) { vals[pos] = vals[pos - 1]; ids[pos] = ids[pos - 1]; pos--; }
        vals[pos] = v; ids[pos] = i;
        if (have < k) have++;
        if (have == k) thr = vals[k - 1];
    };
    int i = 0;
    for (; i < n && have < k; i++) offer(row[i], i);
    for (; i + 16 <= n; i += 16) {
        float m = -INFINITY;      // (not row[i]: a NaN there would hide the block -- every comparison with it is false)
        for (int j = 0; j < 16; j++) m = row[i + j] > m ? row[i + j] : m;
        if (m > thr)
            for (int j = 0; j < 16; j++) offer(row[i + j], i + j);
    }
    for (; i < n; i++) offer(row[i], i);
    return have;
}

// The same selection when the launch has left the maxima of the row's 64-row blocks behind it (a resident launch): the k-th largest block maximum t0 is a lower
// bound of the k-th largest logit (k blocks hold a value >= t0), so every candidate lies in a block whose maximum is >= t0 -- k blocks, give or take ties, instead of
// n / 64.  They are walked in index order with the same insertion rule, so values, ids and their order are those of host_topk (and of topk_kernel).
static int host_topk_blocks(const float *row, int n, int k, const float *bmax, int nblocks, float *vals, int32_t *ids) {
    if (nblocks < k || nblocks > 1024) return host_topk(row, n, k, vals, ids);
    float top[64];
    int have_b = 0;
    for (int b = 0; b < nblocks; b++) {      // the k largest block maxima, descending (insertion into <= 64 slots: ~700 compares for BioGPT's 663 blocks)
        const float v = bmax[b];
        if (v != v) return host_topk(row, n, k, vals, ids);
        if (have_b == k && !(v > top[k - 1])) continue;
        int pos = have_b < k ? have_b : k - 1;
        while (pos > 0 && top[pos - 1] < v) { top[pos] = top[pos - 1]; pos--; }
        top[pos] = v;
        if (have_b < k) have_b++;
    }
    const float t0 = top[k - 1];
    int have = 0;
    float thr = -INFINITY;
    for (int b = 0; b < nblocks; b++) {
        if (!(bmax[b] >= t0)) continue;
        const int i0 = b * 64, i1 = std::min(n, i0 + 64);
        for (int i = i0; i < i1; i++) {
            const float v = row[i];
            if ((have == k && !(v > thr)) || v != v) continue;
            int pos = have < k ? have : k - 1;
            while (pos > 0 && vals[pos - 1] < v) { vals[pos] = vals[pos - 1]; ids[pos] = ids[pos - 1]; pos--; }
            vals[pos] = v; ids[pos] = i;
            if (have < k) have++;
            if (have == k) thr = vals[k - 1];
        }
    }
    return have;
}

static int eval_topk_once(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n, int32_t n_past, int32_t k, float *vals_out, int32_t *ids_out) {
    if (!vals_out || !ids_out) BG_FAIL(-1, "null output buffer");
    if (!ctx) BG_FAIL(-1, "null context");
    if (k < 1 || k > 64) BG_FAIL(-1, "k must be in [1, 64]");
    k = std::min<int32_t>(k, ctx->hp.n_vocab);
    if (n == 1 && ctx->opt.resident && !ctx->opt.no_graph) {
        // a loop of single-token calls: the resident launch serves it (no launch per call -- the selection kernel could not run beside it anyway), the selection
        // runs over the pinned row on the host; the same k pairs in the same order as topk_kernel's
        XpCallScope xp_scope(ctx);
        clear_error();
        if (!check_eval_args(ctx, tokens, n, n_past)) return -1;
        if (!ctx->res_live) HIP_TRY(-2, hipSetDevice(ctx->device));
        const int r = resident_eval(ctx, tokens[0], n_past);
        if (r < 0) return r;
        if (r == 1) {
            const float *bmax = ctx->row_cur + bgk::xp_blockmax_offset(ctx->hp.n_vocab);
            const int got = ctx->opt.topk_blocks ? host_topk_blocks(ctx->row_cur, ctx->hp.n_vocab, k, bmax, ctx->lm_blocks, vals_out, ids_out)
                                                 : host_topk(ctx->row_cur, ctx->hp.n_vocab, k, vals_out, ids_out);
            if (got == k) return k;
            // NaNs in the row: the device path below has a defined answer for it (the position is evaluated again: same inputs, same K / V row)
        }
    }
    const int rc = biogpt_hip_eval_device(ctx, tokens, n, n_past);
    if (rc) return rc;
    if (!ctx->topk_host) HIP_TRY(-2, hipHostMalloc(reinterpret_cast<void **>(&ctx->topk_host), 64 * 8 + 16, hipHostMallocDefault));
    // the kernel writes its <= 64 pairs straight into pinned host memory: no copy command behind it
    float *out_val = reinterpret_cast<float *>(ctx->topk_host);
    int32_t *out_idx = reinterpret_cast<int32_t *>(ctx->topk_host + 64 * 4);
    hipLaunchKernelGGL(bgk::topk_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->logits, ctx->hp.n_vocab, k, ctx->pmax_val, ctx->lm_blocks,
                       out_val, out_idx, out_idx + 64);
    HIP_TRY(-2, hipGetLastError());
    uint32_t *const got = (ctx->dev_stamp_expect != 0 && ctx->mbox_host) ? reinterpret_cast<uint32_t *>(ctx->mbox_host + 64 * 8) + 1 : nullptr;
    if (got) HIP_TRY(-2, hipMemcpyAsync(got, ctx->seq_dev + bgk::SEQ_LM_HEAD, 4, hipMemcpyDeviceToHost, ctx->stream));   // the device row's lineage (a replayed five-launch step)
    // low-latency wait: poll the stream instead of sleeping on it (the caller is blocked on this token anyway)
    if (!poll_stream(ctx)) return -2;
    ctx->mbox_synced = ctx->mbox_sent;
    if (!xpipe_check(ctx)) return -2;
    if (got) {
        const uint32_t want = ctx->dev_stamp_expect;
        ctx->dev_stamp_expect = 0;
        if (*got != want) {      // the row is another call's: the call again on eager launches, the selection again
            ctx->stale_rows++;
            if (ctx->opt.verbose) fprintf(stderr, "biogpt_hip[%p]: the replayed eval at position %d left the row of call %u instead of %u on the device: repeated on eager launches\n", (void *)ctx, n_past, *got, want);
            uint32_t *const fix = reinterpret_cast<uint32_t *>(ctx->mbox_host + 64 * 8);
            *fix = ctx->mbox_sent;
            HIP_TRY(-2, hipMemcpyAsync(ctx->mbox_ctr, fix, 4, hipMemcpyHostToDevice, ctx->stream));
            if (!upload_state(ctx, tokens, n, n_past) || !enqueue_forward(ctx, n, false, n_past + n)) return -2;
            hipLaunchKernelGGL(bgk::topk_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->logits, ctx->hp.n_vocab, k, ctx->pmax_val, ctx->lm_blocks,
                               out_val, out_idx, out_idx + 64);
            HIP_TRY(-2, hipGetLastError());
            if (!poll_stream(ctx)) return -2;
            if (!xpipe_check(ctx)) return -2;
        }
    }
    const float *hv = reinterpret_cast<const float *>(ctx->topk_host);
    const int32_t *hi = reinterpret_cast<const int32_t *>(ctx->topk_host + 64 * 4);
    if (hi[64] == k) {
        std::memcpy(vals_out, hv, (size_t)k * 4);
        std::memcpy(ids_out, hi, (size_t)k * 4);
        return k;
    }
    // degenerate row (thousands of logits tie with the k-th workgroup maximum, or NaNs): the full row and a host partial sort
    const size_t V = (size_t)ctx->hp.n_vocab;
    std::vector<float> row(V);
    HIP_TRY(-2, hipMemcpy(row.data(), ctx->logits, V * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> order(V);
    for (size_t i = 0; i < V; i++) order[i] = (int32_t)i;
    std::partial_sort(order.begin(), order.begin() + k, order.end(), [&](int32_t a, int32_t b) { return row[(size_t)a] > row[(size_t)b] || (row[(size_t)a] == row[(size_t)b] && a < b); });
    for (int i = 0; i < k; i++) { vals_out[i] = row[(size_t)order[(size_t)i]]; ids_out[i] = order[(size_t)i]; }
    return k;
}

// arg-max with the lowest index winning ties (what std::max_element returns), eight independent running maxima so that the compiler can keep them in
// one vector register: ~4 us for 42 k logits against ~20 us for the scalar loop
static int32_t argmax_first(const float *v, size_t n) {
    float best[8]; int32_t at[8];
    for (int j = 0; j < 8; j++) { best[j] = -INFINITY; at[j] = 0x7fffffff; }
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int j = 0; j < 8; j++)
            if (v[i + j] > best[j]) { best[j] = v[i + j]; at[j] = (int32_t)(i + j); }
    for (; i < n; i++)
        if (v[i] > best[i & 7] ) { best[i & 7] = v[i]; at[i & 7] = (int32_t)i; }
    float b = -INFINITY; int32_t a = 0x7fffffff;
    for (int j = 0; j < 8; j++)
        if (at[j] != 0x7fffffff && (best[j] > b || (best[j] == b && at[j] < a))) { b = best[j]; a = at[j]; }
    return a == 0x7fffffff ? 0 : a;
}

// The reference's host loop (main.cpp:91-151, greedy) as a C++ caller would run it on this library -- one eval call per
// token, the sampler on the host -- timed without any scripting-language overhead: mode 0 = biogpt_hip_eval (the whole
// logits row crosses PCIe, host arg-max), mode 1 = biogpt_hip_eval_topk with k = 40 (the CLI's top_k; 512 bytes cross).
int biogpt_hip_bench_api_loop(biogpt_hip_ctx *ctx, const int32_t *prompt, int32_t n_prompt, int32_t n_predict, int32_t mode, int32_t *out_ids,
                              double *seconds_out) {
    clear_error();
    if (!ctx || !prompt || n_prompt < 1 || n_predict < 1 || mode < 0 || mode > 4) BG_FAIL(-1, "bad argument");
    if (!check_eval_args(ctx, prompt, n_prompt, 0)) return -1;
    n_predict = std::min(n_predict, ctx->hp.n_positions - n_prompt);
    const size_t V = (size_t)ctx->hp.n_vocab;
    std::vector<float> logits(mode == 0 ? V : 64);
    int32_t ids[64];
    const auto t0 = std::chrono::steady_clock::now();
    int32_t tok = 0;
    int n_past = 0;
    for (int k = 0; k < n_predict; k++) {
        const int32_t *in = (k == 0) ? prompt : &tok;
        const int n_in = (k == 0) ? n_prompt : 1;
        if (mode == 0) {
            if (biogpt_hip_eval(ctx, in, n_in, n_past, logits.data()) != 0) return -2;
            tok = (int32_t)(std::max_element(logits.begin(), logits.end()) - logits.begin());
        } else if (mode == 3 || mode == 4) {   // the row read in place (pinned host memory the launch wrote), arg-max over 8 interleaved lanes; mode 4 (diagnostic): no arg-max, token fixed
            const float *row = nullptr;
            if (biogpt_hip_eval_inplace(ctx, in, n_in, n_past, &row) != 0) return -2;
            tok = mode == 3 ? argmax_first(row, V) : 2;
        } else if (mode == 1) {
            if (biogpt_hip_eval_topk(ctx, in, n_in, n_past, 40, logits.data(), ids) < 0) return -2;
            tok = ids[0];
        } else {   // mode 2 (diagnostic): the eval and a stream synchronise only -- no output leaves the device; token fixed
            if (biogpt_hip_eval_device(ctx, in, n_in, n_past) != 0) return -2;
            HIP_TRY(-2, hipStreamSynchronize(ctx->stream));
            ctx->mbox_synced = ctx->mbox_sent;
    if (!xpipe_check(ctx)) return -2;
            tok = 2;
        }
        n_past += n_in;
        if (out_ids) out_ids[k] = tok;
    }
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return n_predict;
}

const float *biogpt_hip_logits_device(const biogpt_hip_ctx *ctx) {
    if (!ctx) return nullptr;
    // a resident launch may hold the last accepted row in its alternate buffer (odd sequence numbers) and may be running one position ahead: it is stopped here -- its
    // results are folded into the ordinary buffers -- so that the pointer names the row of the last evaluated token, as the header says
    if (ctx->res_live || (ctx->res_acc & 1u)) (void)resident_stop(const_cast<biogpt_hip_ctx *>(ctx));
    return ctx->logits;
}

int biogpt_hip_read_logits(biogpt_hip_ctx *ctx, float *out) {
    if (!ctx || !out) BG_FAIL(-1, "null argument");
    clear_error();
    HIP_TRY(-2, hipSetDevice(ctx->device));
    if (!resident_stop(ctx)) return -2;
    HIP_TRY(-2, hipMemcpyAsync(out, ctx->logits, (size_t)ctx->hp.n_vocab * 4, hipMemcpyDeviceToHost, ctx->stream));
    uint32_t *const got = ctx->mbox_host ? reinterpret_cast<uint32_t *>(ctx->mbox_host + 64 * 8) + 1 : nullptr;
    if (ctx->dev_stamp_expect != 0 && got) HIP_TRY(-2, hipMemcpyAsync(got, ctx->seq_dev + bgk::SEQ_LM_HEAD, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(-2, hipStreamSynchronize(ctx->stream));
    ctx->mbox_synced = ctx->mbox_sent;
    if (!xpipe_check(ctx)) return -2;
    if (ctx->dev_stamp_expect != 0 && got) {
        // the device row of a replayed five-launch eval (biogpt_hip_eval_device): the same lineage check as eval_once's, the same repair
        const uint32_t want = ctx->dev_stamp_expect;
        ctx->dev_stamp_expect = 0;
        if (*got != want) {
            ctx->stale_rows++;
            if (ctx->opt.verbose) fprintf(stderr, "biogpt_hip[%p]: the replayed eval at position %d left the row of call %u instead of %u on the device: repeated on eager launches\n", (void *)ctx, ctx->dev_stamp_n_past, *got, want);
            uint32_t *const fix = reinterpret_cast<uint32_t *>(ctx->mbox_host + 64 * 8);
            *fix = ctx->mbox_sent;
            HIP_TRY(-2, hipMemcpyAsync(ctx->mbox_ctr, fix, 4, hipMemcpyHostToDevice, ctx->stream));
            const int32_t tok = ctx->dev_stamp_tok;
            if (!upload_state(ctx, &tok, 1, ctx->dev_stamp_n_past) || !enqueue_forward(ctx, 1, false, ctx->dev_stamp_n_past + 1)) return -2;
            HIP_TRY(-2, hipMemcpyAsync(out, ctx->logits, (size_t)ctx->hp.n_vocab * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(-2, hipStreamSynchronize(ctx->stream));
            if (!xpipe_check(ctx)) return -2;
        }
    }
    return 0;
}

// {single-token evals replayed as captured five-launch steps, rows found to be another call's and repeated} (kernels.hip.h: SEQ_*)
int biogpt_hip_lineage_stats(biogpt_hip_ctx *ctx, int64_t *out2) {
    if (!ctx || !out2) BG_FAIL(-1, "null argument");
    out2[0] = ctx->graph_evals; out2[1] = ctx->stale_rows;
    return 0;
}

int biogpt_hip_synchronize(biogpt_hip_ctx *ctx) {
    if (!ctx) BG_FAIL(-1, "null context");
    if (!resident_stop(ctx)) return -2;
    HIP_TRY(-2, hipStreamSynchronize(ctx->stream));
    if (!xpipe_check(ctx)) {      // reported to the caller here: there is no call to repeat, and a later, unrelated failure must not trigger a spurious retry
        ctx->xp_tripped = false;
        const int from = ctx->unsynced_from;
        ctx->unsynced_from = -1;
        if (from >= 0) BG_FAIL(-2, "the pipelined decode step failed; the evals from position %d on must be repeated (the context now uses the five-launch layer)", from);
        return -2;
    }
    return 0;
}

// logits_out == nullptr: the row stays in the context's pinned host buffer (biogpt_hip_eval_inplace)
static int eval_once(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n, int32_t n_past, float *logits_out) {
    XpCallScope xp_scope(ctx);
    if (n == 1) {      // one token: a launch that stays on the device between the calls of the caller's loop
        clear_error();
        if (!check_eval_args(ctx, tokens, n, n_past)) return -1;
        if (!ctx->res_live) HIP_TRY(-2, hipSetDevice(ctx->device));      // a live resident launch needs no HIP call at all
        const int r = resident_eval(ctx, tokens[0], n_past);
        if (r < 0) return r;
        if (r == 1) { if (logits_out) std::memcpy(logits_out, ctx->row_cur, (size_t)ctx->hp.n_vocab * 4); return 0; }
    }
    bool in_graph = false;      // the replayed graph already wrote the pinned row
    const int rc = eval_device_impl(ctx, tokens, n, n_past, 1, &in_graph);
    if (rc) return rc;
    const size_t bytes = (size_t)ctx->hp.n_vocab * 4;
    if (!in_graph) {   // device -> pinned staging -> caller's (pageable) buffer: one DMA instead of the runtime's chunked staging
        if (!ctx->logits_host) HIP_TRY(-2, hipHostMalloc(reinterpret_cast<void **>(&ctx->logits_host), host_row_bytes(ctx), hipHostMallocDefault));
        HIP_TRY(-2, hipMemcpyAsync(ctx->logits_host, ctx->logits, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!poll_stream(ctx)) return -2;   // poll: the caller is blocked on this token anyway
    ctx->mbox_synced = ctx->mbox_sent;
    if (!xpipe_check(ctx)) return -2;
    if (in_graph && ctx->stamp_expect != 0) {
        // the row of a replayed five-launch graph carries the sequence number its FIRST node fetched, forwarded by the last layer and the lm_head as they started: any
        // other number means that the row is not this call's (profiles/two_contexts_r4.txt, profiles/stale_row_r5.txt).  Repair: the device's replay counter is put
        // right and the call is repeated on eager launches (same token, same position: the K / V row is written again with the same values).
        const uint32_t got = reinterpret_cast<const uint32_t *>(ctx->logits_host)[host_stamp_index(ctx)];
        const uint32_t want = ctx->stamp_expect;
        ctx->stamp_expect = 0;
        if (got != want) {
            ctx->stale_rows++;
            if (ctx->opt.verbose) fprintf(stderr, "biogpt_hip[%p]: the replayed eval at position %d returned the row of call %u instead of %u: repeated on eager launches\n", (void *)ctx, n_past, got, want);
            uint32_t *const fix = reinterpret_cast<uint32_t *>(ctx->mbox_host + 64 * 8);
            *fix = ctx->mbox_sent;
            HIP_TRY(-2, hipMemcpyAsync(ctx->mbox_ctr, fix, 4, hipMemcpyHostToDevice, ctx->stream));
            if (!upload_state(ctx, tokens, n, n_past) || !enqueue_forward(ctx, n, false, n_past + n)) return -2;
            HIP_TRY(-2, hipMemcpyAsync(ctx->logits_host, ctx->logits, bytes, hipMemcpyDeviceToHost, ctx->stream));
            if (!poll_stream(ctx)) return -2;
            if (!xpipe_check(ctx)) return -2;
        }
    }
    ctx->row_cur = ctx->logits_host;
    if (logits_out) std::memcpy(logits_out, ctx->logits_host, bytes);
    return 0;
}

int biogpt_hip_eval_all(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n, int32_t n_past, float *logits_out) {
    clear_error();
    if (!logits_out) BG_FAIL(-1, "null logits buffer");
    if (!check_eval_args(ctx, tokens, n, n_past)) return -1;
    HIP_TRY(-2, hipSetDevice(ctx->device));
    if (!resident_stop(ctx)) return -2; disarm_lineage(ctx);
    if ((size_t)n > ctx->logits_all_rows) {
        if (ctx->logits_all) (void)hipFree(ctx->logits_all);
        ctx->logits_all = nullptr;
        HIP_TRY(-2, hipMalloc(&ctx->logits_all, (size_t)n * ctx->hp.n_vocab * 4));
        ctx->logits_all_rows = (size_t)n;
        // the captured batched-decode graphs hold the old logits_all pointer
        for (auto &g : ctx->graph_batch) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
        ctx->graph_batch_n = 0;
    }
    if (!upload_state(ctx, tokens, n, n_past)) return -2;
    if (!enqueue_forward(ctx, n, true, n_past + n)) return -2;
    HIP_TRY(-2, hipMemcpyAsync(logits_out, ctx->logits_all, (size_t)n * ctx->hp.n_vocab * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(-2, hipStreamSynchronize(ctx->stream));
    return 0;
}

// Prompt ingestion = consecutive evals of n_batch tokens (main.cpp:129-137), each attending to everything before
// it and to its own chunk (no mask inside an eval, F1).  Nothing else couples the chunks, so up to
// BIOGPT_HIP_PROMPT_COLS columns (several chunks) go through the layers in ONE pass -- every weight byte is
// streamed once for all of them (default 512 columns) -- with the attention of column i limited to the keys its own chunk would
// have seen (DevState::chunk).  Per-column arithmetic is unchanged: logits and KV rows are bit-identical to
// the chunk-by-chunk evaluation.  Leaves the last token's logits in ctx->logits.
bool enqueue_prompt(biogpt_hip_ctx *c, const int32_t *tokens, int n, int n_past, int n_batch, int *last_cols = nullptr) {
    const int max_cols = std::max(1, c->opt.prompt_cols);   // measured (Q4_0, -b 8, 512-token prompt): 16 -> 17.9k, 64 -> 36k, 128 -> 66k, 256 -> 87k, 512 -> 97k prompt tok/s
    const int group = n_batch >= max_cols ? n_batch : (max_cols / n_batch) * n_batch;   // whole chunks per pass
    if (std::min(group, n) >= c->opt.mfma_min(64) && is_quantized(ftype_to_type(c->hp.ftype)) && !ensure_tile_images(c)) return false;
    for (int at = 0; at < n;) {
        const int m = std::min(group, n - at);
        if (!upload_state(c, tokens + at, m, n_past + at, m > n_batch ? n_batch : 0)) return false;
        if (!enqueue_forward(c, m, false, n_past + at + m)) return false;
        if (last_cols) *last_cols = m;   // the device state now describes this pass (its n_past, its m tokens)
        at += m;
    }
    return true;
}

static int eval_prompt_once(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n_tokens, int32_t n_past, int32_t n_batch, float *logits_out) {
    XpCallScope xp_scope(ctx);
    clear_error();
    if (n_batch < 1) BG_FAIL(-1, "n_batch must be >= 1");
    if (!check_eval_args(ctx, tokens, n_tokens, n_past)) return -1;
    HIP_TRY(-2, hipSetDevice(ctx->device));
    if (!resident_stop(ctx)) return -2; disarm_lineage(ctx);
    if (!enqueue_prompt(ctx, tokens, n_tokens, n_past, n_batch)) return -2;
    if (logits_out) {
        HIP_TRY(-2, hipMemcpyAsync(logits_out, ctx->logits, (size_t)ctx->hp.n_vocab * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(-2, hipStreamSynchronize(ctx->stream));
        ctx->mbox_synced = ctx->mbox_sent;
        if (!xpipe_check(ctx)) return -2;      // a prompt of up to 8 tokens is a pipelined launch (kernels_xcols.hip.h; one token: kernels_xpipe.hip.h)
    }
    return 0;
}

static int generate_greedy_once(biogpt_hip_ctx *ctx, const int32_t *prompt, int32_t n_prompt, int32_t n_batch,
                                int32_t n_predict, int32_t *out_ids, double *seconds_out) {
    XpCallScope xp_scope(ctx);
    clear_error();
    if (!ctx || !prompt || !out_ids) BG_FAIL(-1, "null argument");
    if (n_batch < 1) BG_FAIL(-1, "n_batch must be >= 1");
    if (n_prompt < 1) BG_FAIL(-1, "empty prompt");
    if (!check_eval_args(ctx, prompt, n_prompt, 0)) return -1;
    n_predict = std::min(n_predict, ctx->hp.n_positions - n_prompt);  // main.cpp:82
    if (n_predict <= 0) return 0;
    HIP_TRY(-2, hipSetDevice(ctx->device));
    if (!resident_stop(ctx)) return -2; disarm_lineage(ctx);
    bool use_graph = ctx->opt.no_graph == 0;
    if ((ctx->opt.dbg & 128) && !ctx->tstamp) {      // profiling builds: stage stamps of the pipelined launches (tools/tail_timeline.py)
        HIP_TRY(-2, hipMalloc(&ctx->tstamp, (size_t)4 << 20));
        HIP_TRY(-2, hipMemset(ctx->tstamp, 0, (size_t)4 << 20));
    }
    // the device's pipeline slot, if it is free, is this call's until its final synchronisation
    // Decided ONCE per bucket, here: xpipe_usable() can take the slot over from a holder that has gone idle in the meantime, so asking again at replay
    // time could name a graph that was never captured.  The graph replayed for a bucket is the one instantiated for it.
    int pl_bucket[6] = {0, 0, 0, 0, 0, 0};
    auto pl_of = [&](int b) { return pl_bucket[b]; };
    if (use_graph)  // instantiate every bucket this run will touch before the clock starts
        for (int b = graph_bucket(n_prompt + 1); b <= graph_bucket(n_prompt + n_predict - 1 > 0 ? n_prompt + n_predict - 1 : 1); b++) {
            pl_bucket[b] = pipeline_pl(ctx, bucket_tmax(ctx, b));
            if (!ensure_graph(ctx, 1, b, pl_bucket[b])) return -2;
        }
    if (use_graph) {   // captured five-launch steps are not replayed beside ANOTHER context's persistent launch (plain_graph_begin): such a run takes eager steps
        bool any_plain = false;
        for (int b = graph_bucket(n_prompt + 1); b <= graph_bucket(n_prompt + n_predict - 1 > 0 ? n_prompt + n_predict - 1 : 1); b++) any_plain = any_plain || pl_bucket[b] == 0;
        if (any_plain && !plain_graph_begin(ctx)) use_graph = false;
    }
    HIP_TRY(-2, hipStreamSynchronize(ctx->stream));

    const auto t0 = std::chrono::steady_clock::now();
    // prompt ingestion in chunks of n_batch (main.cpp:129-137), several chunks per pass; then the first sampled token.
    // Fused steps (graph replay, contexts up to 256 keys) sample the PREVIOUS token in their first kernel, so between
    // them no sampler kernel runs: `pending` = the last lm_head's arg-max has not been recorded yet.
    auto fused = [&](int T) { return use_graph && fused_decode_ok(ctx, bucket_tmax(ctx, graph_bucket(T))); };
    int last_cols = 0;
    if (!enqueue_prompt(ctx, prompt, n_prompt, 0, n_batch, &last_cols)) return -2;
    bool pending = n_predict > 1 && fused(n_prompt + 1);
    if (pending) {
        hipLaunchKernelGGL(bgk::advance_state_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->state, last_cols);
        HIP_TRY(-2, hipGetLastError());
    } else if (!enqueue_argmax(ctx, last_cols)) {
        return -2;
    }
    ctx->gen_launches = 0;
    for (int k = 1; k < n_predict; k++) {  // one eval + one sample per further token
        const int T = n_prompt + k;  // keys visible to this token: n_past + 1
        if (pending && !fused(T)) {  // leaving the fused range: record the token the unfused step will embed
            if (!enqueue_argmax(ctx, 0)) return -2;
            pending = false;
        }
        const int multi = pending ? std::min(xpipe_multi_tokens(ctx, T), n_predict - k) : 0;
        if (multi > 1) {   // the generation loop itself runs on the device: `multi` tokens in ONE launch (kernels_xpipe.hip.h)
            if (!enqueue_decode_fused(ctx, bucket_tmax(ctx, graph_bucket(T)), 2, 1, 0, -1, -1, multi)) return -2;
            if (ctx->gen_launches < 16) ctx->gen_launch_tokens[ctx->gen_launches++] = multi;
            k += multi - 1;
        } else if (use_graph) {
            HIP_TRY(-2, hipGraphLaunch(ctx->graph_step[pl_of(graph_bucket(T))][1][graph_bucket(T)], ctx->stream));
        } else {
            if (!enqueue_forward(ctx, 1, false, T) || !enqueue_argmax(ctx, 1)) return -2;
        }
    }
    if (pending && !enqueue_argmax(ctx, 0)) return -2;   // the last token's sampler
    HIP_TRY(-2, hipStreamSynchronize(ctx->stream));
    if (!xpipe_check(ctx)) return -2;
    const auto t1 = std::chrono::steady_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
    HIP_TRY(-2, hipMemcpy(out_ids, reinterpret_cast<uint8_t *>(ctx->state) + sizeof(bgk::DevState) + (size_t)ctx->hp.n_positions * 4,
                          (size_t)n_predict * 4, hipMemcpyDeviceToHost));
    return n_predict;
}

// The XCD-pipelined decode step assumes that its 256 workgroups are spread 32 per XCD and become resident together; kernels of
// other streams or processes dispatched in between can break that (the launch then drains with an error word and garbage
// outputs).  These calls are idempotent for given arguments, so they are simply repeated once on the five-launch layer, which the
// context keeps from then on.
// n_past: the position the failed call evaluates.  Earlier asynchronous single-token evals (biogpt_hip_eval_device) that were still in flight went through the same
// tripped pipeline: their K / V rows are garbage and repeating only THIS call would return wrong logits with rc 0 -- then the caller is told where to resume instead.
static bool xpipe_retry(biogpt_hip_ctx *ctx, int n_past) {
    if (!ctx || !ctx->xp_tripped) return false;
    ctx->xp_tripped = false;
    const int from = ctx->unsynced_from;
    ctx->unsynced_from = -1;
    fprintf(stderr, "biogpt_hip: %s\n", last_error());
    if (from >= 0 && from < n_past) {
        bg::set_error("biogpt_hip_eval", "the pipelined decode step failed while evals from position %d on were still in flight: their K / V rows are invalid -- re-evaluate from n_past = %d (the context now uses the five-launch layer)", from, from);
        return false;
    }
    return true;
}
int biogpt_hip_eval_inplace(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n, int32_t n_past, const float **row_out) {
    if (!row_out) BG_FAIL(-1, "null row pointer");
    int rc = eval_once(ctx, tokens, n, n_past, nullptr);
    if (rc != 0 && xpipe_retry(ctx, n_past)) rc = eval_once(ctx, tokens, n, n_past, nullptr);
    *row_out = rc == 0 ? ctx->row_cur : nullptr;
    return rc;
}
int64_t biogpt_hip_chunk_launches(const biogpt_hip_ctx *ctx) { return ctx ? ctx->xc_launches : -1; }
int biogpt_hip_fpipe_stamps(biogpt_hip_ctx *ctx, uint64_t *out, int n) {
    if (!ctx || !out || n < 0 || ctx->fp_state != 1 || !ctx->opt.fpipe_stamps) return -1;
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
    const int m = n < 3 * 1024 ? n : 3 * 1024;
    return hipMemcpy(out, ctx->fp_ctl + 16, (size_t)m * 8, hipMemcpyDeviceToHost) == hipSuccess ? m : -1;
}
int64_t biogpt_hip_fpipe_launches(const biogpt_hip_ctx *ctx) { return ctx ? (ctx->fp_state == 1 ? ctx->fp_launches : -1) : -2; }
int biogpt_hip_generate_launches(const biogpt_hip_ctx *ctx, int32_t *tokens_out, int cap) {
    if (!ctx) return -1;
    for (int i = 0; i < ctx->gen_launches && i < cap && tokens_out; i++) tokens_out[i] = ctx->gen_launch_tokens[i];
    return ctx->gen_launches;
}
int biogpt_hip_resident_stats(const biogpt_hip_ctx *ctx, int64_t *out4) {
    if (!ctx || !out4) return -1;
    out4[0] = ctx->spec_hits; out4[1] = ctx->spec_misses; out4[2] = ctx->spec_streak; out4[3] = ctx->spec_need;
    return 0;
}
int biogpt_hip_eval_prompt(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n_tokens, int32_t n_past, int32_t n_batch, float *logits_out) {
    int rc = eval_prompt_once(ctx, tokens, n_tokens, n_past, n_batch, logits_out);
    if (rc != 0 && xpipe_retry(ctx, n_past)) rc = eval_prompt_once(ctx, tokens, n_tokens, n_past, n_batch, logits_out);
    return rc;
}
int biogpt_hip_eval(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n, int32_t n_past, float *logits_out) {
    if (!logits_out) BG_FAIL(-1, "null logits buffer");
    int rc = eval_once(ctx, tokens, n, n_past, logits_out);
    if (rc != 0 && xpipe_retry(ctx, n_past)) rc = eval_once(ctx, tokens, n, n_past, logits_out);
    return rc;
}
int biogpt_hip_eval_topk(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n, int32_t n_past, int32_t k, float *vals_out, int32_t *ids_out) {
    int rc = eval_topk_once(ctx, tokens, n, n_past, k, vals_out, ids_out);
    if (rc < 0 && xpipe_retry(ctx, n_past)) rc = eval_topk_once(ctx, tokens, n, n_past, k, vals_out, ids_out);
    return rc;
}
int biogpt_hip_generate_greedy(biogpt_hip_ctx *ctx, const int32_t *prompt, int32_t n_prompt, int32_t n_batch, int32_t n_predict, int32_t *out_ids,
                               double *seconds_out) {
    int rc = generate_greedy_once(ctx, prompt, n_prompt, n_batch, n_predict, out_ids, seconds_out);
    if (rc < 0 && xpipe_retry(ctx, 0)) rc = generate_greedy_once(ctx, prompt, n_prompt, n_batch, n_predict, out_ids, seconds_out);
    return rc;
}

static int hp_cols(const biogpt_hip_ctx *c) { return c->hp.n_positions; }   // scratch is sized [n_positions] columns

static int generate_greedy_batch_once(biogpt_hip_ctx *ctx, const int32_t *prompts, const int32_t *prompt_lens, int32_t n_seqs,
                                      int32_t n_batch, int32_t n_predict, int32_t *out_ids, double *seconds_out) {
    XpCallScope xp_scope(ctx);
    struct XcBatchScope { biogpt_hip_ctx *c; ~XcBatchScope() { if (c) c->xc_batch = 0; } } xc_scope{ctx};
    clear_error();
    if (!ctx || !prompts || !prompt_lens || !out_ids) BG_FAIL(-1, "null argument");
    if (!ctx->ready) BG_FAIL(-1, "model has no tensors loaded (empty model): cannot evaluate");
    if (n_seqs < 1 || n_seqs > 512) BG_FAIL(-1, "n_seqs must be in [1, 512]");   // activation buffers hold n_positions >= 512 columns; each sequence owns a full F32 KV cache
    if (n_seqs > hp_cols(ctx)) BG_FAIL(-1, "n_seqs (%d) exceeds the %d activation columns of this model", n_seqs, hp_cols(ctx));
    if (n_batch < 1) BG_FAIL(-1, "n_batch must be >= 1");
    const auto &hp = ctx->hp;
    const int P = hp.n_positions, D = hp.d_model, V = hp.n_vocab;
    int max_len = 0;
    {
        size_t off = 0;
        for (int s = 0; s < n_seqs; s++) {
            if (prompt_lens[s] < 1) BG_FAIL(-1, "empty prompt (sequence %d)", s);
            if (!check_eval_args(ctx, prompts + off, prompt_lens[s], 0)) return -1;
            max_len = std::max(max_len, prompt_lens[s]);
            off += (size_t)prompt_lens[s];
        }
    }
    n_predict = std::min(n_predict, P - max_len);  // main.cpp:82, for the longest prompt
    if (n_predict <= 0) return 0;
    HIP_TRY(-2, hipSetDevice(ctx->device));
    if (!resident_stop(ctx)) return -2; disarm_lineage(ctx);
    const size_t seq_stride = (size_t)hp.n_layer * P * D;
    if (n_seqs > ctx->batch_cap) {  // per-sequence F32 KV caches + state (192 MiB per BioGPT-base sequence)
        for (void *p : {(void *)ctx->bk, (void *)ctx->bv, (void *)ctx->seq, (void *)ctx->seq_gen}) if (p) (void)hipFree(p);
        ctx->bk = ctx->bv = nullptr; ctx->seq = nullptr; ctx->seq_gen = nullptr; ctx->batch_cap = 0;
        HIP_TRY(-2, hipMalloc(&ctx->bk, seq_stride * 4 * n_seqs));
        HIP_TRY(-2, hipMalloc(&ctx->bv, seq_stride * 4 * n_seqs));
        HIP_TRY(-2, hipMalloc(&ctx->seq, sizeof(bgk::SeqState) * n_seqs));
        HIP_TRY(-2, hipMalloc(&ctx->seq_gen, (size_t)n_seqs * P * 4));
        ctx->batch_cap = n_seqs;
        for (auto &g : ctx->graph_batch) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    }
    {   // matrix-core chain: decode steps have n_seqs columns, the prompt pass all prompt tokens; build the tiled weights before any graph capture
        long total = 0;
        for (int s = 0; s < n_seqs; s++) total += prompt_lens[s];
        if (std::max<long>(n_seqs, total) >= ctx->opt.mfma_min(48) && !ensure_tile_images(ctx)) return -2;
    }
    if ((size_t)n_seqs > ctx->logits_all_rows) {
        if (ctx->logits_all) (void)hipFree(ctx->logits_all);
        ctx->logits_all = nullptr;
        HIP_TRY(-2, hipMalloc(&ctx->logits_all, (size_t)n_seqs * V * 4));
        ctx->logits_all_rows = (size_t)n_seqs;
        for (auto &g : ctx->graph_batch) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    }
    if (ctx->graph_batch_n != n_seqs) {
        for (auto &g : ctx->graph_batch) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
        ctx->graph_batch_n = n_seqs;
    }
    std::vector<bgk::SeqState> hs((size_t)n_seqs);
    for (int s = 0; s < n_seqs; s++) {   // the prompt pass below leaves the LAST prompt token to the first decode step
        hs[(size_t)s] = bgk::SeqState{};
        hs[(size_t)s].n_past = prompt_lens[s] - 1;
        hs[(size_t)s].seq_id = s;
    }
    {
        size_t o = 0;
        for (int s = 0; s < n_seqs; s++) { o += (size_t)prompt_lens[s]; hs[(size_t)s].token = prompts[o - 1]; }
    }
    HIP_TRY(-2, hipMemcpy(ctx->seq, hs.data(), sizeof(bgk::SeqState) * n_seqs, hipMemcpyHostToDevice));

    // 2 .. 8 sequences: the decode steps as column-per-XCD launches while this call holds the device's pipeline slot (decided ONCE, before any capture: a graph is
    // replayed only in the state it was captured for -- graph_batch[6 * pl + bucket])
    ctx->xc_batch = (n_seqs >= 2 && n_seqs <= 8 && max_len + 1 <= 256 && xcols_prepare(ctx, n_seqs, std::min(256, max_len + 1)) && xpipe_usable(ctx, 256)) ? 1 : 0;
    const int pl = ctx->xc_batch;
    auto batch_step = [&](int t_max) -> bool {
        if (!enqueue_forward(ctx, n_seqs, false, t_max, true)) return false;
        hipLaunchKernelGGL(bgk::argmax_rows_kernel, dim3(n_seqs), dim3(1024), 0, ctx->stream, ctx->logits_all, V, V, ctx->seq, 0, ctx->seq_gen, P, 1);
        HIP_TRY(false, hipGetLastError());
        return true;
    };
    const bool use_graph = ctx->opt.no_graph == 0 && (pl != 0 || plain_graph_begin(ctx));      // (a captured five-launch step is not replayed beside another context's persistent launch)
    if (use_graph) {
        for (int b = graph_bucket(max_len + 1); b <= graph_bucket(std::max(1, max_len + n_predict - 1)); b++) {
            if (ctx->graph_batch[6 * pl + b]) continue;
            hipGraph_t g = nullptr;
            HIP_TRY(-2, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            const bool ok = batch_step(bucket_tmax(ctx, b));
            hipError_t e = hipStreamEndCapture(ctx->stream, &g);
            if (!ok) { if (g) (void)hipGraphDestroy(g); return -2; }
            HIP_TRY(-2, e);
            HIP_TRY(-2, hipGraphInstantiate(&ctx->graph_batch[6 * pl + b], g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
        }
    }
    HIP_TRY(-2, hipStreamSynchronize(ctx->stream));

    const auto t0 = std::chrono::steady_clock::now();
    // Prompt ingestion for ALL sequences together (main.cpp:129-137 per sequence): every prompt token is a column that
    // knows its sequence, its position and the end of its own n_batch-chunk (SeqState::seq_id / n_past / t_vis); whole
    // chunks are packed into passes of up to BIOGPT_HIP_PROMPT_COLS columns.  The pass only has to fill the KV caches:
    // the first decode step below re-evaluates each sequence's LAST prompt token (same K/V row, same visible keys as
    // its chunk gave it) and its arg-max is the first sampled token.
    {
        const int max_cols = std::min(std::max(std::max(1, ctx->opt.prompt_cols), n_batch), hp_cols(ctx));   // the activation scratch holds n_positions columns
        std::vector<bgk::SeqState> cols;
        int pass_tmax = 0;
        auto flush = [&]() -> bool {
            if (cols.empty()) return true;
            if (cols.size() > ctx->cols_cap) {
                if (ctx->cols) (void)hipFree(ctx->cols);
                ctx->cols = nullptr; ctx->cols_cap = 0;
                HIP_TRY(false, hipMalloc(&ctx->cols, sizeof(bgk::SeqState) * cols.size()));
                ctx->cols_cap = cols.size();
            }
            HIP_TRY(false, hipMemcpyAsync(ctx->cols, cols.data(), sizeof(bgk::SeqState) * cols.size(), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(false, hipStreamSynchronize(ctx->stream));   // the host vector is reused for the next pass
            if (!enqueue_forward(ctx, (int)cols.size(), false, pass_tmax, true, ctx->cols)) return false;
            cols.clear();
            pass_tmax = 0;
            return true;
        };
        size_t off = 0;
        for (int s = 0; s < n_seqs; s++) {
            const int len = prompt_lens[s];
            for (int at = 0; at < len; at += n_batch) {
                const int m = std::min(n_batch, len - at);
                if (!cols.empty() && (int)cols.size() + m > max_cols && !flush()) return -2;
                for (int i = 0; i < m; i++) {
                    bgk::SeqState cst{};
                    cst.n_past = at + i; cst.token = prompts[off + (size_t)(at + i)]; cst.seq_id = s; cst.t_vis = at + m;
                    cols.push_back(cst);
                }
                pass_tmax = std::max(pass_tmax, at + m);
            }
            off += (size_t)len;
        }
        if (!flush()) return -2;
    }
    if (!batch_step(max_len)) return -2;   // last prompt token of every sequence -> first sampled token, n_past = prompt length
    for (int k = 1; k < n_predict; k++) {  // batched decode: one column per sequence
        const int t_max = max_len + k;
        if (use_graph) HIP_TRY(-2, hipGraphLaunch(ctx->graph_batch[6 * pl + graph_bucket(t_max)], ctx->stream));
        else if (!batch_step(t_max)) return -2;
    }
    HIP_TRY(-2, hipStreamSynchronize(ctx->stream));
    const auto t1 = std::chrono::steady_clock::now();
    if (!xpipe_check(ctx)) return -2;      // (steps as column-per-XCD launches: a disturbed one spoils the run -- the caller below repeats it on the launch chain)
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
    std::vector<int32_t> gen((size_t)n_seqs * P);
    HIP_TRY(-2, hipMemcpy(gen.data(), ctx->seq_gen, gen.size() * 4, hipMemcpyDeviceToHost));
    for (int s = 0; s < n_seqs; s++) std::memcpy(out_ids + (size_t)s * n_predict, gen.data() + (size_t)s * P, (size_t)n_predict * 4);
    return n_predict;
}
int biogpt_hip_generate_greedy_batch(biogpt_hip_ctx *ctx, const int32_t *prompts, const int32_t *prompt_lens, int32_t n_seqs,
                                     int32_t n_batch, int32_t n_predict, int32_t *out_ids, double *seconds_out) {
    int rc = generate_greedy_batch_once(ctx, prompts, prompt_lens, n_seqs, n_batch, n_predict, out_ids, seconds_out);
    if (rc < 0 && xpipe_retry(ctx, 0)) rc = generate_greedy_batch_once(ctx, prompts, prompt_lens, n_seqs, n_batch, n_predict, out_ids, seconds_out);
    return rc;
}

// profiling builds (BIOGPT_HIP_PROFILE_HOOKS + BIOGPT_HIP_DBG=128): the raw 100 MHz stage stamps the pipelined launches left (kernels_xpipe.hip.h XP_WALL / XP_TAIL)
int biogpt_hip_debug_stamps(biogpt_hip_ctx *ctx, size_t offset, size_t count, unsigned long long *out) {
    clear_error();
    if (!ctx || !out) BG_FAIL(-1, "null argument");
    const size_t cap = ((size_t)4 << 20) / 8;                 // stamps in the buffer; checked without a sum that could wrap
    if (!ctx->tstamp || offset > cap || count > cap - offset) BG_FAIL(-1, "no stamps (profiling build with BIOGPT_HIP_DBG=128 only)");
    HIP_TRY(-2, hipSetDevice(ctx->device));
    if (!resident_stop(ctx)) return -2;
    HIP_TRY(-2, hipStreamSynchronize(ctx->stream));
    HIP_TRY(-2, hipMemcpy(out, ctx->tstamp + offset, count * 8, hipMemcpyDeviceToHost));
    return 0;
}

int biogpt_hip_read_kv(biogpt_hip_ctx *ctx, int which, size_t offset, size_t count, float *out) {
    if (!ctx || !out) BG_FAIL(-1, "null argument");
    const size_t L = (size_t)ctx->hp.n_layer, P = (size_t)ctx->hp.n_positions, D = (size_t)ctx->hp.d_model, H = (size_t)ctx->hp.n_head;
    const size_t dk = D / H, total = L * P * D;
    if (offset > total || count > total - offset) BG_FAIL(-1, "KV range out of bounds");      // (no sum that could wrap)
    if (count == 0) return 0;
    HIP_TRY(-2, hipSetDevice(ctx->device));
    if (!resident_stop(ctx)) return -2;
    // the device cache is head-major [layer][head][pos][dk]; the caller sees the reference's flat [layer][pos][d_model]
    // view (biogpt.cpp:331-335).  Only the requested range is gathered on the device and copied.
    float *stage = nullptr;
    HIP_TRY(-2, hipMalloc(&stage, count * 4));
    const float *cache = which ? ctx->memory_v : ctx->memory_k;
    hipLaunchKernelGGL(bgk::kv_gather_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, cache, stage, (unsigned long long)offset,
                       (unsigned long long)count, (int)P, (int)D, (int)H, (int)dk);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, stage, count * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(stage);
    HIP_TRY(-2, e);
    return 0;
}

#include "engine_bench.inc"
// SURVEY 8 f1 on the device: `nrows` rows of `k` f32 values (host memory) -> the file's block format of `type`, byte-identical to
// the host quantizer (biogpt_hip_quantize_file uses the host one: it has to work without a GPU)
int biogpt_hip_quantize_rows_device(int device, int32_t type, const float *src, int64_t nrows, int64_t k, uint8_t *dst) {
    clear_error();
    if (!src || !dst || nrows < 1 || k < QK || k % QK) BG_FAIL(-1, "bad argument (row length must be a multiple of %d)", QK);
    if (!is_quantized(type)) BG_FAIL(-1, "type %d is not a block-quantized format", type);
    if (!select_device(device)) return -1;
    const long long nblocks = (long long)nrows * (k / QK);
    const size_t in_bytes = (size_t)nrows * (size_t)k * 4, out_bytes = (size_t)nblocks * file_block_bytes(type);
    float *d_src = nullptr;
    uint8_t *d_dst = nullptr;
    HIP_TRY(-2, hipMalloc(&d_src, in_bytes));
    if (hipMalloc(&d_dst, out_bytes) != hipSuccess) { (void)hipFree(d_src); BG_FAIL(-2, "hipMalloc of %zu bytes failed", out_bytes); }
    hipError_t e = hipMemcpy(d_src, src, in_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bgk::quantize_blocks_kernel, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, 0, d_src, d_dst, nblocks, (int)type,
                           (int)file_block_bytes(type));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(dst, d_dst, out_bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d_src); (void)hipFree(d_dst);
    HIP_TRY(-2, e);
    return 0;
}

int biogpt_hip_quantize_file(const char *fname_in, const char *fname_out, int32_t ftype) {
    clear_error();
    if (!fname_in || !fname_out) BG_FAIL(-1, "null file name");
    return quantize_file(fname_in, fname_out, ftype) ? 0 : -1;
}

int biogpt_hip_write_synthetic(const char *fname, const biogpt_hip_hparams *hp, uint64_t seed) {
    clear_error();
    if (!fname || !hp) BG_FAIL(-1, "null argument");
    return write_synthetic(fname, *hp, seed) ? 0 : -1;
}

}  // extern "C"
