/*
 * include/biogpt_hip.h -- C-ABI of the MI355X-native BioGPT decoder engine.
 *
 * This is the drop-in boundary for the reference's model-library path:
 *
 *   reference (C++, biogpt.h)                         this library (extern "C")
 *   ------------------------------------------------  ---------------------------------------
 *   biogpt_model_load()      biogpt.h:128-132         biogpt_hip_load()
 *                            biogpt.cpp:27-453
 *   biogpt_eval()            biogpt.h:145-151         biogpt_hip_eval()
 *                            biogpt.cpp:812-847
 *   biogpt_graph()           biogpt.h:139-143         (internal: fixed launch sequence / hipGraph;
 *                            biogpt.cpp:624-810        nothing to size, see biogpt_compat.h)
 *   generation loop          main.cpp:91-151          biogpt_hip_generate_greedy()  (device-resident
 *   + top_k=1 sampler        biogpt.cpp:908-980        loop: argmax + token feedback stay in HBM)
 *   biogpt_model_quantize_internal + quantize CLI     biogpt_hip_quantize_file()
 *                            biogpt.cpp:459-621, quantize.cpp:8-135
 *   gpt_tokenize             biogpt.cpp:850-875       biogpt_hip_tokenize()         (host-only: Moses word
 *   gpt_decode               biogpt.cpp:877-906       biogpt_hip_decode()            splitting + byte BPE)
 *   teardown                 main.cpp:164-169         biogpt_hip_free()
 *
 * Plain pointers and sizes only: no C++/torch types cross this boundary.  The C++ wrappers with
 * the reference's exact signatures live in include/biogpt_compat.h and are implemented on top of
 * these entry points.  All functions are thread-compatible per context (one eval at a time per
 * context; different contexts -- e.g. one per GPU -- may run concurrently).
 *
 * Error convention: the reference returns false + fprintf(stderr).  Here: pointer
 * results are NULL on failure, int results are 0 on success and negative on failure; the message
 * is printed to stderr (prefixed like the reference's "%s: ...") and kept for
 * biogpt_hip_last_error().  There is NO CPU fallback: without a usable HIP device every compute
 * entry point fails loudly.
 */
#ifndef BIOGPT_HIP_H
#define BIOGPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct biogpt_hip_ctx biogpt_hip_ctx;
typedef struct biogpt_hip_vocab biogpt_hip_vocab;   /* vocabulary + merge ranks of one model file (host memory) */

/* tokenizer status: the reference's moses_tokenize throws std::length_error for this input (see below) */
#define BIOGPT_HIP_E_LENGTH (-7)

/* header ints in file order (biogpt.cpp:54-60) + the merge count found in the file */
typedef struct biogpt_hip_hparams {
    int32_t n_vocab;
    int32_t n_layer;
    int32_t n_head;
    int32_t n_positions;
    int32_t d_ff;
    int32_t d_model;
    int32_t ftype;    /* ggml_ftype: 0 f32, 1 f16, 2 q4_0, 3 q4_1, 7 q8_0, 8 q5_0, 9 q5_1 */
    int32_t n_merges;
} biogpt_hip_hparams;

/* Last error message of the calling thread ("" if none). */
const char *biogpt_hip_last_error(void);

/* Library / build identification, e.g. "biogpt-hip gfx950 r1". */
const char *biogpt_hip_version(void);

/* ---- load / free --------------------------------------------------------------------------
 * biogpt_hip_load: parse `fname` (ggml-model.bin), repack every tensor into
 * the device arena on HIP device `device`, allocate the F32 KV cache (biogpt.cpp:324-358).
 * Replaces biogpt_model_load (biogpt.cpp:27-453); same validation and the same failure cases
 * (bad magic, bad vocab size, bad ftype, unknown tensor, wrong shape/size, missing tensors);
 * a file with zero tensors loads with a warning (biogpt.cpp:442-443) and cannot be evaluated.
 * Unlike the reference the merge count is taken from the file (F6) and embed_positions is sized
 * by the file (F5). */
biogpt_hip_ctx *biogpt_hip_load(const char *fname, int device, int verbosity);

/* Same, but the weight arena lives in caller-owned device memory (e.g. a torch uint8 tensor that
 * is then broadcast over RCCL); arena_bytes must be >= biogpt_hip_arena_bytes_for(). */
biogpt_hip_ctx *biogpt_hip_load_into(const char *fname, int device, int verbosity,
                                     void *device_arena, size_t arena_bytes);

/* Create a context around an arena that ALREADY holds repacked weights (received by broadcast from
 * a rank that loaded the file): no file access.  hp must equal the loading rank's hparams. */
biogpt_hip_ctx *biogpt_hip_attach(const biogpt_hip_hparams *hp, int device,
                                  void *device_arena, size_t arena_bytes);

/* Size in bytes of the device weight arena for these hparams (deterministic layout). */
size_t biogpt_hip_arena_bytes_for(const biogpt_hip_hparams *hp);

/* The context's arena (owned or external) -- what a multi-GPU launcher broadcasts. */
void  *biogpt_hip_arena_ptr(biogpt_hip_ctx *ctx);
size_t biogpt_hip_arena_bytes(const biogpt_hip_ctx *ctx);

void biogpt_hip_free(biogpt_hip_ctx *ctx);

/* An attached context (biogpt_hip_attach) has weights but no vocabulary: copy the token / merge tables of a context
 * that was loaded from the file, so that every replica can tokenize and decode (biogpt.cpp:850-906). */
int biogpt_hip_share_vocab(biogpt_hip_ctx *dst, const biogpt_hip_ctx *src);

/* ---- single-process multi-GPU replicas (replaces the ONE biogpt_model_load of examples/main/main.cpp:38) ----
 * The file is read once on devices[0]; the packed weight arena goes to the other devices with ONE RCCL broadcast
 * (ncclCommInitAll, in-process; librccl.so is resolved at run time); every device gets its own context.  Prompt g is
 * served by replica g mod n, one host thread per device, no collective on the data path.  NULL + message on failure. */
typedef struct biogpt_hip_replicas biogpt_hip_replicas;
biogpt_hip_replicas *biogpt_hip_replicas_load(const char *fname, const int *devices, int n_devices, int verbosity);
int biogpt_hip_replicas_count(const biogpt_hip_replicas *r);
biogpt_hip_ctx *biogpt_hip_replicas_ctx(biogpt_hip_replicas *r, int i);   /* borrowed: freed by biogpt_hip_replicas_free */
double biogpt_hip_replicas_broadcast_seconds(const biogpt_hip_replicas *r);
/* Greedy continuations (main.cpp:91-151 with --top_k 1) of n_prompts independent prompts (ids concatenated, lengths in
 * prompt_lens): out_ids[g][n_predict], out_counts[g] = ids produced for prompt g (n_predict clamped like main.cpp:82; may
 * be NULL); *seconds_out = wall time of the whole call.  Returns n_prompts or < 0. */
int biogpt_hip_replicas_generate_greedy(biogpt_hip_replicas *r, const int32_t *prompts, const int32_t *prompt_lens, int32_t n_prompts,
                                        int32_t n_batch, int32_t n_predict, int32_t *out_ids, int32_t *out_counts, double *seconds_out);
void biogpt_hip_replicas_free(biogpt_hip_replicas *r);

/* The BIOGPT_HIP_* tuning / debugging switches are read from the environment once, when a context is
 * created (no getenv on any launch path).  This re-reads them for an existing context and drops its
 * captured graphs (tests and sweep tools; the reference has no counterpart). */
int biogpt_hip_refresh_options(biogpt_hip_ctx *ctx);

/* The single-token decode step of BioGPT-base-shaped block-quantized models (Q4_0 .. Q8_0) with at most 256 keys runs as ONE
 * persistent launch pipelined over the 8 XCDs (csrc/kernels_xpipe.hip.h).  1: this context uses it; 0: switched off
 * (BIOGPT_HIP_XPIPE=0) or another context of the device holds the path; -1: not available (model shape / weight type /
 * device; another PROCESS holds the device's lock file) or abandoned after a disturbed launch (the
 * call was repeated on the five-launch layer). */
int biogpt_hip_xpipe_state(const biogpt_hip_ctx *ctx);

int biogpt_hip_get_hparams(const biogpt_hip_ctx *ctx, biogpt_hip_hparams *out);
int biogpt_hip_n_tensors(const biogpt_hip_ctx *ctx); /* tensors found in the file (389 for BioGPT) */

/* vocab / merges as read from the file (biogpt.cpp:72-156); pointers stay valid until free */
int biogpt_hip_vocab_token(const biogpt_hip_ctx *ctx, int32_t id, const char **bytes, int32_t *len);
int biogpt_hip_merge(const biogpt_hip_ctx *ctx, int32_t rank, const char **bytes, int32_t *len);

/* ---- eval ---------------------------------------------------------------------------------
 * One forward pass over n_tokens tokens at offset n_past; writes the LAST token's n_vocab logits
 * to host memory.  Replaces biogpt_eval (biogpt.cpp:812-847): same op order, no intra-chunk
 * causal mask (F1), K/V rows appended to the F32 cache at [n_past, n_past+n_tokens).
 * Requires 0 <= n_past, n_past + n_tokens <= n_positions, ids in [0, n_vocab).
 * n_threads of the reference has no meaning here and is not a parameter. */
int biogpt_hip_eval(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n_tokens, int32_t n_past,
                    float *logits_out);

/* biogpt_eval (biogpt.cpp:812-847) without the final copy: *row_out points at the context's pinned host buffer that the launch
 * itself wrote the n_vocab logits of the last token into (valid until the next call on this context).  A loop of single-token calls
 * (main.cpp:91-151) -- through this entry or biogpt_hip_eval -- is served by ONE pipelined launch that stays on the device between
 * the calls and takes each next token from a pinned mailbox; it leaves the device after BIOGPT_HIP_RESIDENT_US
 * (default 1000) microseconds without a call, or as soon as the context is asked to do anything else. */
int biogpt_hip_eval_inplace(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n_tokens, int32_t n_past,
                            const float **row_out);

/* The launch behind a loop of single-token evals may run AHEAD of a greedy caller: once four consecutive calls have named exactly the
 * arg-max of the row before (lowest id on ties -- what std::max_element over the logits returns, main.cpp:109-128 with top_k = 1), the
 * launch starts the next position from its own arg-max while the caller still reads the row, and the next call only confirms the token
 * (any other token / position ends that launch, costs one token's time, and doubles the number of matching calls needed before the
 * next attempt -- up to 64; BIOGPT_HIP_SPEC=0 switches it off).  Results are the same either way.  out4 = {calls served by a pass that was already
 * running, calls that named a different token than the running pass, current run of matching calls, matches needed}. */
int biogpt_hip_resident_stats(const biogpt_hip_ctx *ctx, int64_t *out4);

/* biogpt_eval with 2 .. 8 tokens -- the chunks of the reference's prompt loop (main.cpp:129-137, n_batch = 8) -- of a BioGPT-base-shaped block-quantized
 * (Q4_0 .. Q8_0) model at up to 256 keys runs as ONE persistent launch with one column per XCD (csrc/kernels_xcols.hip.h; BIOGPT_HIP_XCOLS=0: the launch chain of
 * kernels_fast.hip.h; same results bit for bit).  biogpt_hip_generate_greedy_batch with 2 .. 8 sequences runs its decode steps through the same kernel (one sequence per
 * XCD, its own K / V cache: no exchange between XCDs) while contexts stay within 256 keys.  Returns how many such launches this context has enqueued or captured (evals;
 * batched generation: one per captured context bucket + the first step) (-1: null context). */
int64_t biogpt_hip_chunk_launches(const biogpt_hip_ctx *ctx);
/* F32 / F16 files: single-token steps this context has enqueued (or captured into a replayed step) as ONE persistent launch for all layers (csrc/kernels_fpipe.hip.h);
 * -1: that launch is not available to this context (shape, device, BIOGPT_HIP_FPIPE=0, or abandoned after a failed launch) -- five launches per layer then */
int64_t biogpt_hip_fpipe_launches(const biogpt_hip_ctx *ctx);
/* Diagnostics of that launch (BIOGPT_HIP_FPIPE_STAMPS=1): stage-border times (s_memrealtime, 100 MHz) of workgroups 0, 128 and 255 of the LAST launch, [3][32 layers][32]:
 * entries 0..8 the first computing wave (A in, A out, B in, C in, C dot, D in, D out, E in, E dot), 16..21 the polling wave (inputs of A, B, C, D, E first / second half seen).
 * Returns the number of values copied, -1 when the launch or the option is not active. */
int biogpt_hip_fpipe_stamps(biogpt_hip_ctx *ctx, uint64_t *out, int n);
/* the last biogpt_hip_generate_greedy call (the loop main.cpp:91-151 with --top_k 1): how many multi-token pipelined launches it took and the tokens of each (up to cap);
 * 0 when every token was a launch / graph replay of its own.  Measurement aid: per-launch durations of a kernel trace divide by these. */
int biogpt_hip_generate_launches(const biogpt_hip_ctx *ctx, int32_t *tokens_out, int cap);

/* Same pass, logits stay in HBM (no PCIe); biogpt_hip_logits_device() returns the device pointer
 * to the n_vocab floats of the last evaluated token.  eval_device is asynchronous on the
 * context's stream; biogpt_hip_synchronize() waits for it. */
int          biogpt_hip_eval_device(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n_tokens, int32_t n_past);
const float *biogpt_hip_logits_device(const biogpt_hip_ctx *ctx);
int          biogpt_hip_read_logits(biogpt_hip_ctx *ctx, float *out);   /* waits for the context's work, then copies that device row (n_vocab floats) to host memory */
int          biogpt_hip_synchronize(biogpt_hip_ctx *ctx);

/* biogpt_eval + the top-k selection of biogpt_sample_top_k_top_p (biogpt.cpp:929-936) on the device: evaluates like
 * biogpt_hip_eval and returns the k <= 64 largest logits of the last token (descending; equal logits: lower id first)
 * and their ids (main.cpp:98-128 only ever samples from the top_k = 40 candidates).  A prompt chunk: the selection runs on the device and
 * 512 bytes instead of the 170 KB row cross PCIe.  A single token -- the caller's loop: the resident launch of biogpt_hip_eval_inplace serves
 * the call and the same selection runs in one pass over the pinned row on the host (no launch per call).  Returns k (clamped to n_vocab) or < 0. */
int biogpt_hip_eval_topk(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n_tokens, int32_t n_past, int32_t k,
                         float *vals_out, int32_t *ids_out);

/* All-rows variant used by parity tests: logits_out is [n_tokens][n_vocab] on the host
 * (the reference computes all rows and returns the last, biogpt.cpp:803,844; F8). */
int biogpt_hip_eval_all(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n_tokens, int32_t n_past,
                        float *logits_out);

/* Prompt ingestion: the result of calling biogpt_hip_eval() on consecutive chunks of n_batch tokens
 * (main.cpp:129-137 with -b n_batch) -- same KV rows, same logits for the last token, bit for bit -- but several
 * chunks travel through the layers per pass (column i only attends to the keys its own chunk would have seen),
 * so the weights are streamed once per pass of up to 512 tokens (BIOGPT_HIP_PROMPT_COLS) instead of once per n_batch.  logits_out may be NULL: the call
 * is then asynchronous (biogpt_hip_logits_device() / biogpt_hip_synchronize()). */
int biogpt_hip_eval_prompt(biogpt_hip_ctx *ctx, const int32_t *tokens, int32_t n_tokens, int32_t n_past, int32_t n_batch,
                           float *logits_out);

/* Greedy generation harness = main.cpp:91-151 with --top_k 1: prompt fed in chunks of n_batch,
 * then n_predict (clamped to n_positions - n_prompt, main.cpp:82) tokens are sampled by arg-max
 * (lowest id wins ties) and fed back, all inside HBM (one captured hipGraph replay per token).
 * out_ids receives the sampled ids; *seconds_out (may be NULL) the wall time of the eval part
 * (main.cpp:96-103 equivalent: prompt evals + decode evals, the last sampled token is not
 * evaluated).  Returns the number of ids written, negative on error. */
int biogpt_hip_generate_greedy(biogpt_hip_ctx *ctx, const int32_t *prompt, int32_t n_prompt,
                               int32_t n_batch, int32_t n_predict, int32_t *out_ids,
                               double *seconds_out);

/* Batched greedy generation: n_seqs independent sequences decoded together on ONE device, one activation
 * column per sequence (own F32 KV cache, own position), so every weight byte is read once per step for all
 * sequences.  Each sequence's ids are identical to what biogpt_hip_generate_greedy() returns for it alone
 * (per-column arithmetic is unchanged).  No counterpart in the reference (it decodes one sequence); this is
 * the single-GPU form of "independent prompts" sharding.  prompts = the prompts concatenated,
 * prompt_lens[n_seqs] their lengths; n_predict is clamped to n_positions - max(prompt_lens); out_ids is
 * [n_seqs][returned n_predict].  Needs the BioGPT-base fast chain (block-quantized weights); n_seqs <= 512 (each
 * sequence owns a full F32 KV cache: 192 MiB at BioGPT-base); from 48 sequences the chain runs on the int8 matrix cores. */
int biogpt_hip_generate_greedy_batch(biogpt_hip_ctx *ctx, const int32_t *prompts, const int32_t *prompt_lens,
                                     int32_t n_seqs, int32_t n_batch, int32_t n_predict, int32_t *out_ids,
                                     double *seconds_out);

/* ---- introspection for tests / profiling ---------------------------------------------------
 * Copy `count` floats of the F32 KV cache (which: 0 = K, 1 = V) starting at element `offset` of
 * the flat [n_layer][n_positions][d_model] array (biogpt.cpp:331-335) to host memory. */
int biogpt_hip_read_kv(biogpt_hip_ctx *ctx, int which, size_t offset, size_t count, float *out);

/* Profiling builds only (-DBIOGPT_HIP_PROFILE_HOOKS, BIOGPT_HIP_DBG=128; no reference counterpart): the raw wall-clock stamps (100 MHz ticks) the
 * pipelined decode launches left in the context's stamp buffer, `count` words from word `offset` (tools/tail_timeline.py). */
int biogpt_hip_debug_stamps(biogpt_hip_ctx *ctx, size_t offset, size_t count, unsigned long long *out);

/* Stand-alone launch of the block-quantized mat-vec kernel on a weight matrix of the loaded
 * model, for kernel-level roofline timing: which = 0 fc1 of layer `layer`, 1 fc2,
 * 2 q/k/v fused, 3 out_proj, 4 lm_head (5-11: internal, see bench.py), 12 lm_head with the weights taken from a
 * different one of 14 device copies every launch (344 MB: not resident in the Infinity Cache), 13 (float files) all layers of one token at 104 keys as the
 * ONE persistent launch of kernels_fpipe.hip.h.  Runs `reps` back-to-back launches on the context's
 * stream bracketed by HIP events and returns the average seconds per launch in *seconds_out and
 * the algorithmic bytes of one launch in *bytes_out. */
int biogpt_hip_bench_matvec(biogpt_hip_ctx *ctx, int which, int layer, int reps,
                            double *seconds_out, double *bytes_out);

/* The decode mat-vec kernel (LayerNorm + Q4_0 W*A8 mat-vec, d_model = 1024 columns) on a synthetic weight
 * stream of `rows` rows -- far larger than the caches when rows >= 2^19 (302 MB): measures what the kernel
 * body sustains per byte when launch latency is amortised (SURVEY 8d roofline check).  `steps` = row steps per
 * wave.  HIP-event timed; *bytes_out = algorithmic bytes of one launch. */
int biogpt_hip_bench_stream(biogpt_hip_ctx *ctx, int32_t rows, int reps, int steps, double *seconds_out, double *bytes_out);

/* Timed replay of the captured single-token decode graph at a fixed n_past (no token feedback
 * side effects beyond the KV row at n_past): average seconds per token over `reps` replays,
 * HIP-event timed on the context's stream. */
int biogpt_hip_bench_decode(biogpt_hip_ctx *ctx, int32_t n_past, int reps, double *seconds_out);

/* The reference's greedy host loop (main.cpp:91-151: one eval call per token, sampler on the host) run in C++ on this
 * library and timed as a whole: mode 0 = biogpt_hip_eval + host arg-max (the logits row crosses PCIe every token),
 * mode 1 = biogpt_hip_eval_topk with k = 40.  out_ids (may be NULL) receives the n_predict ids.  Measurement aid. */
int biogpt_hip_bench_api_loop(biogpt_hip_ctx *ctx, const int32_t *prompt, int32_t n_prompt, int32_t n_predict, int32_t mode,
                              int32_t *out_ids, double *seconds_out);

/* ---- text <-> ids (host-only, no GPU needed) -------------------------------------
 * The reference's tokenizer stack -- moses_tokenize (mosestokenizer.cpp:290-358), bpe (bpe.cpp:20-91),
 * gpt_tokenize / gpt_decode (biogpt.cpp:850-906), moses_detokenize (mosestokenizer.cpp:360-466) --
 * re-implemented without std::regex, byte-for-byte equal in output including its quirks.
 *
 * Vocabulary handles: biogpt_hip_vocab_load() parses only the header, vocab and merges of a model file;
 * biogpt_hip_ctx_vocab() borrows the one a loaded context already holds (NULL for an attached context).
 * String results: the function returns the byte length of the result (without the terminating NUL) and
 * writes it only when cap >= length + 1 -- call again with a larger buffer otherwise.  Lists of words
 * travel as one '\n'-joined string.  Negative = error; BIOGPT_HIP_E_LENGTH where the reference throws
 * std::length_error (a period-final word that is neither an abbreviation nor a listed prefix, followed by
 * a word starting with a byte >= 0x80; mosestokenizer.cpp:264).
 *
 * Data files: nonbreaking_prefixes/nonbreaking_prefix.<lang> are read from the data directory
 * ($BIOGPT_DATA_DIR, else "../data" like mosestokenizer.cpp:11-12; a missing file = empty list, as in
 * the reference).  The five perluniprops byte classes are built in; <dir>/perluniprops/Is*.txt override
 * them when present.  lang: "" (what the reference CLI effectively passes), "en", "fr", ... */
biogpt_hip_vocab       *biogpt_hip_vocab_load(const char *fname);
/* from arrays: tokens[id] / merges[rank] ("left right" records as stored in the file), explicit byte lengths */
biogpt_hip_vocab       *biogpt_hip_vocab_create(const char *const *tokens, const int32_t *token_lens, int32_t n_tokens,
                                                const char *const *merges, const int32_t *merge_lens, int32_t n_merges);
void                    biogpt_hip_vocab_free(biogpt_hip_vocab *v);
const biogpt_hip_vocab *biogpt_hip_ctx_vocab(const biogpt_hip_ctx *ctx);
int biogpt_hip_tokenizer_set_data_dir(const char *dir);

/* gpt_tokenize: ids of `text`, always starting with 2 ("</s>"); pieces missing from the vocabulary are
 * dropped with a warning on stderr.  Returns the id count; writes min(count, cap) ids. */
int biogpt_hip_tokenize(const biogpt_hip_vocab *v, const char *text, const char *lang, int32_t *out_ids, int32_t cap);
/* gpt_decode of the vocabulary strings of `ids` (an id outside the table decodes as empty). */
int biogpt_hip_decode(const biogpt_hip_vocab *v, const int32_t *ids, int32_t n, const char *lang, char *out, int32_t cap);
/* gpt_decode on explicit token strings ('\n'-joined) */
int biogpt_hip_decode_strings(const char *tokens_nl, const char *lang, char *out, int32_t cap);

/* the stages, exposed for parity tests */
int biogpt_hip_moses_tokenize(const char *text, const char *lang, char *out, int32_t cap);        /* words, '\n'-joined */
int biogpt_hip_moses_detokenize(const char *tokens_nl, const char *lang, char *out, int32_t cap);
int biogpt_hip_bpe(const biogpt_hip_vocab *v, const char *word, char *out, int32_t cap);          /* pieces joined by ' ' */
/* membership table (256 x 0/1) of byte class `which`: 0 IsAlnum, 1 IsAlpha, 2 IsLower, 3 IsN, 4 IsSc */
int biogpt_hip_tokenizer_byte_class(int which, uint8_t *out256);

/* ---- host-side tools (no GPU needed) --------------------------------------------------------
 * File -> file quantizer, replaces examples/quantize (quantize.cpp:8-135 + biogpt.cpp:459-621):
 * ftype in {2,3,7,8,9}; every 2-D tensor whose name contains "weight" is quantized in rows of
 * ne[0]; header ftype rewritten; vocab/merges copied verbatim. */
int biogpt_hip_quantize_file(const char *fname_in, const char *fname_out, int32_t ftype);

/* The quantizer's inner loops on the device (biogpt.cpp:565-603 -> ggml_quantize_q4_0 .. q8_0): nrows rows of k
 * f32 values in host memory -> the FILE's block format of ggml type `type` (2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0) in host
 * memory, byte-identical to biogpt_hip_quantize_file's host encoder. */
int biogpt_hip_quantize_rows_device(int device, int32_t type, const float *src, int64_t nrows, int64_t k, uint8_t *dst);

/* Write a synthetic seeded BioGPT model: F32 (ftype 0) or F16 (ftype 1) file in the
 * reference's format, weights ~ N(0, 0.02^2), LayerNorm gains 1 + N(0, 0.02^2), biases N(0, 0.02^2),
 * embed_tokens row 1 zero; n_merges merge records are written (40000 keeps the reference's
 * loader happy, F6). */
int biogpt_hip_write_synthetic(const char *fname, const biogpt_hip_hparams *hp, uint64_t seed);

/* Measurement: the decode mat-vec over EVERY block-quantized matrix of the loaded model (24 x {q/k/v, out_proj, fc1, fc2} + lm_head = the W of SURVEY 8(d)) in one
 * launch, rows spread over the chip, each against a resident Q8 activation vector of its shape; launches alternate between two copies of the weights so that the
 * 256 MB Infinity Cache cannot serve them.  seconds_out per launch, bytes_out = SURVEY 8(d)'s algorithmic bytes per launch, check_out = max |device - host| over
 * sampled rows (0 = bit-identical to the reference's arithmetic; -1 = format not checked).  The figure north_star's ">= 70 % of the HBM roofline on the Q4_0
 * single-token decode mat-vec at d_model = 1024" asks for; replaces nothing of biogpt.cpp (its loop body is biogpt.cpp:705-803). */
int biogpt_hip_bench_sweep(biogpt_hip_ctx *ctx, int which /* 0: every matrix, two copies; 1: lm_head alone; 2: q/k/v + out_proj of every layer (the K = 1024, D x D shapes); 3: fc2 (K = 4096); 4: fc1 -- 1 .. 4 on as many copies in turn as exceed the Infinity Cache */, int reps, double *seconds_out, double *bytes_out, double *check_out);
/* the same, and (each may be NULL) the launch's output rows (matrix after matrix: per layer q/k/v, out_proj, fc1, fc2, then lm_head -- those `which` selects) and the two Q8
 * activation vectors (xq_out: 1024 + 4096 int8, xd_out: 32 + 128 block scales) they were computed with, so that a test can recompute rows with the oracle's vec_dot of
 * biogpt.cpp:705-716,767,803's mat-vecs */
int biogpt_hip_bench_sweep_ex(biogpt_hip_ctx *ctx, int which, int reps, double *seconds_out, double *bytes_out, double *check_out, float *rows_out, size_t rows_cap, int8_t *xq_out, float *xd_out);

/* Single-token evals of a context that does not hold the device's pipeline slot are replayed as a captured five-launch step.  The row of such a replay carries the
 * sequence number its first node fetched (forwarded by the last layer's last kernel and by the lm_head as they start); biogpt_hip_eval / biogpt_hip_eval_inplace /
 * biogpt_hip_read_logits compare it with the call's and, if it is another call's, repeat the call on eager launches.  out2 = {evals replayed that way, rows repeated}.
 * The contract behind it: the row biogpt_eval returns is the row of THIS eval (biogpt.cpp:840-844). */
int biogpt_hip_lineage_stats(biogpt_hip_ctx *ctx, int64_t *out2);

#ifdef __cplusplus
}
#endif
#endif /* BIOGPT_HIP_H */
