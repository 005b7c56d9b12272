/*
 * oracle/biogpt_oracle.h -- CPU restatement of the reference's BioGPT forward pass.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (biogpt.cpp_amd/, include/) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" for the ggml arithmetic.  The reference (PABannier/biogpt.cpp)
 * delegates all arithmetic to ggml, an un-vendored git submodule that is absent from
 * /root/reference, and the reference has no forward-pass tests.
 * What IS pinned (tests/test_oracle_*.py):
 *   - the ggml-model.bin byte format, against files written by the reference's own convert.py;
 *   - model semantics (embedding scale, position offset +2, head split, Q scaling after bias,
 *     residual wiring, LayerNorm placement, lm_head) against HuggingFace BioGptForCausalLM fp32
 *     logits on a tiny seeded model (BO_MODE_HF switches below, match ~1e-5);
 *   - block codecs against hand-computed known-answer blocks.
 * The ggml-specific numerics (W*A8 integer block dots, fp16 GELU/exp tables, double LayerNorm
 * statistics) follow SURVEY.md Appendix A, which is recall of ggerganov/ggml (Oct-Nov 2023).
 */
#ifndef BIOGPT_ORACLE_H
#define BIOGPT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml_type ids used by the file format */
enum {
    BO_TYPE_F32  = 0,
    BO_TYPE_F16  = 1,
    BO_TYPE_Q4_0 = 2,
    BO_TYPE_Q4_1 = 3,
    BO_TYPE_Q5_0 = 6,
    BO_TYPE_Q5_1 = 7,
    BO_TYPE_Q8_0 = 8,
    BO_TYPE_Q8_1 = 9,
};

/* numerics switches; the defaults (all zero) are the ggml CPU path the reference runs on */
typedef struct bo_opts {
    int   gelu_erf;   /* 0: tanh-approx GELU through the fp16 table (ggml); 1: exact erf GELU (HF) */
    int   exp_f32;    /* 0: softmax exp through the fp16 table (ggml); 1: expf in f32 (HF)          */
    int   causal;     /* 0: no intra-chunk mask (reference); 1: causal mask (HF)         */
    float ln_eps;     /* 0 -> 1e-5 (biogpt.cpp:24); HF uses 1e-12                                    */
    int   n_threads;  /* 0 -> 1; rows of each mul_mat are split across threads (deterministic)      */
    int   assoc;      /* bit mask; how ggml's SIMD kernels differ from its scalar fallbacks [ggml-recall, SURVEY A.2/A.3]:
                       *   0      scalar fallbacks: one running f32 sum in block order; activations rounded half away from
                       *          zero (roundf) with id = 1/d  -- the parity mode the HIP kernels are compared with
                       *   bit 0  dots in the shape of the AVX2 kernels: 8 lane accumulators updated with fma per block and
                       *          a horizontal add at the end; F32 dots (QK^T, PV) as 4 x 8-lane fma accumulators
                       *   bit 1  activations quantized as the AVX2 kernels do: id = 127/amax, round to nearest-even
                       *   3      both = what an AVX2 build of the reference computes
                       *   bit 2  (with bits 0 / 1) the same dots / conversions executed with AVX2 + FMA intrinsics where the CPU has them
                       *          (bo_have_avx2()): identical values, the speed of SIMD code -- 7 is the CPU baseline bench.py times
                       * All are legal outcomes of "the reference's CPU path" (ggml is not bit-reproducible across ISAs);
                       * tests/test_oracle_assoc.py measures the envelope between them. */
} bo_opts;

typedef struct bo_model bo_model;

/* ---- scalar helpers ---- */
uint16_t bo_fp32_to_fp16(float f);
float    bo_fp16_to_fp32(uint16_t h);
float    bo_gelu_table(float x);   /* fp16-table tanh GELU  */
float    bo_exp_table(float x);    /* fp16-table exp        */

/* ---- block codecs ---- */
size_t bo_type_block_bytes(int type);          /* bytes per 32-element block (4/2 per element for F32/F16 -> returns 0 for those) */
size_t bo_row_bytes(int type, int64_t k);      /* bytes of a row of k elements */
/* quantize n elements laid out as rows of k; returns bytes written */
size_t bo_quantize(int type, const float *src, void *dst, int64_t n, int64_t k);
void   bo_dequantize_row(int type, const void *src, float *dst, int64_t k);
/* dot(W row, x) the way ggml's CPU mul_mat does it: x is converted to the type's vec_dot_type first */
float  bo_vec_dot(int wtype, int64_t k, const void *wrow, const float *x);
float  bo_vec_dot_q(int wtype, int64_t k, const void *wrow, const void *yblocks);   /* activation row given as Q8_0 / Q8_1 blocks */
int    bo_have_avx2(void);                     /* 1: this CPU runs the AVX2 + FMA forms of the SIMD-shaped dots (bo_opts.assoc bit 2) */

/* ---- model file ---- */
bo_model *bo_load(const char *path, char *err, size_t errlen);
void      bo_free(bo_model *m);
/* out[8] = n_vocab, n_layer, n_head, n_positions, d_ff, d_model, ftype, n_merges(from file) */
void      bo_hparams(const bo_model *m, int32_t out[8]);
void      bo_set_opts(bo_model *m, const bo_opts *o);
int       bo_n_tensors(const bo_model *m);

/* one forward pass over n tokens at offset n_past (biogpt.cpp:624-847).
 * logits_last: [n_vocab] row of the last token (what biogpt_eval returns), may be NULL
 * logits_all : [n][n_vocab] all rows, may be NULL
 * returns 0 on success */
int bo_eval(bo_model *m, const int32_t *tokens, int n, int n_past, float *logits_last, float *logits_all);

/* debugging taps: copy of the hidden state after layer `layer` (or -1: after embedding,
 * n_layer: after the final LayerNorm) for the most recent eval; out is [n][d_model] */
int bo_tap(const bo_model *m, int layer, float *out);
/* raw view of the F32 KV cache: which = 0 (K) / 1 (V); returns pointer to [n_layer][n_positions][d_model] */
const float *bo_kv(const bo_model *m, int which);

/* greedy generation harness (examples/main/main.cpp:91-151 with --top_k 1): feeds the prompt in
 * chunks of n_batch, then n_predict single-token evals; writes the n_predict sampled ids to out_ids;
 * returns total seconds spent inside eval (main.cpp:96-103), or <0 on error */
double bo_generate_greedy(bo_model *m, const int32_t *prompt, int n_prompt, int n_batch, int n_predict, int32_t *out_ids);

/* file -> file quantizer (biogpt.cpp:459-621 + quantize.cpp:8-135); ftype in {2,3,7,8,9} */
int bo_quantize_file(const char *in_path, const char *out_path, int ftype, char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
