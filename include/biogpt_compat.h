/*
 * include/biogpt_compat.h -- C++ source-level drop-in for the reference's model library.
 *
 * A program written against the reference's biogpt.h (examples/main/main.cpp is the only caller,
 * SURVEY.md 8b) compiles against this header instead and runs its forward passes on the MI355X
 * engine: the three model-library functions below keep the reference's exact signatures and are
 * thin wrappers over the C-ABI in biogpt_hip.h.
 *
 *   reference declaration                      here
 *   ----------------------------------------   ----------------------------------------------------
 *   biogpt_model_load      biogpt.h:128-132    -> biogpt_hip_load(); fills hparams + vocab maps
 *   biogpt_graph           biogpt.h:139-143    -> returns an opaque placeholder: the engine owns its
 *                                                 launch sequence, there is no compute buffer to size
 *   biogpt_eval            biogpt.h:145-151    -> biogpt_hip_eval(); n_threads / allocr are ignored
 *   biogpt_sample_top_k_top_p  biogpt.h:163-169 -> host sampler, same algorithm + RNG draws
 *   the 11 ggml symbols main.cpp touches (main.cpp:12,14,53,54,62,65,66,67,164,166-169)
 *                                              -> shims below; ggml_free(model.ctx) releases the engine
 *
 * Types keep the reference's member names (biogpt.h:25-107) because callers read them
 * (main.cpp:57-58,82,115,142,148,164-169); tensor handles exist for source compatibility only and
 * stay null -- weights live in the engine's device arena.
 *   gpt_tokenize / gpt_decode  biogpt.h:153-161 -> biogpt_hip_tokenize() / biogpt_hip_decode_strings()
 *                                              (csrc/tokenizer.cpp; same output bytes, same std::length_error)
 */
#pragma once

#include <cstdint>
#include <fstream>
#include <map>
#include <random>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "biogpt_hip.h"

/* ---- opaque stand-ins for the ggml handle types the reference's structs and main.cpp mention ---- */
struct ggml_context;
struct ggml_tensor;
struct ggml_cgraph;
struct ggml_allocr;
struct ggml_backend;
struct ggml_backend_buffer;
typedef struct ggml_backend *ggml_backend_t;
typedef struct ggml_backend_buffer *ggml_backend_buffer_t;
enum ggml_ftype : int { GGML_FTYPE_ALL_F32 = 0, GGML_FTYPE_MOSTLY_F16 = 1, GGML_FTYPE_MOSTLY_Q4_0 = 2, GGML_FTYPE_MOSTLY_Q4_1 = 3,
                        GGML_FTYPE_MOSTLY_Q8_0 = 7, GGML_FTYPE_MOSTLY_Q5_0 = 8, GGML_FTYPE_MOSTLY_Q5_1 = 9 };

extern "C" {
void    ggml_time_init(void);
int64_t ggml_time_us(void);
size_t  ggml_backend_get_alignment(ggml_backend_t backend);
struct ggml_allocr *ggml_allocr_new_measure(size_t alignment);
size_t  ggml_allocr_alloc_graph(struct ggml_allocr *alloc, struct ggml_cgraph *graph);
void    ggml_allocr_free(struct ggml_allocr *alloc);
ggml_backend_buffer_t ggml_backend_alloc_buffer(ggml_backend_t backend, size_t size);
struct ggml_allocr *ggml_allocr_new_from_buffer(ggml_backend_buffer_t buffer);
void    ggml_free(struct ggml_context *ctx); /* releases the engine context stored in biogpt_model::ctx */
void    ggml_backend_buffer_free(ggml_backend_buffer_t buffer);
void    ggml_backend_free(ggml_backend_t backend);
}

#define BIOGPT_FILE_MAGIC 0x67676d6c /* 'ggml' */

/* raw little-endian field I/O used by examples/quantize/quantize.cpp (biogpt.h:15-23) */
template <typename T> static inline void read_safe(std::ifstream &in, T &field) { in.read(reinterpret_cast<char *>(&field), sizeof(T)); }
template <typename T> static inline void write_safe(std::ofstream &out, T &field) { out.write(reinterpret_cast<const char *>(&field), sizeof(T)); }

typedef std::pair<std::string, std::string> word_pair; /* bpe.h */

struct biogpt_hparams { /* biogpt.h:25-35 */
    int32_t n_vocab = 42384, n_merges = 40000, d_ff = 4096, d_model = 1024;
    int32_t n_layer = 24, n_head = 16, n_positions = 1024;
    int32_t ftype = 0;
};

struct biogpt_vocab { /* biogpt.h:37-48 */
    using id = int32_t;
    using token = std::string;
    int n_vocab = 42384;
    int n_merges = 40000;
    std::map<token, id> token_to_id;
    std::map<id, token> id_to_token;
    std::map<word_pair, int> bpe_ranks;
};

typedef std::vector<biogpt_vocab::id> token_sequence;

struct biogpt_layer_decoder { /* biogpt.h:52-76: handles kept for source compatibility, always null */
    ggml_tensor *q_proj_w = nullptr, *k_proj_w = nullptr, *v_proj_w = nullptr, *o_proj_w = nullptr;
    ggml_tensor *q_proj_b = nullptr, *k_proj_b = nullptr, *v_proj_b = nullptr, *o_proj_b = nullptr;
    ggml_tensor *ln_0_w = nullptr, *ln_1_w = nullptr, *ln_0_b = nullptr, *ln_1_b = nullptr;
    ggml_tensor *fc_0_w = nullptr, *fc_0_b = nullptr, *fc_1_w = nullptr, *fc_1_b = nullptr;
};

struct biogpt_model { /* biogpt.h:78-107 */
    biogpt_hparams hparams;
    ggml_tensor *embed_tokens = nullptr, *embed_pos = nullptr, *ln_w = nullptr, *ln_b = nullptr, *lm_head = nullptr;
    ggml_tensor *memory_k = nullptr, *memory_v = nullptr;
    std::vector<biogpt_layer_decoder> layers_decoder;
    ggml_context *ctx = nullptr; /* carries the biogpt_hip_ctx*; freed by ggml_free() (main.cpp:164) */
    std::map<std::string, ggml_tensor *> tensors;
    int n_loaded = 0;
    ggml_backend_t backend = nullptr;
    ggml_backend_buffer_t buffer_w = nullptr;
    ggml_backend_buffer_t buffer_kv = nullptr;
};

struct biogpt_params { /* biogpt.h:109-126 */
    int32_t seed = -1;
    int32_t n_threads = std::min(4, (int32_t)std::thread::hardware_concurrency());
    int32_t n_predict = 200;
    int32_t top_k = 40;
    float top_p = 0.9f;
    float temp = 0.9f;
    uint8_t verbosity = 0;
    int32_t n_batch = 8;
    std::string model = "../ggml_weights/ggml-model.bin";
    std::string prompt;
    std::string lang;
};

/* the engine context behind a loaded model (for callers that want the C-ABI extras) */
inline biogpt_hip_ctx *biogpt_model_hip(const biogpt_model &model) { return reinterpret_cast<biogpt_hip_ctx *>(model.ctx); }

bool biogpt_model_load(const std::string &fname, biogpt_model &model, biogpt_vocab &vocab, const uint8_t verbosity);

struct ggml_cgraph *biogpt_graph(const biogpt_model &model, struct ggml_allocr *allocr, const token_sequence &embed_inp,
                                 const int n_past);

bool biogpt_eval(const biogpt_model &model, const token_sequence &embed_inp, std::vector<float> &logits,
                 struct ggml_allocr *allocr, const int n_past, const int n_threads);

/* tensor section of a model file, stream to stream (biogpt.cpp:459-621); whole-file form: biogpt_hip_quantize_file() */
void biogpt_model_quantize_internal(std::ifstream &fin, std::ofstream &fout, const ggml_ftype ftype);

biogpt_vocab::id biogpt_sample_top_k_top_p(const biogpt_vocab &vocab, const float *logits, int top_k, double top_p,
                                           double temp, std::mt19937 &rng);
/* Extension, not in the reference: biogpt_eval (biogpt.h:145-151) + biogpt_sample_top_k_top_p (:163-169) in one call with
 * the top-k selection on the device -- the logits row stays in HBM, 512 bytes cross PCIe.  Replaces the pair at
 * examples/main/main.cpp:98 + :118 for callers that opt in. */
biogpt_vocab::id biogpt_eval_sample_top_k_top_p(const biogpt_model &model, const biogpt_vocab &vocab, const token_sequence &embed_inp,
                                                const int n_past, int top_k, double top_p, double temp, std::mt19937 &rng);

bool biogpt_params_parse(int argc, char **argv, biogpt_params &params);
void biogpt_print_usage(char **argv, const biogpt_params &params);

/* text <-> ids (biogpt.cpp:850-906); prefix lists come from $BIOGPT_DATA_DIR or ../data, like the reference */
token_sequence gpt_tokenize(biogpt_vocab &vocab, const std::string &text, const std::string &lang);
std::string gpt_decode(std::vector<std::string> &tokens, const std::string &lang);
