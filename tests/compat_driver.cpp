// Test driver for include/biogpt_compat.h.  It exercises every model-library symbol the reference's CLI
// uses (biogpt.h:128-169 + the ggml handles of examples/main/main.cpp:47-70,164-169), with the prompt phase
// and the sampling phase as two separate loops.
//
//   compat_driver MODEL N_PREDICT TOP_K ID [ID...]          positional form, token ids in
//   compat_driver --flags <reference-style flags>           -p "2 17 45" is a list of ids
//   compat_driver --text <reference-style flags>            -p "some text": gpt_tokenize in, gpt_decode out
// prints the sampled ids on one line; --text also prints "prompt ids: ..." before and "text: ..." after.
#include <cstdio>
#include <cstdlib>
#include <sstream>

#include "biogpt.h"
#include "ggml.h"
#include "ggml-alloc.h"

namespace {

struct Session {
    biogpt_model model;
    biogpt_vocab vocab;
    ggml_allocr *allocr = nullptr;
    ggml_backend_buffer_t scratch = nullptr;
    std::vector<float> logits;
    int n_past = 0;

    bool open(const biogpt_params &prm) {
        if (!biogpt_model_load(prm.model, model, vocab, prm.verbosity)) return false;
        // what main.cpp does to size its compute buffer; every call is a shim here
        ggml_allocr *probe = ggml_allocr_new_measure(ggml_backend_get_alignment(model.backend));
        const int worst = std::min(model.hparams.n_positions, prm.n_batch);
        ggml_cgraph *g = biogpt_graph(model, probe, token_sequence((size_t)worst, 0), model.hparams.n_positions - worst);
        const size_t bytes = ggml_allocr_alloc_graph(probe, g);
        ggml_allocr_free(probe);
        scratch = ggml_backend_alloc_buffer(model.backend, bytes);
        allocr = ggml_allocr_new_from_buffer(scratch);
        return true;
    }
    bool feed(const token_sequence &chunk, int n_threads) {
        if (!biogpt_eval(model, chunk, logits, allocr, n_past, n_threads)) return false;
        n_past += (int)chunk.size();
        return true;
    }
    void close() {
        ggml_free(model.ctx);
        ggml_backend_buffer_free(model.buffer_w);
        ggml_backend_buffer_free(model.buffer_kv);
        ggml_backend_buffer_free(scratch);
        ggml_backend_free(model.backend);
    }
};

token_sequence parse_ids(const std::string &text) {
    token_sequence ids;
    std::istringstream in(text);
    int v;
    while (in >> v) ids.push_back(v);
    return ids;
}

}  // namespace

int main(int argc, char **argv) {
    ggml_time_init();
    biogpt_params prm;
    token_sequence prompt;
    const bool text_mode = argc >= 2 && std::string(argv[1]) == "--text";
    const bool fused_sampler = argc >= 2 && std::string(argv[1]) == "--flags-device-topk";   // biogpt_eval_sample_top_k_top_p instead of eval + sample
    if (argc >= 2 && (text_mode || fused_sampler || std::string(argv[1]) == "--flags")) {
        if (!biogpt_params_parse(argc - 1, argv + 1, prm)) return 2;
        if (!text_mode) prompt = parse_ids(prm.prompt);
    } else {
        if (argc < 5) return 2;
        prm.model = argv[1];
        prm.n_predict = std::atoi(argv[2]);
        prm.top_k = std::atoi(argv[3]);
        for (int a = 4; a < argc; a++) prompt.push_back(std::atoi(argv[a]));
    }

    if (argc >= 7 && std::string(argv[1]) == "--sample-only") {
        // host-only check of biogpt_sample_top_k_top_p: argv = --sample-only FILE rows n_vocab top_k top_p temp [seed];
        // FILE holds rows x n_vocab float32 logits; one id per row from ONE mt19937 stream
        FILE *f = fopen(argv[2], "rb");
        if (!f) return 1;
        const int rows = std::atoi(argv[3]), nv = std::atoi(argv[4]);
        biogpt_vocab vocab;
        for (int i = 0; i < nv; i++) vocab.id_to_token[i] = "t";
        std::vector<float> lg((size_t)nv);
        std::mt19937 rng(argc >= 9 ? (unsigned)std::atoi(argv[8]) : 7u);
        for (int r = 0; r < rows; r++) {
            if (fread(lg.data(), 4, (size_t)nv, f) != (size_t)nv) return 1;
            printf("%d ", biogpt_sample_top_k_top_p(vocab, lg.data(), std::atoi(argv[5]), std::atof(argv[6]), std::atof(argv[7]), rng));
        }
        printf("\n");
        fclose(f);
        return 0;
    }
    Session s;
    if (!s.open(prm)) {
        fprintf(stderr, "failed to load model from '%s'\n", prm.model.c_str());
        return 1;
    }
    if (text_mode) {
        prompt = gpt_tokenize(s.vocab, prm.prompt, prm.lang);
        printf("prompt ids:");
        for (size_t i = 0; i < prompt.size(); i++) printf(" %d", prompt[i]);
        printf("\n");
    }
    const int budget = std::min(prm.n_predict, s.model.hparams.n_positions - (int)prompt.size());
    token_sequence sampled;

    // phase 1: the prompt, n_batch ids per eval
    for (size_t at = 0; at < prompt.size(); at += (size_t)prm.n_batch) {
        const size_t n = std::min((size_t)prm.n_batch, prompt.size() - at);
        if (!s.feed(token_sequence(prompt.begin() + at, prompt.begin() + at + n), prm.n_threads)) return 1;
    }
    // phase 2: sample, print, feed back (the last sampled id is not evaluated)
    std::mt19937 rng(7);
    for (int k = 0; k < budget; k++) {
        biogpt_vocab::id id;
        if (fused_sampler && k > 0) {   // evaluate the previous id and sample in one call: only the top-k logits leave the device
            id = biogpt_eval_sample_top_k_top_p(s.model, s.vocab, token_sequence(1, sampled.back()), s.n_past, prm.top_k, prm.top_p, prm.temp, rng);
            s.n_past += 1;
        } else {
            id = biogpt_sample_top_k_top_p(s.vocab, s.logits.data(), prm.top_k, prm.top_p, prm.temp, rng);
        }
        printf("%d ", id);
        sampled.push_back(id);
        if (!fused_sampler && k + 1 < budget && !s.feed(token_sequence(1, id), prm.n_threads)) return 1;
    }
    printf("\n");
    if (text_mode) {
        std::vector<std::string> words;
        for (size_t i = 0; i < prompt.size(); i++) words.push_back(s.vocab.id_to_token[prompt[i]]);
        for (size_t i = 0; i < sampled.size(); i++) words.push_back(s.vocab.id_to_token[sampled[i]]);
        printf("text: %s\n", gpt_decode(words, prm.lang).c_str());
    }
    fprintf(stderr, "vocab %zu tokens, %zu merges, n_loaded %d, %lld us\n", s.vocab.id_to_token.size(), s.vocab.bpe_ranks.size(),
            s.model.n_loaded, (long long)ggml_time_us());
    s.close();
    return 0;
}
