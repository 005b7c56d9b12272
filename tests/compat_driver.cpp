// Test driver for include/biogpt_compat.h: the generation loop of the reference's CLI
// (examples/main/main.cpp:36-151) written against the reference's C++ API names, with token ids as
// input/output instead of text (the tokenizer is out of scope).
//   usage: compat_driver MODEL N_PREDICT TOP_K ID [ID...]   -> prints the sampled ids, one line
#include <cstdio>
#include <cstdlib>

#include "biogpt.h"
#include "ggml.h"
#include "ggml-alloc.h"

int main(int argc, char **argv) {
    ggml_time_init();
    biogpt_params params;
    token_sequence embed_inp;
    if (argc >= 2 && std::string(argv[1]) == "--flags") {
        // reference-style flags (biogpt_params_parse); the prompt is a list of ids "2 17 45"
        if (!biogpt_params_parse(argc - 1, argv + 1, params)) return 2;
        size_t pos = 0;
        while (pos < params.prompt.size()) {
            size_t used = 0;
            embed_inp.push_back(std::stoi(params.prompt.substr(pos), &used));
            pos += used;
            while (pos < params.prompt.size() && params.prompt[pos] == ' ') pos++;
        }
    } else {
        if (argc < 5) return 2;
        params.model = argv[1];
        params.n_predict = std::atoi(argv[2]);
        params.top_k = std::atoi(argv[3]);
        for (int i = 4; i < argc; i++) embed_inp.push_back(std::atoi(argv[i]));
    }
    std::mt19937 rng(7);

    biogpt_vocab vocab;
    biogpt_model model;
    if (!biogpt_model_load(params.model, model, vocab, params.verbosity)) {
        fprintf(stderr, "failed to load model from '%s'\n", params.model.c_str());
        return 1;
    }
    // compute-buffer measurement dance of main.cpp:47-70 (all no-ops on this engine)
    struct ggml_allocr *allocr = ggml_allocr_new_measure(ggml_backend_get_alignment(model.backend));
    int n_tokens = std::min(model.hparams.n_positions, params.n_batch);
    struct ggml_cgraph *gf = biogpt_graph(model, allocr, token_sequence(n_tokens, 0), model.hparams.n_positions - n_tokens);
    size_t mem_size = ggml_allocr_alloc_graph(allocr, gf);
    ggml_allocr_free(allocr);
    ggml_backend_buffer_t buf_compute = ggml_backend_alloc_buffer(model.backend, mem_size);
    allocr = ggml_allocr_new_from_buffer(buf_compute);

    params.n_predict = std::min(params.n_predict, model.hparams.n_positions - (int)embed_inp.size());
    int n_past = 0;
    std::vector<float> logits;
    token_sequence embed;
    for (size_t i = embed.size(); i < embed_inp.size() + params.n_predict; i++) {
        if (!embed.empty() && !biogpt_eval(model, embed, logits, allocr, n_past, params.n_threads)) return 1;
        n_past += (int)embed.size();
        embed.clear();
        if (i >= embed_inp.size()) {
            const int n_vocab = model.hparams.n_vocab;
            biogpt_vocab::id id = biogpt_sample_top_k_top_p(vocab, logits.data() + (logits.size() - n_vocab), params.top_k,
                                                            params.top_p, params.temp, rng);
            embed.push_back(id);
            printf("%d ", id);
        } else {
            for (size_t k = i; k < embed_inp.size(); k++) {
                embed.push_back(embed_inp[k]);
                if ((int32_t)embed.size() >= params.n_batch) break;
            }
            i += embed.size() - 1;
        }
    }
    printf("\n");
    fprintf(stderr, "vocab %zu tokens, %zu merges, n_loaded %d, %lld us\n", vocab.id_to_token.size(), vocab.bpe_ranks.size(),
            model.n_loaded, (long long)ggml_time_us());
    ggml_free(model.ctx);
    ggml_backend_buffer_free(model.buffer_w);
    ggml_backend_buffer_free(model.buffer_kv);
    ggml_backend_buffer_free(buf_compute);
    ggml_backend_free(model.backend);
    return 0;
}
