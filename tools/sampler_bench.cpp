#include "biogpt_compat.h"
#include <chrono>
#include <cstdio>
#include <random>
#include <algorithm>
typedef std::pair<double, int> scored;
int main() {
    biogpt_vocab v;
    for (int i = 0; i < 42384; i++) v.id_to_token[i] = "t";
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> lg(42384);
    for (auto &x : lg) x = nd(g);
    lg[77] = lg[4000]; lg[100] = lg[200] = 9.0f;      // ties
    // reference way, for the same rng state
    int mism = 0;
    for (int rep = 0; rep < 200; rep++) {
        for (auto &x : lg) x = nd(g);
        if (rep % 3 == 0) { lg[5] = lg[4001] = lg[30000] = 3.5f; }
        std::mt19937 r1(rep), r2(rep);
        const int a = biogpt_sample_top_k_top_p(v, lg.data(), 40, 0.9, 0.9, r1);
        std::vector<scored> cand(lg.size());
        for (size_t i = 0; i < lg.size(); i++) cand[i] = scored(lg[i] * (1.0 / 0.9), (int)i);
        std::stable_sort(cand.begin(), cand.end(), [](const scored &x, const scored &y) { return x.first > y.first; });
        cand.resize(40);
        // softmax / top-p / draw as the library does it: reuse through top_k = 40 on a vector that only holds the 40 (others -inf)
        std::vector<float> only(lg.size(), -INFINITY);
        for (auto &c : cand) only[(size_t)c.second] = lg[(size_t)c.second];
        const int b = biogpt_sample_top_k_top_p(v, only.data(), 40, 0.9, 0.9, r2);
        mism += a != b;
    }
    printf("mismatches against a stable full sort: %d of 200\n", mism);
    std::mt19937 r(7);
    const auto t0 = std::chrono::steady_clock::now();
    long acc = 0;
    for (int rep = 0; rep < 2000; rep++) { lg[rep % 42384] += 0.001f; acc += biogpt_sample_top_k_top_p(v, lg.data(), 40, 0.9, 0.9, r); }
    printf("biogpt_sample_top_k_top_p(top_k 40): %.1f us per call (%ld)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 2000 * 1e6, acc);
    return 0;
}
