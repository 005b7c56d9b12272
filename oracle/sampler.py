"""CPU restatement of the reference's sampler, biogpt_sample_top_k_top_p (biogpt.cpp:908-980).

TEST INFRASTRUCTURE ONLY (see biogpt_oracle.h): imported by tests/ as the checker of the product's host sampler
(csrc/compat.cpp), never by the product.

The reference draws from std::mt19937 through std::discrete_distribution<>.  Both are restated here from their
specifications rather than called: the Mersenne Twister is numpy's own implementation with the classic
init_genrand seeding (= std::mt19937(seed), ISO C++ [rand.eng.mers]); discrete_distribution follows libstdc++
(bits/random.tcc): probabilities normalised by their in-order sum, partial sums with the last forced to 1.0, ONE
generate_canonical<double, 53> draw (two 32-bit outputs: (a + b * 2^32) / 2^64) and lower_bound over the partial
sums; fewer than two probabilities -> index 0 WITHOUT consuming a draw.
"""
import math

import numpy as np


class Mt19937:
    """std::mt19937(seed): 32-bit outputs in the standard's order."""

    def __init__(self, seed):
        self._bg = np.random.MT19937()
        self._bg._legacy_seeding(int(seed) & 0xFFFFFFFF)

    def __call__(self):
        return int(self._bg.random_raw())


def generate_canonical_53(rng):
    a = float(rng())
    b = float(rng())
    r = (a + b * 4294967296.0) / 18446744073709551616.0
    return math.nextafter(1.0, 0.0) if r >= 1.0 else r


def discrete_distribution(probs, rng):
    if len(probs) < 2:
        return 0
    total = 0.0
    for p in probs:
        total += p
    cp, run = [], 0.0
    for p in probs:
        run += p / total
        cp.append(run)
    cp[-1] = 1.0
    u = generate_canonical_53(rng)
    lo, hi = 0, len(cp)                      # std::lower_bound: first partial sum >= u
    while lo < hi:
        mid = (lo + hi) // 2
        if cp[mid] < u:
            lo = mid + 1
        else:
            hi = mid
    return lo


def sample_top_k_top_p(logits, top_k, top_p, temp, rng):
    """biogpt.cpp:908-980.  logits: float32 array; top_p / temp: the doubles the caller passes (the CLI's are floats,
    biogpt.h:115-116, widened).  Returns the sampled id."""
    logits = np.asarray(logits, dtype=np.float32)
    scale = 1.0 / float(temp)
    scaled = logits.astype(np.float64) * scale                      # logits[i]*scale in double (:924)
    # std::partial_sort by descending score (:929-934); ties keep the lower id first (unspecified in the reference)
    order = np.argsort(-scaled, kind="stable")[:top_k]
    cand = [(float(scaled[i]), int(i)) for i in order]
    maxl = -math.inf
    for v, _ in cand:
        maxl = max(maxl, v)
    probs, total = [], 0.0
    for v, _ in cand:
        p = math.exp(v - maxl)
        probs.append(p)
        total += p
    probs = [p / total for p in probs]
    if top_p < 1.0:
        cumsum = 0.0
        for i in range(top_k):
            cumsum += probs[i]
            if cumsum >= top_p:
                probs = probs[:i + 1]
                cand = cand[:i + 1]
                break
        inv = 1.0 / cumsum
        probs = [p * inv for p in probs]
    return cand[discrete_distribution(probs, rng)][1]
