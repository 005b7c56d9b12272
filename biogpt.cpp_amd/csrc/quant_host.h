// Host-side block quantizers for the ggml-model.bin weight formats and the two host tools built on
// them: the file->file quantizer (replaces examples/quantize: quantize.cpp:8-135 +
// biogpt.cpp:459-621) and the synthetic model writer used by bench/tests.
// Block formats and rounding rules: SURVEY.md Appendix A.1 / A.2.
#pragma once

#include "host_common.h"

namespace bg {

// Quantize `nrows` rows of `k` floats each into the FILE block layout of `type`; returns bytes written.
size_t quantize_rows(int32_t type, const float *src, int64_t nrows, int64_t k, uint8_t *dst);

// Dequantize one row (file layout) back to floats.
void dequantize_row(int32_t type, const uint8_t *src, int64_t k, float *dst);

bool quantize_file(const std::string &in_path, const std::string &out_path, int32_t ftype);

bool write_synthetic(const std::string &path, const biogpt_hip_hparams &hp, uint64_t seed);

}  // namespace bg
