// Host-side shared definitions for the BioGPT MI355X engine: tensor type ids of the
// ggml-model.bin format, fp16 conversion, error plumbing.
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/biogpt_hip.h"

namespace bg {

// ggml_type ids as stored per tensor in the file (biogpt.cpp:412, :551)
enum TensorType : int32_t {
    T_F32  = 0,
    T_F16  = 1,
    T_Q4_0 = 2,
    T_Q4_1 = 3,
    T_Q5_0 = 6,
    T_Q5_1 = 7,
    T_Q8_0 = 8,
    T_INVALID = -1,
};

constexpr int QK = 32;  // elements per quantization block
constexpr uint32_t FILE_MAGIC = 0x67676d6c;  // 'ggml' (biogpt.h:13, convert.py:90)

// header ftype (ggml_ftype) -> tensor type of the 2-D weight matrices (biogpt.cpp:160)
inline TensorType ftype_to_type(int32_t ftype) {
    switch (ftype) {
        case 0: return T_F32;
        case 1: return T_F16;
        case 2: return T_Q4_0;
        case 3: return T_Q4_1;
        case 7: return T_Q8_0;
        case 8: return T_Q5_0;
        case 9: return T_Q5_1;
        default: return T_INVALID;
    }
}

inline const char *type_name(int32_t t) {
    switch (t) {
        case T_F32: return "f32";
        case T_F16: return "f16";
        case T_Q4_0: return "q4_0";
        case T_Q4_1: return "q4_1";
        case T_Q5_0: return "q5_0";
        case T_Q5_1: return "q5_1";
        case T_Q8_0: return "q8_0";
        default: return "?";
    }
}

// bytes of one 32-element block in the FILE layout (0 for the float types)
inline size_t file_block_bytes(int32_t t) {
    switch (t) {
        case T_Q4_0: return 18;
        case T_Q4_1: return 20;
        case T_Q5_0: return 22;
        case T_Q5_1: return 24;
        case T_Q8_0: return 34;
        default: return 0;
    }
}

inline bool is_quantized(int32_t t) { return file_block_bytes(t) != 0; }

inline size_t file_row_bytes(int32_t t, int64_t k) {
    if (t == T_F32) return (size_t)k * 4;
    if (t == T_F16) return (size_t)k * 2;
    return (size_t)(k / QK) * file_block_bytes(t);
}

// ---- fp16 <-> fp32 (IEEE binary16, round-to-nearest-even, subnormals kept) -----------------
inline uint32_t f32_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bits_f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

inline uint16_t f32_to_f16(float f) {
    uint32_t u = f32_bits(f);
    const uint16_t s = (uint16_t)((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (uint16_t)(s | 0x7e00u);   // NaN
    if (u >= 0x477ff000u) return (uint16_t)(s | 0x7c00u);  // >= 65520 rounds to inf
    if (u < 0x38800000u) {
        // result is subnormal (or zero): adding 0.5 makes the FPU round at 2^-24 granularity
        const float a = bits_f32(u) + 0.5f;
        return (uint16_t)(s | (uint16_t)(f32_bits(a) - 0x3f000000u));
    }
    const uint32_t r = u + 0x0fffu + ((u >> 13) & 1u);     // round mantissa to 10 bits, ties to even
    return (uint16_t)(s | (uint16_t)((r - 0x38000000u) >> 13));
}

inline float f16_to_f32(uint16_t h) {
    const uint32_t s = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t em = h & 0x7fffu;
    if (em >= 0x7c00u) return bits_f32(s | 0x7f800000u | ((em & 0x3ffu) << 13));
    if (em < 0x0400u) return bits_f32(s | f32_bits((float)em * 5.9604644775390625e-8f));  // em * 2^-24
    return bits_f32(s | ((em << 13) + 0x38000000u));
}

// ---- error plumbing ---------------------------------------------------------------------------
// Mirrors the reference's "fprintf(stderr, "%s: ...", __func__) ; return false" convention and
// keeps the message for biogpt_hip_last_error().
void set_error(const char *func, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *last_error();
void clear_error();

// tokenizer_capi.cpp: the vocabulary handle a context carries for biogpt_hip_ctx_vocab()
biogpt_hip_vocab *make_vocab(const std::vector<std::string> &tokens, const std::vector<std::string> &merges);
void drop_vocab(biogpt_hip_vocab *v);

#define BG_FAIL(ret, ...)                     \
    do {                                      \
        ::bg::set_error(__func__, __VA_ARGS__); \
        return ret;                           \
    } while (0)

}  // namespace bg
