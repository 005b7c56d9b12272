/*
 * oracle/biogpt_oracle.c -- CPU restatement of the reference's BioGPT forward pass.
 *
 * TEST INFRASTRUCTURE ONLY (see biogpt_oracle.h).  "parity unpinned" for the ggml arithmetic:
 * ggml is absent from /root/reference, so every ggml semantic below follows SURVEY.md Appendix A
 * (recall of ggerganov/ggml, Oct-Nov 2023 window) and is marked [ggml-recall].  The model
 * semantics and the file format are pinned against HuggingFace BioGPT and the reference's
 * convert.py (tests/golden/make_golden.py).
 *
 * Reference lines restated (cited per function):
 *   file format     biogpt.cpp:27-453, convert.py:28-97
 *   forward pass    biogpt.cpp:624-810 (biogpt_graph), :812-847 (biogpt_eval)
 *   quantizer       biogpt.cpp:459-621, examples/quantize/quantize.cpp:8-135
 *   greedy harness  examples/main/main.cpp:91-151, biogpt.cpp:908-980 with top_k = 1
 *
 * Everything is plain scalar C in the order the reference's graph emits its ops; the only
 * parallelism is an OpenMP split over output rows of each mul_mat (does not change any sum order).
 */
#define _POSIX_C_SOURCE 200809L
#include "biogpt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QK 32
#define BO_MAGIC 0x67676d6c /* 'ggml' biogpt.h:13, convert.py:90 */
#define NORM_EPS 1e-5f      /* biogpt.cpp:24 */

/* ------------------------------------------------------------------------------------------
 * fp16 <-> fp32, IEEE round-to-nearest-even, subnormals kept  [ggml-recall: GGML_FP32_TO_FP16]
 * ---------------------------------------------------------------------------------------- */
uint16_t bo_fp32_to_fp16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax   = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
    const int32_t e = (int32_t)(ax >> 23) - 127 + 15;
    uint32_t man    = ax & 0x7fffffu;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift     = 14 - e;
        uint32_t half       = man >> shift;
        const uint32_t rem  = man & ((1u << shift) - 1u);
        const uint32_t mid  = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1u))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t half      = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++; /* carry may roll into inf: correct */
    return (uint16_t)(sign | half);
}

float bo_fp16_to_fp32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e    = (h >> 10) & 0x1fu;
    const uint32_t man  = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (man == 0) {
            x = sign;
        } else { /* subnormal: value = man * 2^-24 */
            float f = (float)man * 5.9604644775390625e-8f;
            memcpy(&x, &f, 4);
            x |= sign;
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (man << 13);
    } else {
        x = sign | ((e - 15 + 127) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* fp16 lookup tables, built once the way ggml_init() builds them [ggml-recall, SURVEY A.5] */
static uint16_t g_table_gelu[1 << 16];
static uint16_t g_table_exp[1 << 16];
static float    g_table_f32[1 << 16];
static int      g_tables_ready = 0;

static inline float gelu_tanh_f32(float x) {
    const float GELU_COEF_A    = 0.044715f;
    const float SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    return 0.5f * x * (1.0f + tanhf(SQRT_2_OVER_PI * x * (1.0f + GELU_COEF_A * x * x)));
}

static void init_tables(void) {
    if (g_tables_ready) return;
#pragma omp critical(bo_tables)
    {
        if (!g_tables_ready) {
            for (uint32_t i = 0; i < (1u << 16); i++) {
                const float f   = bo_fp16_to_fp32((uint16_t)i);
                g_table_f32[i]  = f;
                g_table_gelu[i] = bo_fp32_to_fp16(gelu_tanh_f32(f));
                g_table_exp[i]  = bo_fp32_to_fp16(expf(f));
            }
            g_tables_ready = 1;
        }
    }
}

float bo_gelu_table(float x) {
    init_tables();
    return g_table_f32[g_table_gelu[bo_fp32_to_fp16(x)]];
}
float bo_exp_table(float x) {
    init_tables();
    return g_table_f32[g_table_exp[bo_fp32_to_fp16(x)]];
}
#define F16(h) (g_table_f32[(h)])

/* ------------------------------------------------------------------------------------------
 * block formats -- packed, little-endian
 * ---------------------------------------------------------------------------------------- */
#pragma pack(push, 1)
typedef struct { uint16_t d; uint8_t qs[16]; } blk_q4_0;                            /* 18 B */
typedef struct { uint16_t d; uint16_t m; uint8_t qs[16]; } blk_q4_1;                /* 20 B */
typedef struct { uint16_t d; uint8_t qh[4]; uint8_t qs[16]; } blk_q5_0;             /* 22 B */
typedef struct { uint16_t d; uint16_t m; uint8_t qh[4]; uint8_t qs[16]; } blk_q5_1; /* 24 B */
typedef struct { uint16_t d; int8_t qs[32]; } blk_q8_0;                             /* 34 B */
typedef struct { float d; float s; int8_t qs[32]; } blk_q8_1;                       /* 40 B */
#pragma pack(pop)

size_t bo_type_block_bytes(int type) {
    switch (type) {
        case BO_TYPE_Q4_0: return sizeof(blk_q4_0);
        case BO_TYPE_Q4_1: return sizeof(blk_q4_1);
        case BO_TYPE_Q5_0: return sizeof(blk_q5_0);
        case BO_TYPE_Q5_1: return sizeof(blk_q5_1);
        case BO_TYPE_Q8_0: return sizeof(blk_q8_0);
        case BO_TYPE_Q8_1: return sizeof(blk_q8_1);
        default: return 0;
    }
}

size_t bo_row_bytes(int type, int64_t k) {
    if (type == BO_TYPE_F32) return (size_t)k * 4;
    if (type == BO_TYPE_F16) return (size_t)k * 2;
    return (size_t)(k / QK) * bo_type_block_bytes(type);
}

#define MIN_I(a, b) ((a) < (b) ? (a) : (b))

/* quantize_row_*_reference [ggml-recall, SURVEY A.2] */
static void quantize_row_q4_0(const float *x, blk_q4_0 *y, int64_t k) {
    for (int64_t i = 0; i < k / QK; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = x[i * QK + j];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d  = max / -8;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d         = bo_fp32_to_fp16(d);
        for (int j = 0; j < QK / 2; j++) {
            const float x0    = x[i * QK + 0 + j] * id;
            const float x1    = x[i * QK + QK / 2 + j] * id;
            const uint8_t xi0 = (uint8_t)MIN_I(15, (int8_t)(x0 + 8.5f));
            const uint8_t xi1 = (uint8_t)MIN_I(15, (int8_t)(x1 + 8.5f));
            y[i].qs[j]        = (uint8_t)(xi0 | (xi1 << 4));
        }
    }
}

static void quantize_row_q4_1(const float *x, blk_q4_1 *y, int64_t k) {
    for (int64_t i = 0; i < k / QK; i++) {
        float min = FLT_MAX, max = -FLT_MAX;
        for (int j = 0; j < QK; j++) {
            const float v = x[i * QK + j];
            if (v < min) min = v;
            if (v > max) max = v;
        }
        const float d  = (max - min) / ((1 << 4) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d         = bo_fp32_to_fp16(d);
        y[i].m         = bo_fp32_to_fp16(min);
        for (int j = 0; j < QK / 2; j++) {
            const float x0    = (x[i * QK + 0 + j] - min) * id;
            const float x1    = (x[i * QK + QK / 2 + j] - min) * id;
            const uint8_t xi0 = (uint8_t)MIN_I(15, (int8_t)(x0 + 0.5f));
            const uint8_t xi1 = (uint8_t)MIN_I(15, (int8_t)(x1 + 0.5f));
            y[i].qs[j]        = (uint8_t)(xi0 | (xi1 << 4));
        }
    }
}

static void quantize_row_q5_0(const float *x, blk_q5_0 *y, int64_t k) {
    for (int64_t i = 0; i < k / QK; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = x[i * QK + j];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d  = max / -16;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d         = bo_fp32_to_fp16(d);
        uint32_t qh    = 0;
        for (int j = 0; j < QK / 2; j++) {
            const float x0    = x[i * QK + 0 + j] * id;
            const float x1    = x[i * QK + QK / 2 + j] * id;
            const uint8_t xi0 = (uint8_t)MIN_I(31, (int8_t)(x0 + 16.5f));
            const uint8_t xi1 = (uint8_t)MIN_I(31, (int8_t)(x1 + 16.5f));
            y[i].qs[j]        = (uint8_t)((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
            qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
            qh |= ((xi1 & 0x10u) >> 4) << (j + QK / 2);
        }
        memcpy(y[i].qh, &qh, 4);
    }
}

static void quantize_row_q5_1(const float *x, blk_q5_1 *y, int64_t k) {
    for (int64_t i = 0; i < k / QK; i++) {
        float min = FLT_MAX, max = -FLT_MAX;
        for (int j = 0; j < QK; j++) {
            const float v = x[i * QK + j];
            if (v < min) min = v;
            if (v > max) max = v;
        }
        const float d  = (max - min) / ((1 << 5) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d         = bo_fp32_to_fp16(d);
        y[i].m         = bo_fp32_to_fp16(min);
        uint32_t qh    = 0;
        for (int j = 0; j < QK / 2; j++) {
            const float x0    = (x[i * QK + 0 + j] - min) * id;
            const float x1    = (x[i * QK + QK / 2 + j] - min) * id;
            const uint8_t xi0 = (uint8_t)(x0 + 0.5f);
            const uint8_t xi1 = (uint8_t)(x1 + 0.5f);
            y[i].qs[j]        = (uint8_t)((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
            qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
            qh |= ((xi1 & 0x10u) >> 4) << (j + QK / 2);
        }
        memcpy(y[i].qh, &qh, 4);
    }
}

static void quantize_row_q8_0(const float *x, blk_q8_0 *y, int64_t k) {
    for (int64_t i = 0; i < k / QK; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = fabsf(x[i * QK + j]);
            if (amax < v) amax = v;
        }
        const float d  = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d         = bo_fp32_to_fp16(d);
        for (int j = 0; j < QK; j++) {
            const float x0 = x[i * QK + j] * id;
            y[i].qs[j]     = (int8_t)roundf(x0); /* scalar path: half away from zero */
        }
    }
}

static void quantize_row_q8_1(const float *x, blk_q8_1 *y, int64_t k) {
    for (int64_t i = 0; i < k / QK; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = fabsf(x[i * QK + j]);
            if (amax < v) amax = v;
        }
        const float d  = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d         = d;
        int sum        = 0;
        for (int j = 0; j < QK / 2; j++) {
            const float v0        = x[i * QK + j] * id;
            const float v1        = x[i * QK + QK / 2 + j] * id;
            y[i].qs[j]            = (int8_t)roundf(v0);
            y[i].qs[QK / 2 + j]   = (int8_t)roundf(v1);
            sum += y[i].qs[j];
            sum += y[i].qs[QK / 2 + j];
        }
        y[i].s = sum * d;
    }
}

/* ggml's AVX2 quantize_row_q8_0 / q8_1 [ggml-recall]: the multiplier is 127/amax (not 1/d) and the rounding is
 * _mm256_round_ps(..., _MM_ROUND_NEAREST) = nearest-even.  nearbyintf under the default rounding mode. */
static void quantize_row_q8_0_simd(const float *x, blk_q8_0 *y, int64_t k) {
    for (int64_t i = 0; i < k / QK; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = fabsf(x[i * QK + j]);
            if (amax < v) amax = v;
        }
        const float d  = amax / 127.f;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        y[i].d         = bo_fp32_to_fp16(d);
        for (int j = 0; j < QK; j++) y[i].qs[j] = (int8_t)nearbyintf(x[i * QK + j] * id);
    }
}
static void quantize_row_q8_1_simd(const float *x, blk_q8_1 *y, int64_t k) {
    for (int64_t i = 0; i < k / QK; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = fabsf(x[i * QK + j]);
            if (amax < v) amax = v;
        }
        const float d  = amax / 127.f;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        y[i].d         = d;
        int sum        = 0;
        for (int j = 0; j < QK; j++) {
            y[i].qs[j] = (int8_t)nearbyintf(x[i * QK + j] * id);
            sum += y[i].qs[j];
        }
        y[i].s = d * (float)sum;
    }
}

size_t bo_quantize(int type, const float *src, void *dst, int64_t n, int64_t k) {
    const size_t rb = bo_row_bytes(type, k);
    uint8_t *out    = (uint8_t *)dst;
    for (int64_t b = 0; b < n; b += k) {
        void *y = out + (size_t)(b / k) * rb;
        switch (type) {
            case BO_TYPE_Q4_0: quantize_row_q4_0(src + b, (blk_q4_0 *)y, k); break;
            case BO_TYPE_Q4_1: quantize_row_q4_1(src + b, (blk_q4_1 *)y, k); break;
            case BO_TYPE_Q5_0: quantize_row_q5_0(src + b, (blk_q5_0 *)y, k); break;
            case BO_TYPE_Q5_1: quantize_row_q5_1(src + b, (blk_q5_1 *)y, k); break;
            case BO_TYPE_Q8_0: quantize_row_q8_0(src + b, (blk_q8_0 *)y, k); break;
            case BO_TYPE_Q8_1: quantize_row_q8_1(src + b, (blk_q8_1 *)y, k); break;
            case BO_TYPE_F16: {
                uint16_t *h = (uint16_t *)y;
                for (int64_t i = 0; i < k; i++) h[i] = bo_fp32_to_fp16(src[b + i]);
            } break;
            case BO_TYPE_F32: memcpy(y, src + b, (size_t)k * 4); break;
            default: return 0;
        }
    }
    return (size_t)(n / k) * rb;
}

/* dequantize_row_* [ggml-recall, SURVEY A.1] */
void bo_dequantize_row(int type, const void *src, float *dst, int64_t k) {
    init_tables();
    const int64_t nb = k / QK;
    switch (type) {
        case BO_TYPE_F32: memcpy(dst, src, (size_t)k * 4); break;
        case BO_TYPE_F16: {
            const uint16_t *h = (const uint16_t *)src;
            for (int64_t i = 0; i < k; i++) dst[i] = F16(h[i]);
        } break;
        case BO_TYPE_Q4_0: {
            const blk_q4_0 *x = (const blk_q4_0 *)src;
            for (int64_t i = 0; i < nb; i++) {
                const float d = F16(x[i].d);
                for (int j = 0; j < QK / 2; j++) {
                    const int x0             = (x[i].qs[j] & 0x0F) - 8;
                    const int x1             = (x[i].qs[j] >> 4) - 8;
                    dst[i * QK + j + 0]      = x0 * d;
                    dst[i * QK + j + QK / 2] = x1 * d;
                }
            }
        } break;
        case BO_TYPE_Q4_1: {
            const blk_q4_1 *x = (const blk_q4_1 *)src;
            for (int64_t i = 0; i < nb; i++) {
                const float d = F16(x[i].d), m = F16(x[i].m);
                for (int j = 0; j < QK / 2; j++) {
                    const int x0             = (x[i].qs[j] & 0x0F);
                    const int x1             = (x[i].qs[j] >> 4);
                    dst[i * QK + j + 0]      = x0 * d + m;
                    dst[i * QK + j + QK / 2] = x1 * d + m;
                }
            }
        } break;
        case BO_TYPE_Q5_0: {
            const blk_q5_0 *x = (const blk_q5_0 *)src;
            for (int64_t i = 0; i < nb; i++) {
                const float d = F16(x[i].d);
                uint32_t qh;
                memcpy(&qh, x[i].qh, 4);
                for (int j = 0; j < QK / 2; j++) {
                    const uint8_t xh_0       = ((qh >> (j + 0)) << 4) & 0x10;
                    const uint8_t xh_1       = ((qh >> (j + 12))) & 0x10;
                    const int32_t x0         = ((x[i].qs[j] & 0x0F) | xh_0) - 16;
                    const int32_t x1         = ((x[i].qs[j] >> 4) | xh_1) - 16;
                    dst[i * QK + j + 0]      = x0 * d;
                    dst[i * QK + j + QK / 2] = x1 * d;
                }
            }
        } break;
        case BO_TYPE_Q5_1: {
            const blk_q5_1 *x = (const blk_q5_1 *)src;
            for (int64_t i = 0; i < nb; i++) {
                const float d = F16(x[i].d), m = F16(x[i].m);
                uint32_t qh;
                memcpy(&qh, x[i].qh, 4);
                for (int j = 0; j < QK / 2; j++) {
                    const uint8_t xh_0       = ((qh >> (j + 0)) << 4) & 0x10;
                    const uint8_t xh_1       = ((qh >> (j + 12))) & 0x10;
                    const int x0             = (x[i].qs[j] & 0x0F) | xh_0;
                    const int x1             = (x[i].qs[j] >> 4) | xh_1;
                    dst[i * QK + j + 0]      = x0 * d + m;
                    dst[i * QK + j + QK / 2] = x1 * d + m;
                }
            }
        } break;
        case BO_TYPE_Q8_0: {
            const blk_q8_0 *x = (const blk_q8_0 *)src;
            for (int64_t i = 0; i < nb; i++) {
                const float d = F16(x[i].d);
                for (int j = 0; j < QK; j++) dst[i * QK + j] = x[i].qs[j] * d;
            }
        } break;
        default: break;
    }
}

/* ggml_vec_dot_*_q8_* scalar paths [ggml-recall, SURVEY A.3] -- y already in the vec_dot_type */
static float vec_dot_q4_0_q8_0(int64_t k, const blk_q4_0 *x, const blk_q8_0 *y) {
    float sumf = 0.0f;
    for (int64_t i = 0; i < k / QK; i++) {
        int sumi = 0;
        for (int j = 0; j < QK / 2; j++) {
            const int v0 = (x[i].qs[j] & 0x0F) - 8;
            const int v1 = (x[i].qs[j] >> 4) - 8;
            sumi += (v0 * y[i].qs[j]) + (v1 * y[i].qs[j + QK / 2]);
        }
        sumf += sumi * F16(x[i].d) * F16(y[i].d);
    }
    return sumf;
}
static float vec_dot_q4_1_q8_1(int64_t k, const blk_q4_1 *x, const blk_q8_1 *y) {
    float sumf = 0.0f;
    for (int64_t i = 0; i < k / QK; i++) {
        int sumi = 0;
        for (int j = 0; j < QK / 2; j++) {
            const int v0 = (x[i].qs[j] & 0x0F);
            const int v1 = (x[i].qs[j] >> 4);
            sumi += (v0 * y[i].qs[j]) + (v1 * y[i].qs[j + QK / 2]);
        }
        sumf += (F16(x[i].d) * y[i].d) * sumi + F16(x[i].m) * y[i].s;
    }
    return sumf;
}
static float vec_dot_q5_0_q8_0(int64_t k, const blk_q5_0 *x, const blk_q8_0 *y) {
    float sumf = 0.0f;
    for (int64_t i = 0; i < k / QK; i++) {
        uint32_t qh;
        memcpy(&qh, x[i].qh, 4);
        int sumi = 0;
        for (int j = 0; j < QK / 2; j++) {
            const uint8_t xh_0 = ((qh & (1u << (j + 0))) >> (j + 0)) << 4;
            const uint8_t xh_1 = ((qh & (1u << (j + 16))) >> (j + 12));
            const int32_t x0   = ((x[i].qs[j] & 0x0F) | xh_0) - 16;
            const int32_t x1   = ((x[i].qs[j] >> 4) | xh_1) - 16;
            sumi += (x0 * y[i].qs[j]) + (x1 * y[i].qs[j + QK / 2]);
        }
        sumf += (F16(x[i].d) * F16(y[i].d)) * sumi;
    }
    return sumf;
}
static float vec_dot_q5_1_q8_1(int64_t k, const blk_q5_1 *x, const blk_q8_1 *y) {
    float sumf = 0.0f;
    for (int64_t i = 0; i < k / QK; i++) {
        uint32_t qh;
        memcpy(&qh, x[i].qh, 4);
        int sumi = 0;
        for (int j = 0; j < QK / 2; j++) {
            const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10;
            const uint8_t xh_1 = ((qh >> (j + 12))) & 0x10;
            const int32_t x0   = (x[i].qs[j] & 0xF) | xh_0;
            const int32_t x1   = (x[i].qs[j] >> 4) | xh_1;
            sumi += (x0 * y[i].qs[j]) + (x1 * y[i].qs[j + QK / 2]);
        }
        sumf += (F16(x[i].d) * y[i].d) * sumi + F16(x[i].m) * y[i].s;
    }
    return sumf;
}
static float vec_dot_q8_0_q8_0(int64_t k, const blk_q8_0 *x, const blk_q8_0 *y) {
    float sumf = 0.0f;
    for (int64_t i = 0; i < k / QK; i++) {
        int sumi = 0;
        for (int j = 0; j < QK; j++) sumi += x[i].qs[j] * y[i].qs[j];
        sumf += sumi * (F16(x[i].d) * F16(y[i].d));
    }
    return sumf;
}
/* ggml_vec_dot_f32 / _f16 scalar fallbacks: product in f32, running sum in double */
static float vec_dot_f32(int64_t k, const float *x, const float *y) {
    double sumf = 0.0;
    for (int64_t i = 0; i < k; i++) sumf += (double)(x[i] * y[i]);
    return (float)sumf;
}
static float vec_dot_f16(int64_t k, const uint16_t *x, const uint16_t *y) {
    double sumf = 0.0;
    for (int64_t i = 0; i < k; i++) sumf += (double)(F16(x[i]) * F16(y[i]));
    return (float)sumf;
}

/* ---- the AVX2 shape of the same dots [ggml-recall]: per block, 8 lanes hold the int32 sums of 4 consecutive
 * byte products (mul_sum_i8_pairs_float), acc_l = fma(d, (float)sum_l, acc_l); at the end hsum_float_8:
 * ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)).  wv[j] is the weight element j of the block as an integer. ---- */
static inline float hsum8(const float a[8]) {
    const float r0 = a[0] + a[4], r1 = a[1] + a[5], r2 = a[2] + a[6], r3 = a[3] + a[7];
    return (r0 + r2) + (r1 + r3);
}
static inline void lanes8(const int wv[32], const int8_t *yq, float q[8]) {
    for (int l = 0; l < 8; l++) {
        int s4 = 0;
        for (int j = 0; j < 4; j++) s4 += wv[4 * l + j] * yq[4 * l + j];
        q[l] = (float)s4;
    }
}
static float vec_dot_simd(int wtype, int64_t k, const void *w, const void *yv) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float summs  = 0.0f;
    for (int64_t i = 0; i < k / QK; i++) {
        int wv[32];
        float d = 0.0f;
        const int8_t *yq = NULL;
        switch (wtype) {
            case BO_TYPE_Q4_0: {
                const blk_q4_0 *x = (const blk_q4_0 *)w; const blk_q8_0 *y = (const blk_q8_0 *)yv;
                for (int j = 0; j < 16; j++) { wv[j] = (x[i].qs[j] & 0x0F) - 8; wv[j + 16] = (x[i].qs[j] >> 4) - 8; }
                d = F16(x[i].d) * F16(y[i].d); yq = y[i].qs;
            } break;
            case BO_TYPE_Q5_0: {
                const blk_q5_0 *x = (const blk_q5_0 *)w; const blk_q8_0 *y = (const blk_q8_0 *)yv;
                uint32_t qh; memcpy(&qh, x[i].qh, 4);
                for (int j = 0; j < 16; j++) {
                    wv[j]      = ((x[i].qs[j] & 0x0F) | (((qh >> j) & 1u) << 4)) - 16;
                    wv[j + 16] = ((x[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1u) << 4)) - 16;
                }
                d = F16(x[i].d) * F16(y[i].d); yq = y[i].qs;
            } break;
            case BO_TYPE_Q8_0: {
                const blk_q8_0 *x = (const blk_q8_0 *)w; const blk_q8_0 *y = (const blk_q8_0 *)yv;
                for (int j = 0; j < 32; j++) wv[j] = x[i].qs[j];
                d = F16(x[i].d) * F16(y[i].d); yq = y[i].qs;
            } break;
            case BO_TYPE_Q4_1: {
                const blk_q4_1 *x = (const blk_q4_1 *)w; const blk_q8_1 *y = (const blk_q8_1 *)yv;
                for (int j = 0; j < 16; j++) { wv[j] = (x[i].qs[j] & 0x0F); wv[j + 16] = (x[i].qs[j] >> 4); }
                d = F16(x[i].d) * y[i].d; yq = y[i].qs;
                summs += F16(x[i].m) * y[i].s;
            } break;
            case BO_TYPE_Q5_1: {
                const blk_q5_1 *x = (const blk_q5_1 *)w; const blk_q8_1 *y = (const blk_q8_1 *)yv;
                uint32_t qh; memcpy(&qh, x[i].qh, 4);
                for (int j = 0; j < 16; j++) {
                    wv[j]      = (x[i].qs[j] & 0x0F) | (((qh >> j) & 1u) << 4);
                    wv[j + 16] = (x[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1u) << 4);
                }
                d = F16(x[i].d) * y[i].d; yq = y[i].qs;
                summs += F16(x[i].m) * y[i].s;
            } break;
            default: return NAN;
        }
        float q[8];
        lanes8(wv, yq, q);
        for (int l = 0; l < 8; l++) acc[l] = fmaf(d, q[l], acc[l]);
    }
    return hsum8(acc) + summs;
}
/* ---- the same shapes written with the intrinsics themselves (x86-64 with AVX2 + FMA; chosen at run time, bo_opts.assoc bit 2): what an AVX2 build of ggml [ggml-recall]
 * executes per block -- bytes_from_nibbles_32, mul_sum_i8_pairs_float (vpsignb / vpmaddubsw / vpmaddwd), _mm256_fmadd_ps, hsum_float_8 -- so that the CPU baseline timed
 * beside the GPU is SIMD code and not its scalar emulation.  Results equal vec_dot_simd / quantize_row_q8_*_simd bit for bit (tests/test_oracle_assoc.py). ---- */
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define BO_HAVE_AVX2 1
#define BO_AVX2 __attribute__((target("avx2,fma")))
static int bo_cpu_avx2(void) {
    static int have = -1;
    if (have < 0) { __builtin_cpu_init(); have = (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0; }
    return have;
}
BO_AVX2 static inline float hsum8_avx2(__m256 x) {
    __m128 res = _mm_add_ps(_mm256_castps256_ps128(x), _mm256_extractf128_ps(x, 1));     /* (a0+a4, a1+a5, a2+a6, a3+a7) */
    res        = _mm_add_ps(res, _mm_movehl_ps(res, res));                                /* (r0+r2, r1+r3) */
    res        = _mm_add_ss(res, _mm_movehdup_ps(res));
    return _mm_cvtss_f32(res);
}
BO_AVX2 static inline __m256 mul_sum_i8_pairs_float(__m256i x, __m256i y) {              /* 8 lanes: float of the int32 sum of 4 consecutive byte products */
    const __m256i ax  = _mm256_sign_epi8(x, x);
    const __m256i sy  = _mm256_sign_epi8(y, x);
    const __m256i dot = _mm256_maddubs_epi16(ax, sy);
    return _mm256_cvtepi32_ps(_mm256_madd_epi16(dot, _mm256_set1_epi16(1)));
}
BO_AVX2 static inline __m256i bytes_from_nibbles_32(const uint8_t *rsi) {                /* elements 0..15 = low nibbles, 16..31 = high nibbles */
    const __m128i tmp   = _mm_loadu_si128((const __m128i *)rsi);
    const __m256i bytes = _mm256_set_m128i(_mm_srli_epi16(tmp, 4), tmp);
    return _mm256_and_si256(_mm256_set1_epi8(0x0F), bytes);
}
BO_AVX2 static float vec_dot_avx2(int wtype, int64_t k, const void *w, const void *yv) {
    __m256 acc   = _mm256_setzero_ps();
    float summs  = 0.0f;
    for (int64_t i = 0; i < k / QK; i++) {
        __m256i bx, by;
        float d;
        switch (wtype) {
            case BO_TYPE_Q4_0: {
                const blk_q4_0 *x = (const blk_q4_0 *)w; const blk_q8_0 *y = (const blk_q8_0 *)yv;
                bx = _mm256_sub_epi8(bytes_from_nibbles_32(x[i].qs), _mm256_set1_epi8(8));
                by = _mm256_loadu_si256((const __m256i *)y[i].qs);
                d  = F16(x[i].d) * F16(y[i].d);
            } break;
            case BO_TYPE_Q8_0: {
                const blk_q8_0 *x = (const blk_q8_0 *)w; const blk_q8_0 *y = (const blk_q8_0 *)yv;
                bx = _mm256_loadu_si256((const __m256i *)x[i].qs);
                by = _mm256_loadu_si256((const __m256i *)y[i].qs);
                d  = F16(x[i].d) * F16(y[i].d);
            } break;
            case BO_TYPE_Q4_1: {
                const blk_q4_1 *x = (const blk_q4_1 *)w; const blk_q8_1 *y = (const blk_q8_1 *)yv;
                bx = bytes_from_nibbles_32(x[i].qs);
                by = _mm256_loadu_si256((const __m256i *)y[i].qs);
                d  = F16(x[i].d) * y[i].d;
                summs += F16(x[i].m) * y[i].s;
            } break;
            case BO_TYPE_Q5_0: case BO_TYPE_Q5_1: {       /* the fifth bits: unpacked in scalar code (ggml: bytes_from_bits_32), the dot is the vector one */
                int8_t wb[32];
                uint32_t qh;
                if (wtype == BO_TYPE_Q5_0) {
                    const blk_q5_0 *x = (const blk_q5_0 *)w; const blk_q8_0 *y = (const blk_q8_0 *)yv;
                    memcpy(&qh, x[i].qh, 4);
                    for (int j = 0; j < 16; j++) {
                        wb[j]      = (int8_t)(((x[i].qs[j] & 0x0F) | (((qh >> j) & 1u) << 4)) - 16);
                        wb[j + 16] = (int8_t)(((x[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1u) << 4)) - 16);
                    }
                    by = _mm256_loadu_si256((const __m256i *)y[i].qs);
                    d  = F16(x[i].d) * F16(y[i].d);
                } else {
                    const blk_q5_1 *x = (const blk_q5_1 *)w; const blk_q8_1 *y = (const blk_q8_1 *)yv;
                    memcpy(&qh, x[i].qh, 4);
                    for (int j = 0; j < 16; j++) {
                        wb[j]      = (int8_t)((x[i].qs[j] & 0x0F) | (((qh >> j) & 1u) << 4));
                        wb[j + 16] = (int8_t)((x[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1u) << 4));
                    }
                    by = _mm256_loadu_si256((const __m256i *)y[i].qs);
                    d  = F16(x[i].d) * y[i].d;
                    summs += F16(x[i].m) * y[i].s;
                }
                bx = _mm256_loadu_si256((const __m256i *)wb);
            } break;
            default: return NAN;
        }
        acc = _mm256_fmadd_ps(_mm256_set1_ps(d), mul_sum_i8_pairs_float(bx, by), acc);
    }
    return hsum8_avx2(acc) + summs;
}
BO_AVX2 static float vec_dot_f32_avx2(int64_t n, const float *x, const float *y) {
    __m256 s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
    const int64_t np = n & ~(int64_t)31;
    for (int64_t i = 0; i < np; i += 32) {
        s0 = _mm256_fmadd_ps(_mm256_loadu_ps(x + i), _mm256_loadu_ps(y + i), s0);
        s1 = _mm256_fmadd_ps(_mm256_loadu_ps(x + i + 8), _mm256_loadu_ps(y + i + 8), s1);
        s2 = _mm256_fmadd_ps(_mm256_loadu_ps(x + i + 16), _mm256_loadu_ps(y + i + 16), s2);
        s3 = _mm256_fmadd_ps(_mm256_loadu_ps(x + i + 24), _mm256_loadu_ps(y + i + 24), s3);
    }
    s0 = _mm256_add_ps(s0, s2); s1 = _mm256_add_ps(s1, s3); s0 = _mm256_add_ps(s0, s1);
    const __m128 t0 = _mm_add_ps(_mm256_castps256_ps128(s0), _mm256_extractf128_ps(s0, 1));
    const __m128 t1 = _mm_hadd_ps(t0, t0);
    float sumf      = _mm_cvtss_f32(_mm_hadd_ps(t1, t1));
    for (int64_t i = np; i < n; i++) sumf += x[i] * y[i];
    return sumf;
}
/* one block of quantize_row_q8_0 / q8_1 as AVX2 ggml does it: max |x| over 4 vectors, id = 127 / amax, nearest-even rounding, pack to int8; returns the block's integer sum */
BO_AVX2 static inline int quantize_block_q8_avx2(const float *x, int8_t *qs, float *d_out) {
    const __m256 sign = _mm256_set1_ps(-0.0f);
    __m256 v0 = _mm256_loadu_ps(x), v1 = _mm256_loadu_ps(x + 8), v2 = _mm256_loadu_ps(x + 16), v3 = _mm256_loadu_ps(x + 24);
    __m256 mx = _mm256_max_ps(_mm256_max_ps(_mm256_andnot_ps(sign, v0), _mm256_andnot_ps(sign, v1)), _mm256_max_ps(_mm256_andnot_ps(sign, v2), _mm256_andnot_ps(sign, v3)));
    __m128 m4 = _mm_max_ps(_mm256_extractf128_ps(mx, 1), _mm256_castps256_ps128(mx));
    m4        = _mm_max_ps(m4, _mm_movehl_ps(m4, m4));
    m4        = _mm_max_ss(m4, _mm_movehdup_ps(m4));
    const float amax = _mm_cvtss_f32(m4);
    *d_out           = amax / 127.f;
    const float id   = (amax != 0.0f) ? 127.f / amax : 0.0f;
    const __m256 mul = _mm256_set1_ps(id);
    __m256i i0 = _mm256_cvtps_epi32(_mm256_round_ps(_mm256_mul_ps(v0, mul), _MM_ROUND_NEAREST)), i1 = _mm256_cvtps_epi32(_mm256_round_ps(_mm256_mul_ps(v1, mul), _MM_ROUND_NEAREST));
    __m256i i2 = _mm256_cvtps_epi32(_mm256_round_ps(_mm256_mul_ps(v2, mul), _MM_ROUND_NEAREST)), i3 = _mm256_cvtps_epi32(_mm256_round_ps(_mm256_mul_ps(v3, mul), _MM_ROUND_NEAREST));
    const __m256i s8 = _mm256_add_epi32(_mm256_add_epi32(i0, i1), _mm256_add_epi32(i2, i3));
    i0 = _mm256_packs_epi32(i0, i1); i2 = _mm256_packs_epi32(i2, i3); i0 = _mm256_packs_epi16(i0, i2);
    i0 = _mm256_permutevar8x32_epi32(i0, _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7));       /* the packs interleave the 128-bit halves: put the bytes back in element order */
    _mm256_storeu_si256((__m256i *)qs, i0);
    __m128i h = _mm_add_epi32(_mm256_castsi256_si128(s8), _mm256_extracti128_si256(s8, 1));
    h         = _mm_add_epi32(h, _mm_shuffle_epi32(h, 0x4E));
    h         = _mm_add_epi32(h, _mm_shuffle_epi32(h, 0xB1));
    return _mm_cvtsi128_si32(h);
}
BO_AVX2 static void quantize_row_q8_0_avx2(const float *x, blk_q8_0 *y, int64_t k) {
    for (int64_t i = 0; i < k / QK; i++) { float d; (void)quantize_block_q8_avx2(x + i * QK, y[i].qs, &d); y[i].d = bo_fp32_to_fp16(d); }
}
BO_AVX2 static void quantize_row_q8_1_avx2(const float *x, blk_q8_1 *y, int64_t k) {
    for (int64_t i = 0; i < k / QK; i++) { float d; const int sum = quantize_block_q8_avx2(x + i * QK, y[i].qs, &d); y[i].d = d; y[i].s = d * (float)sum; }
}
#else
#define BO_HAVE_AVX2 0
static int bo_cpu_avx2(void) { return 0; }
#endif
int bo_have_avx2(void) { return bo_cpu_avx2(); }

/* ggml_vec_dot_f32 with GGML_SIMD (AVX2: step 32 = 4 accumulators x 8 lanes, fma; reduce 0+=2, 1+=3, 0+=1, then
 * low128 + high128 and two hadds; scalar leftovers in float) */
static float vec_dot_f32_simd(int64_t n, const float *x, const float *y) {
    float sum[4][8];
    memset(sum, 0, sizeof(sum));
    const int64_t np = n & ~(int64_t)31;
    for (int64_t i = 0; i < np; i += 32)
        for (int j = 0; j < 4; j++)
            for (int l = 0; l < 8; l++) sum[j][l] = fmaf(x[i + 8 * j + l], y[i + 8 * j + l], sum[j][l]);
    float t0[4];
    for (int l = 0; l < 8; l++) { sum[0][l] += sum[2][l]; sum[1][l] += sum[3][l]; }
    for (int l = 0; l < 8; l++) sum[0][l] += sum[1][l];
    for (int l = 0; l < 4; l++) t0[l] = sum[0][l] + sum[0][l + 4];
    float sumf = (t0[0] + t0[1]) + (t0[2] + t0[3]);
    for (int64_t i = np; i < n; i++) sumf += x[i] * y[i];
    return sumf;
}

/* bit 2 of bo_opts.assoc: the intrinsics where the CPU has them (same values as the emulations above, several times faster) */
static float vec_dot_simd_x(int intr, int wtype, int64_t k, const void *w, const void *yv) {
#if BO_HAVE_AVX2
    if (intr && bo_cpu_avx2()) return vec_dot_avx2(wtype, k, w, yv);
#endif
    (void)intr;
    return vec_dot_simd(wtype, k, w, yv);
}
static float vec_dot_f32_simd_x(int intr, int64_t n, const float *x, const float *y) {
#if BO_HAVE_AVX2
    if (intr && bo_cpu_avx2()) return vec_dot_f32_avx2(n, x, y);
#endif
    (void)intr;
    return vec_dot_f32_simd(n, x, y);
}
static void quantize_row_q8_simd_x(int intr, int q81, const float *x, void *y, int64_t k) {
#if BO_HAVE_AVX2
    if (intr && bo_cpu_avx2()) { if (q81) quantize_row_q8_1_avx2(x, (blk_q8_1 *)y, k); else quantize_row_q8_0_avx2(x, (blk_q8_0 *)y, k); return; }
#endif
    (void)intr;
    if (q81) quantize_row_q8_1_simd(x, (blk_q8_1 *)y, k); else quantize_row_q8_0_simd(x, (blk_q8_0 *)y, k);
}

/* activation ("src1") row conversion to the weight type's vec_dot_type [SURVEY A.3] */
static int vec_dot_type(int wtype) {
    switch (wtype) {
        case BO_TYPE_F32: return BO_TYPE_F32;
        case BO_TYPE_F16: return BO_TYPE_F16;
        case BO_TYPE_Q4_0:
        case BO_TYPE_Q5_0:
        case BO_TYPE_Q8_0: return BO_TYPE_Q8_0;
        case BO_TYPE_Q4_1:
        case BO_TYPE_Q5_1: return BO_TYPE_Q8_1;
        default: return -1;
    }
}

static float vec_dot_typed(int wtype, int64_t k, const void *w, const void *y) {
    switch (wtype) {
        case BO_TYPE_F32: return vec_dot_f32(k, (const float *)w, (const float *)y);
        case BO_TYPE_F16: return vec_dot_f16(k, (const uint16_t *)w, (const uint16_t *)y);
        case BO_TYPE_Q4_0: return vec_dot_q4_0_q8_0(k, (const blk_q4_0 *)w, (const blk_q8_0 *)y);
        case BO_TYPE_Q4_1: return vec_dot_q4_1_q8_1(k, (const blk_q4_1 *)w, (const blk_q8_1 *)y);
        case BO_TYPE_Q5_0: return vec_dot_q5_0_q8_0(k, (const blk_q5_0 *)w, (const blk_q8_0 *)y);
        case BO_TYPE_Q5_1: return vec_dot_q5_1_q8_1(k, (const blk_q5_1 *)w, (const blk_q8_1 *)y);
        case BO_TYPE_Q8_0: return vec_dot_q8_0_q8_0(k, (const blk_q8_0 *)w, (const blk_q8_0 *)y);
        default: return NAN;
    }
}

float bo_vec_dot(int wtype, int64_t k, const void *wrow, const float *x) {
    init_tables();
    const int vt = vec_dot_type(wtype);
    if (vt < 0) return NAN;
    void *y = malloc(bo_row_bytes(vt, k) + 64);
    bo_quantize(vt, x, y, k, k);
    const float r = vec_dot_typed(wtype, k, wrow, y);
    free(y);
    return r;
}

/* the same with the activation row already in the weight type's vec_dot_type blocks (blk_q8_0 / blk_q8_1 bytes): the scalar vec_dot alone */
float bo_vec_dot_q(int wtype, int64_t k, const void *wrow, const void *yblocks) {
    init_tables();
    if (vec_dot_type(wtype) < 0) return NAN;
    return vec_dot_typed(wtype, k, wrow, yblocks);
}

/* ------------------------------------------------------------------------------------------
 * model file
 * ---------------------------------------------------------------------------------------- */
typedef struct bo_tensor {
    char    name[96];
    int     type;
    int64_t ne0, ne1; /* ne0 = innermost (row length) */
    void   *data;
    size_t  nbytes;
} bo_tensor;

typedef struct bo_layer {
    bo_tensor *q_w, *k_w, *v_w, *o_w, *q_b, *k_b, *v_b, *o_b;
    bo_tensor *ln0_w, *ln0_b, *ln1_w, *ln1_b;
    bo_tensor *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} bo_layer;

struct bo_model {
    int32_t n_vocab, n_layer, n_head, n_positions, d_ff, d_model, ftype, n_merges;
    bo_tensor *tensors;
    int        n_tensors;
    bo_tensor *embed_tokens, *embed_pos, *ln_w, *ln_b, *lm_head;
    bo_layer  *layers;
    float     *memory_k, *memory_v; /* [n_layer][n_positions][d_model] F32, biogpt.cpp:331-335 */
    bo_opts    opts;
    float     *taps; /* [(n_layer+2)][tap_n][d_model] */
    int        tap_n;
};

static int ftype_to_type(int ftype) { /* SURVEY A.4 */
    switch (ftype) {
        case 0: return BO_TYPE_F32;
        case 1: return BO_TYPE_F16;
        case 2: return BO_TYPE_Q4_0;
        case 3: return BO_TYPE_Q4_1;
        case 7: return BO_TYPE_Q8_0;
        case 8: return BO_TYPE_Q5_0;
        case 9: return BO_TYPE_Q5_1;
        default: return -1;
    }
}

static bo_tensor *find_tensor(bo_model *m, const char *name) {
    for (int i = 0; i < m->n_tensors; i++)
        if (strcmp(m->tensors[i].name, name) == 0) return &m->tensors[i];
    return NULL;
}

#define FAIL(...)                                     \
    do {                                              \
        if (err) snprintf(err, errlen, __VA_ARGS__);  \
        if (f) fclose(f);                             \
        bo_free(m);                                   \
        return NULL;                                  \
    } while (0)

static int skip_strings(FILE *f, int32_t count) {
    for (int32_t i = 0; i < count; i++) {
        uint32_t len;
        if (fread(&len, 4, 1, f) != 1) return -1;
        if (fseek(f, (long)len, SEEK_CUR) != 0) return -1;
    }
    return 0;
}

bo_model *bo_load(const char *path, char *err, size_t errlen) {
    init_tables();
    bo_model *m = (bo_model *)calloc(1, sizeof(bo_model));
    FILE *f     = fopen(path, "rb");
    if (!f) FAIL("failed to open '%s'", path);

    uint32_t magic = 0;
    if (fread(&magic, 4, 1, f) != 1 || magic != BO_MAGIC) FAIL("bad magic"); /* biogpt.cpp:41-48 */
    int32_t hp[7];
    if (fread(hp, 4, 7, f) != 7) FAIL("short header"); /* biogpt.cpp:54-60 */
    m->n_vocab = hp[0]; m->n_layer = hp[1]; m->n_head = hp[2]; m->n_positions = hp[3];
    m->d_ff = hp[4]; m->d_model = hp[5]; m->ftype = hp[6];
    const int wtype = ftype_to_type(m->ftype);
    if (wtype < 0) FAIL("bad ftype value %d", m->ftype); /* biogpt.cpp:160-165 */

    int32_t nv = 0;
    if (fread(&nv, 4, 1, f) != 1 || nv != m->n_vocab) FAIL("bad vocab size %d != %d", nv, m->n_vocab);
    if (skip_strings(f, nv)) FAIL("short vocab");
    int32_t nm = 0;
    if (fread(&nm, 4, 1, f) != 1 || nm < 0) FAIL("bad merges count"); /* F6: count taken from the file */
    m->n_merges = nm;
    if (skip_strings(f, nm)) FAIL("short merges");

    const int cap = 5 + 16 * m->n_layer;
    m->tensors    = (bo_tensor *)calloc((size_t)cap, sizeof(bo_tensor));
    for (;;) { /* biogpt.cpp:369-434 */
        int32_t n_dims, length, ttype;
        if (fread(&n_dims, 4, 1, f) != 1) break; /* EOF */
        if (fread(&length, 4, 1, f) != 1 || fread(&ttype, 4, 1, f) != 1) FAIL("short tensor header");
        if (n_dims < 1 || n_dims > 2 || length <= 0 || length >= 96) FAIL("bad tensor header");
        int32_t ne[2] = {1, 1};
        for (int i = 0; i < n_dims; i++)
            if (fread(&ne[i], 4, 1, f) != 1) FAIL("short tensor dims");
        if (m->n_tensors >= cap) FAIL("too many tensors");
        bo_tensor *t = &m->tensors[m->n_tensors];
        if (fread(t->name, 1, (size_t)length, f) != (size_t)length) FAIL("short tensor name");
        t->name[length] = 0;
        t->type = ttype; t->ne0 = ne[0]; t->ne1 = ne[1];
        if (ttype != BO_TYPE_F32 && ttype != BO_TYPE_F16 && (bo_type_block_bytes(ttype) == 0 || ttype == BO_TYPE_Q8_1))
            FAIL("tensor '%s' has unsupported type %d", t->name, ttype);
        if (ttype != BO_TYPE_F32 && ttype != BO_TYPE_F16 && (ne[0] % QK) != 0) FAIL("tensor '%s' row not multiple of 32", t->name);
        t->nbytes = bo_row_bytes(ttype, ne[0]) * (size_t)ne[1];
        t->data   = malloc(t->nbytes + 64);
        if (!t->data) FAIL("oom");
        if (fread(t->data, 1, t->nbytes, f) != t->nbytes) FAIL("tensor '%s' truncated", t->name);
        m->n_tensors++;
    }
    fclose(f);
    f = NULL;

    if (m->n_tensors == 0) return m; /* "assuming empty model for testing" biogpt.cpp:442-443 */
    if (m->n_tensors != cap) FAIL("not all tensors loaded: expected %d, got %d", cap, m->n_tensors);

    /* bind names (biogpt.cpp:258-317) and check shapes (biogpt.cpp:401-417; F5: embed_pos rows from the file) */
    const int D = m->d_model, F = m->d_ff, V = m->n_vocab;
#define BIND(dst, nm_, e0, e1, matrix)                                                                 \
    do {                                                                                               \
        dst = find_tensor(m, nm_);                                                                     \
        if (!dst) FAIL("missing tensor '%s'", nm_);                                                    \
        if (dst->ne0 != (e0) || ((e1) >= 0 && dst->ne1 != (e1))) FAIL("tensor '%s' has wrong shape", nm_); \
        if (!(matrix) && dst->type != BO_TYPE_F32) FAIL("tensor '%s' must be F32", nm_);               \
        if ((matrix) && dst->type != wtype) FAIL("tensor '%s' type %d != file wtype %d", nm_, dst->type, wtype); \
    } while (0)
    BIND(m->lm_head, "output_projection.weight", D, V, 1);
    BIND(m->embed_tokens, "biogpt.embed_tokens.weight", D, V, 1);
    BIND(m->embed_pos, "biogpt.embed_positions.weight", D, -1, 1);
    if (m->embed_pos->ne1 < m->n_positions + 2) FAIL("embed_positions has too few rows");
    BIND(m->ln_w, "biogpt.layer_norm.weight", D, 1, 0);
    BIND(m->ln_b, "biogpt.layer_norm.bias", D, 1, 0);
    m->layers = (bo_layer *)calloc((size_t)m->n_layer, sizeof(bo_layer));
    for (int i = 0; i < m->n_layer; i++) {
        bo_layer *L = &m->layers[i];
        char nm_[128];
#define LN(dst, suffix, e0, e1, matrix)                                     \
    snprintf(nm_, sizeof nm_, "biogpt.layers.%d." suffix, i);             \
    BIND(dst, nm_, e0, e1, matrix)
        LN(L->q_w, "self_attn.q_proj.weight", D, D, 1);
        LN(L->k_w, "self_attn.k_proj.weight", D, D, 1);
        LN(L->v_w, "self_attn.v_proj.weight", D, D, 1);
        LN(L->o_w, "self_attn.out_proj.weight", D, D, 1);
        LN(L->q_b, "self_attn.q_proj.bias", D, 1, 0);
        LN(L->k_b, "self_attn.k_proj.bias", D, 1, 0);
        LN(L->v_b, "self_attn.v_proj.bias", D, 1, 0);
        LN(L->o_b, "self_attn.out_proj.bias", D, 1, 0);
        LN(L->ln0_w, "self_attn_layer_norm.weight", D, 1, 0);
        LN(L->ln0_b, "self_attn_layer_norm.bias", D, 1, 0);
        LN(L->ln1_w, "final_layer_norm.weight", D, 1, 0);
        LN(L->ln1_b, "final_layer_norm.bias", D, 1, 0);
        LN(L->fc1_w, "fc1.weight", D, F, 1);
        LN(L->fc1_b, "fc1.bias", F, 1, 0);
        LN(L->fc2_w, "fc2.weight", F, D, 1);
        LN(L->fc2_b, "fc2.bias", D, 1, 0);
    }
    const size_t kv = (size_t)m->n_layer * (size_t)m->n_positions * (size_t)D;
    m->memory_k = (float *)calloc(kv, 4);
    m->memory_v = (float *)calloc(kv, 4);
    if (!m->memory_k || !m->memory_v) FAIL("oom (kv)");
    return m;
}

void bo_free(bo_model *m) {
    if (!m) return;
    if (m->tensors) {
        for (int i = 0; i < m->n_tensors; i++) free(m->tensors[i].data);
        free(m->tensors);
    }
    free(m->layers);
    free(m->memory_k);
    free(m->memory_v);
    free(m->taps);
    free(m);
}

void bo_hparams(const bo_model *m, int32_t out[8]) {
    out[0] = m->n_vocab; out[1] = m->n_layer; out[2] = m->n_head; out[3] = m->n_positions;
    out[4] = m->d_ff; out[5] = m->d_model; out[6] = m->ftype; out[7] = m->n_merges;
}
void bo_set_opts(bo_model *m, const bo_opts *o) { m->opts = *o; }
int  bo_n_tensors(const bo_model *m) { return m->n_tensors; }
const float *bo_kv(const bo_model *m, int which) { return which ? m->memory_v : m->memory_k; }

/* ------------------------------------------------------------------------------------------
 * ops, in ggml's scalar formulation [ggml-recall, SURVEY A.5]
 * ---------------------------------------------------------------------------------------- */

/* ggml_mul_mat(W, x): out[n][m] = vec_dot(W row m, convert(x row n))   */
static void mul_mat(const bo_model *mdl, const bo_tensor *W, const float *x, int N, float *out, int64_t m_lo, int64_t m_hi) {
    const int64_t K = W->ne0, M = W->ne1;
    const int vt    = vec_dot_type(W->type);
    const size_t yb = bo_row_bytes(vt, K);
    uint8_t *y      = NULL;
    if (vt != BO_TYPE_F32) {
        y = (uint8_t *)malloc(yb * (size_t)N + 64);
        for (int n = 0; n < N; n++) {
            if ((mdl->opts.assoc & 2) && vt == BO_TYPE_Q8_0) quantize_row_q8_simd_x(mdl->opts.assoc & 4, 0, x + (size_t)n * K, y + (size_t)n * yb, K);
            else if ((mdl->opts.assoc & 2) && vt == BO_TYPE_Q8_1) quantize_row_q8_simd_x(mdl->opts.assoc & 4, 1, x + (size_t)n * K, y + (size_t)n * yb, K);
            else bo_quantize(vt, x + (size_t)n * K, y + (size_t)n * yb, K, K);
        }
    }
    const int simd = (mdl->opts.assoc & 1) && W->type != BO_TYPE_F32 && W->type != BO_TYPE_F16;
    const size_t wb   = bo_row_bytes(W->type, K);
    const int threads = mdl->opts.n_threads > 0 ? mdl->opts.n_threads : 1;
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t mm = m_lo; mm < m_hi; mm++) {
        const uint8_t *wrow = (const uint8_t *)W->data + (size_t)mm * wb;
        for (int n = 0; n < N; n++) {
            const void *yy            = (vt == BO_TYPE_F32) ? (const void *)(x + (size_t)n * K) : (const void *)(y + (size_t)n * yb);
            out[(size_t)n * M + mm]   = simd ? vec_dot_simd_x(mdl->opts.assoc & 4, W->type, K, wrow, yy)
                                             : ((mdl->opts.assoc & 1) && W->type == BO_TYPE_F32) ? vec_dot_f32_simd_x(mdl->opts.assoc & 4, K, (const float *)wrow, (const float *)yy)
                                                                                               : vec_dot_typed(W->type, K, wrow, yy);
        }
    }
    free(y);
}

/* ggml_norm then (repeat(w) * cur) + repeat(b)  (biogpt.cpp:693-700) */
static void layer_norm(const float *x, const float *w, const float *b, float *y, int n, float eps) {
    double sum = 0.0;
    for (int i = 0; i < n; i++) sum += (double)x[i];
    const float mean = (float)(sum / n);
    double sum2      = 0.0;
    for (int i = 0; i < n; i++) {
        const float v = x[i] - mean;
        y[i]          = v;
        sum2 += (double)(v * v);
    }
    const float variance = (float)(sum2 / n);
    const float scale    = 1.0f / sqrtf(variance + eps);
    for (int i = 0; i < n; i++) {
        float v = y[i] * scale; /* ggml_vec_scale_f32 */
        v       = w[i] * v;     /* ggml_mul            */
        y[i]    = v + b[i];     /* ggml_add            */
    }
}

static inline float gelu_erf_f32(float x) { return (float)(0.5 * (double)x * (1.0 + erf((double)x / 1.4142135623730951))); }

/* ------------------------------------------------------------------------------------------
 * forward pass: biogpt_graph (biogpt.cpp:624-810) + row selection of biogpt_eval (:840-844)
 * ---------------------------------------------------------------------------------------- */
int bo_eval(bo_model *m, const int32_t *tokens, int N, int n_past, float *logits_last, float *logits_all) {
    if (!m || !m->layers || N < 1) return -1;
    const int D = m->d_model, F = m->d_ff, V = m->n_vocab, H = m->n_head, P = m->n_positions, L = m->n_layer;
    const int dk = D / H;
    const int T  = n_past + N;
    if (n_past < 0 || T > P) return -2;
    for (int i = 0; i < N; i++)
        if (tokens[i] < 0 || tokens[i] >= V) return -3;
    const float eps   = m->opts.ln_eps > 0.0f ? m->opts.ln_eps : NORM_EPS;
    const int threads = m->opts.n_threads > 0 ? m->opts.n_threads : 1;
    (void)threads;

    float *inpL = (float *)malloc(sizeof(float) * (size_t)N * D);
    float *cur  = (float *)malloc(sizeof(float) * (size_t)N * D);
    float *q    = (float *)malloc(sizeof(float) * (size_t)N * D);
    float *kc   = (float *)malloc(sizeof(float) * (size_t)N * D);
    float *vc   = (float *)malloc(sizeof(float) * (size_t)N * D);
    float *att  = (float *)malloc(sizeof(float) * (size_t)N * D);
    float *ff   = (float *)malloc(sizeof(float) * (size_t)N * F);
    float *tmp  = (float *)malloc(sizeof(float) * (size_t)N * D);
    float *tmpr = (float *)malloc(sizeof(float) * (size_t)D);

    free(m->taps);
    m->tap_n = N;
    m->taps  = (float *)calloc((size_t)(L + 2) * N * D, sizeof(float));

    /* token embeddings * sqrt(d_model) + position embeddings at n_past+i+2 (biogpt.cpp:664-686; F7) */
    const float embed_scale = sqrtf((float)D);
    for (int i = 0; i < N; i++) {
        const size_t rb = bo_row_bytes(m->embed_tokens->type, D);
        bo_dequantize_row(m->embed_tokens->type, (const uint8_t *)m->embed_tokens->data + (size_t)tokens[i] * rb, tmp, D);
        for (int d = 0; d < D; d++) tmp[d] *= embed_scale; /* ggml_scale */
        const int32_t pos = n_past + i + 2;
        bo_dequantize_row(m->embed_pos->type, (const uint8_t *)m->embed_pos->data + (size_t)pos * rb, tmpr, D);
        for (int d = 0; d < D; d++) inpL[(size_t)i * D + d] = tmp[d] + tmpr[d]; /* ggml_add */
    }
    memcpy(m->taps, inpL, sizeof(float) * (size_t)N * D);

    const float q_scale = 1.0f / sqrtf((float)dk); /* biogpt.cpp:678-683 */

    for (int l = 0; l < L; l++) {
        const bo_layer *ly = &m->layers[l];
        /* self-attention layer norm (biogpt.cpp:691-701) */
        for (int i = 0; i < N; i++)
            layer_norm(inpL + (size_t)i * D, (const float *)ly->ln0_w->data, (const float *)ly->ln0_b->data, cur + (size_t)i * D, D, eps);

        /* q/k/v projections, bias, Q scale after bias (biogpt.cpp:705-718) */
        mul_mat(m, ly->q_w, cur, N, q, 0, D);
        mul_mat(m, ly->k_w, cur, N, kc, 0, D);
        mul_mat(m, ly->v_w, cur, N, vc, 0, D);
        for (int i = 0; i < N; i++)
            for (int d = 0; d < D; d++) {
                const size_t o = (size_t)i * D + d;
                q[o]  = ((const float *)ly->q_b->data)[d] + q[o];
                q[o]  = q[o] * q_scale;
                kc[o] = ((const float *)ly->k_b->data)[d] + kc[o];
                vc[o] = ((const float *)ly->v_b->data)[d] + vc[o];
            }
        /* KV append (biogpt.cpp:721-727) */
        float *Kl = m->memory_k + ((size_t)l * P) * D;
        float *Vl = m->memory_v + ((size_t)l * P) * D;
        memcpy(Kl + (size_t)n_past * D, kc, sizeof(float) * (size_t)N * D);
        memcpy(Vl + (size_t)n_past * D, vc, sizeof(float) * (size_t)N * D);

        /* attention over all T = n_past+N keys, no mask unless opts.causal (biogpt.cpp:729-764; F1) */
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int hi = 0; hi < H * N; hi++) {
            const int h = hi / N, i = hi % N;
            float *S  = (float *)malloc(sizeof(float) * (size_t)T);
            const float *qv = q + (size_t)i * D + (size_t)h * dk;
            int Tlim = T;
            if (m->opts.causal) Tlim = n_past + i + 1;
            for (int j = 0; j < Tlim; j++) /* KQ = mul_mat(K, Q) */
                S[j] = (m->opts.assoc & 1) ? vec_dot_f32_simd_x(m->opts.assoc & 4, dk, Kl + (size_t)j * D + (size_t)h * dk, qv) : vec_dot_f32(dk, Kl + (size_t)j * D + (size_t)h * dk, qv);
            /* ggml_soft_max */
            float mx = -INFINITY;
            for (int j = 0; j < Tlim; j++)
                if (S[j] > mx) mx = S[j];
            double sum = 0.0;
            for (int j = 0; j < Tlim; j++) {
                float val;
                if (m->opts.exp_f32) {
                    val = expf(S[j] - mx);
                } else {
                    val = F16(g_table_exp[bo_fp32_to_fp16(S[j] - mx)]);
                }
                sum += (double)val;
                S[j] = val;
            }
            sum              = 1.0 / sum;
            const float fsum = (float)sum;
            for (int j = 0; j < Tlim; j++) S[j] *= fsum; /* ggml_vec_scale_f32 */
            /* KQV = mul_mat(V_trans, attn_weights): dot over T for each of the dk dims */
            if (m->opts.assoc & 1) { /* V_trans row (contiguous over the keys) . probabilities, SIMD-shaped */
                float *vt_row = (float *)malloc(sizeof(float) * (size_t)Tlim);
                for (int d = 0; d < dk; d++) {
                    for (int j = 0; j < Tlim; j++) vt_row[j] = Vl[(size_t)j * D + (size_t)h * dk + d];
                    att[(size_t)i * D + (size_t)h * dk + d] = vec_dot_f32_simd_x(m->opts.assoc & 4, Tlim, vt_row, S);
                }
                free(vt_row);
            } else
            for (int d = 0; d < dk; d++) {
                double acc = 0.0;
                for (int j = 0; j < Tlim; j++) acc += (double)(Vl[(size_t)j * D + (size_t)h * dk + d] * S[j]);
                att[(size_t)i * D + (size_t)h * dk + d] = (float)acc;
            }
            free(S);
        }

        /* out projection + bias + residual (biogpt.cpp:767-772) */
        mul_mat(m, ly->o_w, att, N, cur, 0, D);
        for (int i = 0; i < N; i++)
            for (int d = 0; d < D; d++) {
                const size_t o = (size_t)i * D + d;
                cur[o] = cur[o] + ((const float *)ly->o_b->data)[d];
                cur[o] = cur[o] + inpL[o];
            }
        float *inpFF = cur; /* alias: cur now holds inpFF */

        /* feed forward (biogpt.cpp:777-795) */
        for (int i = 0; i < N; i++)
            layer_norm(inpFF + (size_t)i * D, (const float *)ly->ln1_w->data, (const float *)ly->ln1_b->data, tmp + (size_t)i * D, D, eps);
        mul_mat(m, ly->fc1_w, tmp, N, ff, 0, F);
        for (int i = 0; i < N; i++)
            for (int d = 0; d < F; d++) {
                const size_t o = (size_t)i * F + d;
                float v = ((const float *)ly->fc1_b->data)[d] + ff[o];
                if (m->opts.gelu_erf) v = gelu_erf_f32(v);
                else v = F16(g_table_gelu[bo_fp32_to_fp16(v)]); /* ggml_vec_gelu_f32 via fp16 table */
                ff[o] = v;
            }
        mul_mat(m, ly->fc2_w, ff, N, tmp, 0, D);
        for (int i = 0; i < N; i++)
            for (int d = 0; d < D; d++) {
                const size_t o = (size_t)i * D + d;
                float v = ((const float *)ly->fc2_b->data)[d] + tmp[o];
                inpL[o] = v + inpFF[o];
            }
        memcpy(m->taps + (size_t)(l + 1) * N * D, inpL, sizeof(float) * (size_t)N * D);
    }

    /* final layer norm + lm head (biogpt.cpp:799-803); only the rows that are returned (F8) */
    for (int i = 0; i < N; i++)
        layer_norm(inpL + (size_t)i * D, (const float *)m->ln_w->data, (const float *)m->ln_b->data, cur + (size_t)i * D, D, eps);
    memcpy(m->taps + (size_t)(L + 1) * N * D, cur, sizeof(float) * (size_t)N * D);
    if (logits_all) {
        mul_mat(m, m->lm_head, cur, N, logits_all, 0, V);
        if (logits_last) memcpy(logits_last, logits_all + (size_t)(N - 1) * V, sizeof(float) * (size_t)V);
    } else if (logits_last) {
        mul_mat(m, m->lm_head, cur + (size_t)(N - 1) * D, 1, logits_last, 0, V);
    }

    free(inpL); free(cur); free(q); free(kc); free(vc); free(att); free(ff); free(tmp); free(tmpr);
    return 0;
}

int bo_tap(const bo_model *m, int layer, float *out) {
    if (!m->taps) return -1;
    const int idx = layer + 1;
    if (idx < 0 || idx > m->n_layer + 1) return -2;
    memcpy(out, m->taps + (size_t)idx * m->tap_n * m->d_model, sizeof(float) * (size_t)m->tap_n * m->d_model);
    return m->tap_n;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* examples/main/main.cpp:91-151 with --top_k 1 (biogpt.cpp:928-935: arg-max) */
double bo_generate_greedy(bo_model *m, const int32_t *prompt, int n_prompt, int n_batch, int n_predict, int32_t *out_ids) {
    const int V = m->n_vocab;
    if (n_predict > m->n_positions - n_prompt) n_predict = m->n_positions - n_prompt; /* main.cpp:82 */
    float *logits  = (float *)malloc(sizeof(float) * (size_t)V);
    int32_t *embed = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_batch > 1 ? n_batch : 1));
    int n_embed = 0, n_past = 0, n_out = 0;
    double t_predict = 0.0;
    for (int i = 0; i < n_prompt + n_predict; i++) {
        if (n_embed > 0) {
            const double t0 = now_s();
            if (bo_eval(m, embed, n_embed, n_past, logits, NULL) != 0) { t_predict = -1.0; break; }
            t_predict += now_s() - t0;
        }
        n_past += n_embed;
        n_embed = 0;
        if (i >= n_prompt) {
            int best = 0;
            for (int v = 1; v < V; v++)
                if (logits[v] > logits[best]) best = v;
            embed[n_embed++] = best;
            out_ids[n_out++] = best;
        } else {
            for (int k = i; k < n_prompt; k++) {
                embed[n_embed++] = prompt[k];
                if (n_embed >= n_batch) break;
            }
            i += n_embed - 1;
        }
    }
    free(logits);
    free(embed);
    return t_predict;
}

/* ------------------------------------------------------------------------------------------
 * file -> file quantizer: quantize.cpp:8-135 (header/vocab/merges copy, ftype rewrite) +
 * biogpt.cpp:459-621 (tensor loop; rule: name contains "weight" && ne[1] != 1)
 * ---------------------------------------------------------------------------------------- */
static int copy_bytes(FILE *in, FILE *out, size_t n) {
    uint8_t buf[1 << 16];
    while (n) {
        const size_t c = n < sizeof buf ? n : sizeof buf;
        if (fread(buf, 1, c, in) != c) return -1;
        if (fwrite(buf, 1, c, out) != c) return -1;
        n -= c;
    }
    return 0;
}

int bo_quantize_file(const char *in_path, const char *out_path, int ftype, char *err, size_t errlen) {
    init_tables();
    const int qtype = ftype_to_type(ftype);
    if (qtype < 0 || bo_type_block_bytes(qtype) == 0) {
        if (err) snprintf(err, errlen, "invalid quantization ftype %d", ftype);
        return -1;
    }
    FILE *fi = fopen(in_path, "rb");
    FILE *fo = fi ? fopen(out_path, "wb") : NULL;
    int rc   = -1;
    float *f32 = NULL;
    uint8_t *raw = NULL, *qbuf = NULL;
#define QFAIL(...) do { if (err) snprintf(err, errlen, __VA_ARGS__); goto done; } while (0)
    if (!fi || !fo) QFAIL("open failed");
    uint32_t magic;
    int32_t hp[7];
    if (fread(&magic, 4, 1, fi) != 1 || magic != BO_MAGIC) QFAIL("bad magic");
    if (fread(hp, 4, 7, fi) != 7) QFAIL("short header");
    hp[6] = ftype; /* quantize.cpp:57 */
    fwrite(&magic, 4, 1, fo);
    fwrite(hp, 4, 7, fo);
    for (int pass = 0; pass < 2; pass++) { /* vocab then merges, verbatim */
        int32_t count;
        if (fread(&count, 4, 1, fi) != 1) QFAIL("short strings");
        fwrite(&count, 4, 1, fo);
        for (int32_t i = 0; i < count; i++) {
            uint32_t len;
            if (fread(&len, 4, 1, fi) != 1) QFAIL("short strings");
            fwrite(&len, 4, 1, fo);
            if (copy_bytes(fi, fo, len)) QFAIL("short strings");
        }
    }
    for (;;) {
        int32_t n_dims, length, ttype;
        if (fread(&n_dims, 4, 1, fi) != 1) break;
        if (fread(&length, 4, 1, fi) != 1 || fread(&ttype, 4, 1, fi) != 1) QFAIL("short tensor header");
        int32_t ne[2] = {1, 1};
        int64_t nel   = 1;
        for (int i = 0; i < n_dims; i++) {
            if (fread(&ne[i], 4, 1, fi) != 1) QFAIL("short dims");
            nel *= ne[i];
        }
        char name[256];
        if (length <= 0 || length >= 256 || fread(name, 1, (size_t)length, fi) != (size_t)length) QFAIL("bad name");
        name[length] = 0;
        const int quantize = (strstr(name, "weight") != NULL) && (ne[1] != 1); /* biogpt.cpp:523 */
        if (quantize) {
            if (ttype != BO_TYPE_F32 && ttype != BO_TYPE_F16) QFAIL("unsupported ttype for integer quantization");
            f32 = (float *)realloc(f32, sizeof(float) * (size_t)nel);
            if (ttype == BO_TYPE_F16) {
                raw = (uint8_t *)realloc(raw, (size_t)nel * 2);
                if (fread(raw, 2, (size_t)nel, fi) != (size_t)nel) QFAIL("truncated");
                for (int64_t i = 0; i < nel; i++) f32[i] = F16(((uint16_t *)raw)[i]);
            } else if (fread(f32, 4, (size_t)nel, fi) != (size_t)nel) QFAIL("truncated");
            ttype = qtype;
        } else {
            const size_t bpe = (ttype == 0) ? 4 : 2;
            raw = (uint8_t *)realloc(raw, (size_t)nel * bpe);
            if (fread(raw, bpe, (size_t)nel, fi) != (size_t)nel) QFAIL("truncated");
        }
        fwrite(&n_dims, 4, 1, fo);
        fwrite(&length, 4, 1, fo);
        fwrite(&ttype, 4, 1, fo);
        for (int i = 0; i < n_dims; i++) fwrite(&ne[i], 4, 1, fo);
        fwrite(name, 1, (size_t)length, fo);
        if (quantize) {
            if (ne[0] % QK) QFAIL("row length not a multiple of 32");
            const size_t qb = bo_row_bytes(qtype, ne[0]) * (size_t)ne[1];
            qbuf = (uint8_t *)realloc(qbuf, qb);
            bo_quantize(qtype, f32, qbuf, nel, ne[0]); /* biogpt.cpp:568: rows of ne[0] */
            fwrite(qbuf, 1, qb, fo);
        } else {
            fwrite(raw, (ttype == 0) ? 4 : 2, (size_t)nel, fo);
        }
    }
    rc = 0;
done:
    if (fi) fclose(fi);
    if (fo) fclose(fo);
    free(f32); free(raw); free(qbuf);
    return rc;
}
