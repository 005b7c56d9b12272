// Issue cost of the VALU instructions the prompt-pass attention is made of (attn_tile_kernel: v_pk_mul_f32, v_cvt_f64_f32, v_add_f64), alone and in the kernel's own mix,
// at 1 / 2 / 4 waves per SIMD.  Every stream is 8 independent chains (no instruction waits for the one before it), 64 instructions per loop trip, 512 trips; wave 0 of
// each workgroup stamps s_memtime around the loop; the figure is shader cycles per wave-instruction PER SIMD (cycles x waves on the SIMD / instructions of all of them).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/microbench24 tools/microbench24.hip && /tmp/microbench24
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY8(STMT) for (int r = 0; r < 8; r++) { REP8(STMT) }

enum { K_ADD_F64, K_FMA_F64, K_MUL_F64, K_CVT_F64_F32, K_CVT_F32_F64, K_PK_MUL_F32, K_MUL_F32, K_FMA_F32, K_PK_FMA_F32, K_ADD_U32, K_MOV_B32, K_CVT_F32_I32, K_ADD_F32, K_PK_ADD_F32, K_MIX_BLK_CVT, K_MIX_BLK_MAGIC, K_MIX_ATTN, K_MIX_FMA, K_MIX_SPLIT, K_N };
static const char *knames[K_N] = {"v_add_f64", "v_fma_f64", "v_mul_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_pk_mul_f32", "v_mul_f32", "v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_mov_b32", "v_cvt_f32_i32", "v_add_f32", "v_pk_add_f32",
                                  "mix, one block of the matrix-core chain: 4 x (cvt_f32_i32, mul, mul, add)", "mix, the same with the bias trick: 2 x (pk_add, pk_mul, pk_mul, pk_add)",
                                  "mix: pk_mul + 2 cvt_f64_f32 + 2 add_f64 (per 2 MACs)", "mix: pk_mul + 2 cvt_f64_f32 + 2 fma_f64(x,1,acc)", "mix: 2 mul_f32 + 2 cvt + 2 add"};
static const int kinstr[K_N] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 8 * 8 * 16, 8 * 8 * 8, 8 * 8 * 5, 8 * 8 * 5, 8 * 8 * 6};   // wave-instructions per loop trip

template <int KIND>
__global__ __launch_bounds__(256) void k(u64 *out, float *sink, float seed) {
    double a[8]; float f[8]; f2 p[8]; unsigned u[8]; double one = 1.0 + (double)seed * 0.0;
    for (int i = 0; i < 8; i++) { a[i] = (double)seed + i; f[i] = seed + i; p[i] = f2{seed + i, seed - i}; u[i] = (unsigned)i + (unsigned)seed; }
    double b = 1.0 + 1e-9 * threadIdx.x; float g = 1.0f + 1e-6f * threadIdx.x; f2 g2 = {g, g};
    __builtin_amdgcn_s_barrier();
    const u64 t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 512; it++) {
        if (KIND == K_ADD_F64) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            BODY8(S)
#undef S
        } else if (KIND == K_FMA_F64) {
#define S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(one));
            BODY8(S)
#undef S
        } else if (KIND == K_MUL_F64) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            BODY8(S)
#undef S
        } else if (KIND == K_CVT_F64_F32) {
#define S(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
            BODY8(S)
#undef S
        } else if (KIND == K_CVT_F32_F64) {
#define S(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            BODY8(S)
#undef S
        } else if (KIND == K_PK_MUL_F32) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(g2));
            BODY8(S)
#undef S
        } else if (KIND == K_MUL_F32) {
#define S(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g));
            BODY8(S)
#undef S
        } else if (KIND == K_FMA_F32) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g));
            BODY8(S)
#undef S
        } else if (KIND == K_PK_FMA_F32) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(g2));
            BODY8(S)
#undef S
        } else if (KIND == K_ADD_U32) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            BODY8(S)
#undef S
        } else if (KIND == K_MOV_B32) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
            BODY8(S)
#undef S
        } else if (KIND == K_CVT_F32_I32) {
#define S(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(u[i]));
            BODY8(S)
#undef S
        } else if (KIND == K_ADD_F32) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g));
            BODY8(S)
#undef S
        } else if (KIND == K_PK_ADD_F32) {
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(g2));
            BODY8(S)
#undef S
        } else if (KIND == K_MIX_BLK_CVT) {      // a lane's 4 outputs of one block: convert the integer dot, x d_w, x d_x, add to the running sum
#define S(i) { float c0, c1, c2, c3; \
               asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(c0) : "v"(u[i])); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(c1) : "v"(u[(i + 1) & 7])); \
               asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(c2) : "v"(u[(i + 2) & 7])); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(c3) : "v"(u[(i + 3) & 7])); \
               asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c0) : "v"(g)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c1) : "v"(g)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c2) : "v"(g)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c3) : "v"(g)); \
               asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c0) : "v"(g)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c1) : "v"(g)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c2) : "v"(g)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c3) : "v"(g)); \
               asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c0)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(i + 1) & 7]) : "v"(c1)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(i + 2) & 7]) : "v"(c2)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(i + 3) & 7]) : "v"(c3)); }
            BODY8(S)
#undef S
        } else if (KIND == K_MIX_BLK_MAGIC) {    // the same 4 outputs when the matrix core's accumulator starts at 0x4B400000: (as_float(acc) - 12582912) is the conversion; all packed
#define S(i) { f2 c0, c1; \
               asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(c0) : "v"(p[i]), "v"(g2)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(c1) : "v"(p[(i + 1) & 7]), "v"(g2)); \
               asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(c0) : "v"(g2)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(c1) : "v"(g2)); \
               asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(c0) : "v"(g2)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(c1) : "v"(g2)); \
               asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i + 2) & 7]) : "v"(c0)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i + 3) & 7]) : "v"(c1)); }
            BODY8(S)
#undef S
        } else if (KIND == K_MIX_ATTN) {       // per chain: one packed multiply, two conversions, two adds into TWO sums (as the kernel: a[i] and a[(i+4)&7] are different accumulators)
#define S(i) { f2 pr; double c0, c1; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pr) : "v"(p[i]), "v"(g2)); \
               asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c0) : "v"(pr.x)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c1) : "v"(pr.y)); \
               asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c0)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c1)); }
            BODY8(S)
#undef S
        } else if (KIND == K_MIX_FMA) {
#define S(i) { f2 pr; double c0, c1; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pr) : "v"(p[i]), "v"(g2)); \
               asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c0) : "v"(pr.x)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c1) : "v"(pr.y)); \
               asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c0), "v"(one)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c1), "v"(one)); }
            BODY8(S)
#undef S
        } else if (KIND == K_MIX_SPLIT) {
#define S(i) { float p0, p1; double c0, c1; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(p[i].x), "v"(g)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(p[i].y), "v"(g)); \
               asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c0) : "v"(p0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c1) : "v"(p1)); \
               asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c0)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c1)); }
            BODY8(S)
#undef S
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += (float)a[i] + f[i] + p[i].x + p[i].y + (float)u[i];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(u64 *out, float *sink) {
    for (int wps = 1; wps <= 4; wps *= 2) {            // waves per SIMD: 256 compute units x (wps workgroups of 4 waves)
        const int nwg = 256 * wps;
        hipLaunchKernelGGL((k<KIND>), dim3(nwg), dim3(256), 0, 0, out, sink, 1.0f);
        hipDeviceSynchronize();
        hipLaunchKernelGGL((k<KIND>), dim3(nwg), dim3(256), 0, 0, out, sink, 1.0f);
        hipDeviceSynchronize();
        std::vector<u64> t((size_t)nwg * 4); hipMemcpy(t.data(), out, t.size() * 8, hipMemcpyDeviceToHost);
        std::sort(t.begin(), t.end());
        const double med = (double)t[t.size() / 2];
        printf("  %d wave%s per SIMD: %7.2f cycles per wave-instruction and SIMD (median wave: %.0f cycles for %d instructions)\n", wps, wps > 1 ? "s" : " ", med * 1.0 / (512.0 * kinstr[KIND]) , med, 512 * kinstr[KIND]);
    }
}

int main() {
    u64 *out; float *sink; hipMalloc((void **)&out, 256 * 4 * 4 * 8 * 2); hipMalloc((void **)&sink, 64);
    printf("# tools/microbench24: shader cycles between two instructions of ONE wave (so: x 1 / waves-per-SIMD = the SIMD's cycles per wave-instruction when the waves interleave perfectly)\n");
#define R(K) printf("%s\n", knames[K]); run<K>(out, sink);
    R(K_ADD_F64) R(K_FMA_F64) R(K_MUL_F64) R(K_CVT_F64_F32) R(K_CVT_F32_F64) R(K_PK_MUL_F32) R(K_MUL_F32) R(K_FMA_F32) R(K_PK_FMA_F32) R(K_ADD_U32) R(K_MOV_B32) R(K_CVT_F32_I32) R(K_ADD_F32) R(K_PK_ADD_F32) R(K_MIX_BLK_CVT) R(K_MIX_BLK_MAGIC) R(K_MIX_ATTN) R(K_MIX_FMA) R(K_MIX_SPLIT)
    return 0;
}
