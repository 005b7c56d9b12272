// Hand-off floor of an ALL-compute-unit persistent decode layer (the shape a float-weight pipeline would need: 50 MB of F32 weights per layer cannot be stationary on one
// XCD): 256 workgroups x 512 threads, five all-to-all hand-offs per layer through {value, tag} granules with write-through stores and L2-bypassing polls --
//   x[1024] -> (12 q/k/v rows per workgroup) qkv[3072] -> (16 head workgroups) att[1024] -> (4 rows) x1[1024] -> (16 rows) h[4096] -> (4 rows) x of the next layer
// -- trivial bodies, optionally with the layer's weight stream (192 KB per workgroup and layer, a quarter behind each publish) kept in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int NL = 24, G_X = 0, G_QKV = 1024, G_ATT = 4096, G_X1 = 5120, G_H = 6144, G_LAYER = 10240;
__device__ __forceinline__ void put(u64 *g, uint32_t tag, uint32_t v) { __hip_atomic_store(g, ((u64)tag << 32) | v, RLX); }
template <int N>
__device__ __forceinline__ uint32_t sweep(const u64 *g, int stride, bool active, uint32_t tag) {
    uint32_t acc = 0;
    for (;;) {
        bool ok = true;
        acc = 0;
        if (active) {
#pragma unroll
            for (int k = 0; k < N; k++) { const u64 a = __hip_atomic_load(g + k * stride, RLX); ok &= (uint32_t)(a >> 32) == tag; acc += (uint32_t)a; }
        }
        if (__all(ok)) return acc;
        __builtin_amdgcn_s_sleep(1);
    }
}
__global__ __launch_bounds__(512) void layers(u64 *gran, const uint4 *weights, size_t wstride, uint32_t tag, int stream_kb, uint32_t *sink, unsigned long long *t_out) {
    __shared__ uint32_t s_acc[8];
    const int g = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const unsigned long long t0 = wall_clock64();
    if (tid < 4) put(gran + G_X + 4 * g + tid, tag, 1u);      // layer 0's input
    uint4 w[6];
    uint32_t keep = 0;
    const int per_q = stream_kb * 1024 / 4 / 16 / 512;          // 16-byte loads per thread and quarter (192 KB -> 6)
    auto stream = [&](int L, int q) {
        const uint4 *base = weights + ((size_t)L * 256 + g) * wstride + (size_t)q * per_q * 512 + tid;
        for (int i = 0; i < per_q && i < 6; i++) { const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(base + (size_t)i * 512)); w[i] = make_uint4(t.x, t.y, t.z, t.w); }
    };
    auto eat = [&]() { for (int i = 0; i < 6; i++) keep ^= w[i].x ^ w[i].w; };
    for (int L = 0; L < NL; L++) {
        u64 *G = gran + (size_t)L * G_LAYER;
        uint32_t a = sweep<4>(G + G_X + tid, 256, tid < 256, tag);                    // x: 4 waves x 4 granules
        if (tid % 64 == 0) s_acc[wave] = a;
        __syncthreads();
        if (stream_kb) eat();
        if (tid < 12) put(G + G_QKV + 12 * g + tid, tag, s_acc[0] + 1u);
        if (stream_kb) stream(L, 0);
        if ((g & 15) == 0) {                                                          // 16 head workgroups
            a = sweep<1>(G + G_QKV + (g >> 4) * 192 + tid, 1, tid < 192, tag);
            if (tid % 64 == 0) s_acc[wave] = a;
            __syncthreads();
            if (tid < 64) put(G + G_ATT + (g >> 4) * 64 + tid, tag, s_acc[0]);
        }
        a = sweep<2>(G + G_ATT + tid, 512, true, tag);
        if (tid % 64 == 0) s_acc[wave] = a;
        __syncthreads();
        if (stream_kb) eat();
        if (tid < 4) put(G + G_X1 + 4 * g + tid, tag, s_acc[1]);
        if (stream_kb) stream(L, 1);
        a = sweep<4>(G + G_X1 + tid, 256, tid < 256, tag);
        if (tid % 64 == 0) s_acc[wave] = a;
        __syncthreads();
        if (stream_kb) eat();
        if (tid < 16) put(G + G_H + 16 * g + tid, tag, s_acc[2]);
        if (stream_kb) stream(L, 2);
        a = sweep<8>(G + G_H + tid, 512, true, tag);
        if (tid % 64 == 0) s_acc[wave] = a;
        __syncthreads();
        if (stream_kb) eat();
        if (L + 1 < NL && tid < 4) put(gran + (size_t)(L + 1) * G_LAYER + G_X + 4 * g + tid, tag, s_acc[3]);
        if (stream_kb) stream(L, 3);
        __syncthreads();
    }
    if (stream_kb) eat();
    if (keep == 0x12345u) sink[0] = keep;
    if (tid == 0) t_out[g] = wall_clock64() - t0;
}
int main() {
    u64 *gran; uint4 *weights; uint32_t *sink; unsigned long long *t_out;
    const size_t wstride = 192 * 1024 / 16;                       // uint4 per (layer, workgroup)
    hipMalloc((void **)&gran, (size_t)NL * G_LAYER * 8); hipMemset(gran, 0, (size_t)NL * G_LAYER * 8);
    hipMalloc((void **)&weights, (size_t)NL * 256 * wstride * 16); hipMemset(weights, 1, (size_t)NL * 256 * wstride * 16);      // 1.2 GB: beyond L2 + Infinity Cache
    hipMalloc((void **)&sink, 64); hipMalloc((void **)&t_out, 256 * 8);
    for (int kb : {0, 96, 192}) {
        uint32_t tag = 1;
        double best = 1e9, worst = 0;
        for (int rep = 0; rep < 12; rep++, tag++) {
            hipLaunchKernelGGL(layers, dim3(256), dim3(512), 0, 0, gran, weights, wstride, tag + 100u * (uint32_t)kb, kb, sink, t_out);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<unsigned long long> t(256);
            hipMemcpy(t.data(), t_out, 256 * 8, hipMemcpyDeviceToHost);
            unsigned long long mx = 0; for (auto v : t) mx = v > mx ? v : mx;
            const double us = mx * 0.01 / NL;
            if (rep >= 2) { best = us < best ? us : best; worst = us > worst ? us : worst; }
        }
        printf("weight stream %3d KB per workgroup and layer (%4.1f MB per layer): %.2f .. %.2f us per layer (five hand-offs, trivial bodies)%s\n", kb, kb * 256 / 1024.0, best, worst,
               kb ? "" : "");
    }
    return 0;
}
