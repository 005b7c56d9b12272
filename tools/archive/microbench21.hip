// Can a compute unit stream weights and hand activations over at the same time?  (round 4, kernels_xcols.hip.h: a wave's vector loads return in order, so a poll behind a
// weight request waits for the fabric; would waves that ONLY poll, beside waves that ONLY stream, keep the hand-offs at their 0.4 us?)
// 256 workgroups x 512 threads; inside every XCD workgroups 2k and 2k + 1 play ping-pong over 64 tagged 8-byte granules (wave 0: plain stores, agent-scope polls -- the in-XCD
// hand-off of kernels_xpipe.hip.h), R round trips, timed on the 100 MHz clock.  Arms:
//   0  nothing else runs
//   1  waves 4 .. 7 of every workgroup stream a buffer the whole time (16-byte loads, 8 per lane in flight): the polling wave has no load of its own
//   2  the same with waves 1 .. 7 streaming
//   3  nobody streams, but the polling wave itself issues ONE 16-byte fabric load per lane in front of every poll pass (the in-order penalty itself)
//   hipcc --offload-arch=gfx950 -O3 -o microbench21 microbench21.hip && ./microbench21
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ u64 wall() { return __builtin_readcyclecounter() * 0 + __builtin_amdgcn_s_memrealtime(); }

__global__ __launch_bounds__(512) void k(u64 *gran, const u4v *stream, size_t stream16, uint32_t *tickets, u64 *ticks, uint32_t *sink, int arm, int rounds) {
    __shared__ int s_info[2];
    __shared__ volatile int s_stop;
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;
        s_info[0] = (int)xcc; s_info[1] = (int)(atomicAdd(tickets + xcc, 1u) & 31u);
        s_stop = 0;
    }
    __syncthreads();
    const int xcd = s_info[0], slot = s_info[1], pair = slot >> 1, me = slot & 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool streamer = (arm == 1 && wave >= 4) || (arm == 2 && wave >= 1);
    if (wave == 0) {
        u64 *ping = gran + ((size_t)(xcd * 16 + pair) * 2 + 0) * 64 + lane, *pong = gran + ((size_t)(xcd * 16 + pair) * 2 + 1) * 64 + lane;
        const u4v *sp = stream + ((size_t)(xcd * 32 + slot) * 65536 + lane) % stream16;
        uint32_t acc = 0;
        const u64 t0 = wall();
        for (int r = 1; r <= rounds; r++) {
            if (me == 0) *ping = ((u64)r << 32) | (u64)lane;      // plain store: the line stays in the XCD's L2
            u64 *w = me == 0 ? pong : ping;
            for (uint32_t spin = 0; spin < 4000000u; spin++) {      // bounded: a lost partner must not hang the device
                if (arm == 3) { const u4v v = __builtin_nontemporal_load(sp + (size_t)((r * 64) & 0xffff)); acc ^= v.x; }
                const u64 a = __hip_atomic_load(w, RLX);
                if (__all((uint32_t)(a >> 32) == (uint32_t)r)) break;
            }
            if (me == 1) *pong = ((u64)r << 32) | (u64)lane;
        }
        const u64 t1 = wall();
        if (lane == 0) { ticks[xcd * 16 + pair] = t1 - t0; s_stop = 1; if (acc == 0x9e3779b9u) sink[0] = acc; }
    } else if (streamer) {
        uint32_t a = 0;
        size_t i = ((size_t)(xcd * 32 + slot) * 8 + wave) * 4096 + lane;
        while (!s_stop) {
            u4v v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = stream[(i + (size_t)q * 64) % stream16];
#pragma unroll
            for (int q = 0; q < 8; q++) a ^= v[q].x ^ v[q].w;
            i += 512;
        }
        if (a == 0x9e3779b9u) sink[1] = a;
    }
}
int main() {
    const size_t sbytes = (size_t)1 << 30;
    u64 *gran, *ticks; u4v *stream; uint32_t *tickets, *sink;
    hipMalloc((void **)&gran, 8 * 16 * 2 * 64 * 8); hipMalloc((void **)&ticks, 128 * 8); hipMalloc((void **)&stream, sbytes); hipMemset(stream, 1, sbytes);
    hipMalloc((void **)&tickets, 64); hipMalloc((void **)&sink, 64);
    const int rounds = 2000;
    const char *names[4] = {"nothing else runs", "waves 4 .. 7 stream (the polling wave has no load of its own)", "waves 1 .. 7 stream", "the polling wave issues one fabric load per lane in front of every poll pass"};
    for (int arm = 0; arm < 4; arm++) {
        double best = 1e30, worst = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(gran, 0, 8 * 16 * 2 * 64 * 8); hipMemset(tickets, 0, 64); hipMemset(ticks, 0, 128 * 8);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, gran, stream, sbytes / 16, tickets, ticks, sink, arm, rounds);
            hipDeviceSynchronize();
            u64 h[128]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
            double sum = 0; int n = 0;
            for (int i = 0; i < 128; i++) if (h[i]) { const double us = h[i] * 0.01 / rounds; sum += us; n++; if (us > worst) worst = us; }
            if (n && sum / n < best) best = sum / n;
        }
        printf("arm %d  %-78s round trip %6.3f us (one way %5.3f), slowest pair %6.3f\n", arm, names[arm], best, best / 2, worst);
    }
    return 0;
}
