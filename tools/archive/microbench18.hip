// Round-4 calibration: what does ONE cross-XCD hand-off cost as a function of (producer XCD, consumer XCD, ADDRESS of the granule) ?
//
// profiles/xpipe_timeline_r3.txt: the x hop (odd XCD -> next even XCD) costs 1.23 - 1.38 us, the x1 hop (even -> odd XCD of the same pair) 0.81 - 0.87 us, same
// 1024 granules, same stores, same sweep.  MI355X = 4 I/O dies x 2 XCDs x 2 HBM stacks; device memory is interleaved over all stacks.  If a granule's latency
// depends on WHERE its line lives relative to producer and consumer, a hand-off of 8 KB (several interleave units) completes with its slowest line, and choosing
// lines near the consumer would shorten every hop.  This measures it: a ping-pong between ONE lane on XCD a and ONE lane on XCD b over the granule at byte offset
// off (ping) and off + 8 (pong), write-through (sc1) stores, sc1 polls without sleep; one-way time = round trips / 2.
//
//   microbench18 [reps]   -> one line per (a, b): the one-way ns for every offset (128 x 256 B steps, then 128 x 4 KB steps), plus a summary
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr u32 SPIN_MAX = 2000000;

__device__ __forceinline__ u64 wall() { return wall_clock64(); }

// ctl[0], ctl[1]: role tickets; ctl[2]: error
__global__ void pingpong(u64 *buf, int a, int b, const int *offs, int n_offs, int reps, u32 *ctl, u64 *out, u32 launch, int local_store) {
    __shared__ int s_role;
    if (threadIdx.x == 0) {
        const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);
        int role = -1;
        if (a != b) {
            if (xcc == a && __hip_atomic_fetch_add(ctl + 0, 1u, RLX_AGENT) == 0u) role = 0;
            else if (xcc == b && __hip_atomic_fetch_add(ctl + 1, 1u, RLX_AGENT) == 0u) role = 1;
        } else if (xcc == a) {
            const u32 t = __hip_atomic_fetch_add(ctl + 0, 1u, RLX_AGENT);
            if (t < 2u) role = (int)t;
        }
        s_role = role;
    }
    __syncthreads();
    const int role = s_role;
    if (role < 0 || threadIdx.x != 0) return;
    for (int i = 0; i < n_offs; i++) {
        u64 *X = buf + offs[i] / 8, *Y = X + 1;
        const u64 base = ((u64)launch << 40) | ((u64)i << 20);
        if (role == 0) {
            const u64 t0 = wall();
            for (int r = 1; r <= reps; r++) {
                if (local_store) __hip_atomic_store(X, base + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(X, base + r, RLX_AGENT);
                u32 spins = 0;
                while (__hip_atomic_load(Y, RLX_AGENT) != base + r) { if (++spins > SPIN_MAX) { __hip_atomic_store(ctl + 2, 1u, RLX_AGENT); return; } }
            }
            out[i] = wall() - t0;
        } else {
            for (int r = 1; r <= reps; r++) {
                u32 spins = 0;
                while (__hip_atomic_load(X, RLX_AGENT) != base + r) { if (++spins > SPIN_MAX) { __hip_atomic_store(ctl + 2, 2u, RLX_AGENT); return; } }
                if (local_store) __hip_atomic_store(Y, base + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(Y, base + r, RLX_AGENT);
            }
        }
    }
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 32;
    std::vector<int> offs;
    for (int k = 0; k < 128; k++) offs.push_back(k * 256);
    for (int k = 0; k < 128; k++) offs.push_back(32768 + k * 4096);
    const int n = (int)offs.size();
    const size_t bytes = 32768 + 128 * 4096 + 4096;
    u64 *buf; int *d_offs; u32 *ctl; u64 *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    CK(hipMalloc(&d_offs, n * 4)); CK(hipMemcpy(d_offs, offs.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ctl, 64)); CK(hipMalloc(&out, n * 8));
    std::vector<u64> h(n);
    u32 launch = 1;
    printf("# one-way hand-off time in ns (100 MHz clock, %d round trips per offset); columns: 128 offsets in 256-byte steps, then 128 in 4-KB steps\n", reps);
    std::vector<double> summary;
    for (int local = 0; local < 2; local++)
        for (int a = 0; a < 8; a++)
            for (int b = 0; b < 8; b++) {
                if (local && a != b) continue;
                CK(hipMemset(ctl, 0, 64)); CK(hipMemset(out, 0, n * 8));
                hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), 0, 0, buf, a, b, d_offs, n, reps, ctl, out, launch++, local);
                CK(hipDeviceSynchronize());
                u32 hc[3];
                CK(hipMemcpy(hc, ctl, 12, hipMemcpyDeviceToHost));
                CK(hipMemcpy(h.data(), out, n * 8, hipMemcpyDeviceToHost));
                if (hc[2]) { printf("%d -> %d: error %u\n", a, b, hc[2]); continue; }
                std::vector<double> ns(n);
                for (int i = 0; i < n; i++) ns[i] = (double)h[i] * 10.0 / (2.0 * reps);
                std::vector<double> s(ns);
                std::sort(s.begin(), s.end());
                printf("%s %d -> %d  min %.0f  p25 %.0f  median %.0f  p75 %.0f  max %.0f :", local ? "plain-store" : "sc1-store", a, b, s[0], s[n / 4], s[n / 2], s[3 * n / 4], s[n - 1]);
                for (int i = 0; i < n; i++) printf(" %.0f", ns[i]);
                printf("\n");
            }
    return 0;
}
