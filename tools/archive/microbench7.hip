// Round-2 calibration, part 4: cost of a device-wide barrier built from per-workgroup FLAGS (no atomics): every
// workgroup stores its own flag write-through (sc1), one wave per workgroup polls ALL flags with 16-byte sc1 loads
// until every flag has reached the epoch.  Every spin is bounded (give-up word) so a non-resident workgroup cannot
// hang the box.  Also: the same barrier carrying a 4 KB "activation" exchange (each workgroup publishes 16 B of it,
// everybody reads all 4 KB after the barrier) -- the pattern of a persistent decode step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint4 load16_sc1(const void *p) {
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// flags: one unsigned per workgroup, G <= 256 -> 64 lanes x 4 flags
__device__ __forceinline__ bool flag_barrier(unsigned *flags, int G, unsigned epoch, unsigned *giveup) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned spins = 0;
        for (;;) {
            bool all = true;
            if (lane * 4 < G) {
                const uint4 f = load16_sc1(flags + lane * 4);
                all = (int)(f.x - epoch) >= 0 && (lane * 4 + 1 >= G || (int)(f.y - epoch) >= 0) && (lane * 4 + 2 >= G || (int)(f.z - epoch) >= 0) &&
                      (lane * 4 + 3 >= G || (int)(f.w - epoch) >= 0);
            }
            if (__all(all)) break;
            if (++spins > (1u << 18) || __hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                if (lane == 0) __hip_atomic_store(giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void k_barriers(unsigned *flags, unsigned *giveup, int G, int rounds, unsigned base, float *vec, int payload, int *bad) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = 1;
    int nb = 0;
    for (int r = 1; r <= rounds; r++) {
        if (payload) {
            // publish this workgroup's 4 floats of round r (write-through), barrier, read all 1024 floats back
            if (threadIdx.x < 4) __hip_atomic_store(vec + (r & 1) * 1024 + blockIdx.x * 4 + threadIdx.x, (float)(base + r) + 0.001f * (blockIdx.x * 4 + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!flag_barrier(flags, G, base + r, giveup)) { if (threadIdx.x == 0) s_ok = 0; }
        __syncthreads();
        if (!s_ok) break;
        if (payload) {
            const uint4 v = load16_sc1(vec + (r & 1) * 1024 + threadIdx.x * 4);
            const float e0 = (float)(base + r) + 0.001f * (threadIdx.x * 4);
            if (threadIdx.x * 4 < G * 4 && __uint_as_float(v.x) != e0) nb++;
        }
    }
    if (nb) atomicAdd(bad, nb);
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *flags, *giveup; float *vec; int *bad;
    CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&giveup, 64)); CK(hipMalloc(&vec, 2 * 1024 * 4)); CK(hipMalloc(&bad, 64));
    CK(hipMemset(flags, 0, 4096)); CK(hipMemset(giveup, 0, 64)); CK(hipMemset(bad, 0, 64)); CK(hipMemset(vec, 0, 8192));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned base = 0;
    for (int payload = 0; payload < 2; payload++)
        for (int G : {32, 64, 128, 256}) {
            const int rounds = 500;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(k_barriers, G, 256, 0, st, flags, giveup, G, rounds, base, vec, payload, bad);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                base += rounds;
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned gu; int hb; CK(hipMemcpy(&gu, giveup, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
                if (rep == 2) printf("G=%3d workgroups, payload %d: %.2f us per barrier round (give-up %u, stale payload words %d)\n", G, payload, ms * 1e3f / rounds, gu, hb);
                if (gu) { printf("barrier gave up -- stopping\n"); return 1; }
            }
        }
    return 0;
}
