// Host <-> device ping-pong through pinned host memory, as the resident decode launch uses it: the host writes a sequence number, ONE lane of a
// resident kernel polls it with system-scope loads and answers into a second pinned word; the host spins on the answer.  Round trip = mailbox read + completion write.
// Variant B: the answer is preceded by a 1 KB row of 16-byte write-through stores + s_waitcnt vmcnt(0), as an lm_head workgroup does it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void pong(volatile uint32_t *mbox, uint32_t *ack, float *row, int n, int with_row) {
    const int lane = threadIdx.x;
    for (uint32_t s = 1; s <= (uint32_t)n; s++) {
        if (lane == 0) while (__hip_atomic_load((uint32_t *)mbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != s) {}
        __builtin_amdgcn_s_barrier();
        if (with_row) {
            v4f v = {(float)s, 1.f, 2.f, 3.f};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(row + 4 * lane), "v"(v) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (lane == 0) __hip_atomic_store(ack, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int main() {
    uint32_t *mbox, *ack; float *row;
    hipHostMalloc((void **)&mbox, 64, hipHostMallocDefault); hipHostMalloc((void **)&ack, 64, hipHostMallocDefault); hipHostMalloc((void **)&row, 4096, hipHostMallocDefault);
    for (int with_row = 0; with_row < 2; with_row++) {
        const int n = 20000;
        *mbox = 0; *ack = 0;
        hipLaunchKernelGGL(pong, dim3(1), dim3(64), 0, 0, mbox, ack, row, n, with_row);
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t s = 1; s <= (uint32_t)n; s++) {
            __atomic_store_n(mbox, s, __ATOMIC_RELEASE);
            while (*(volatile uint32_t *)ack != s) {}
        }
        const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / n;
        hipDeviceSynchronize();
        printf("round trip host -> device -> host%s: %.2f us\n", with_row ? " (+ 1 KB row, store wait)" : "", us);
    }
    return 0;
}
