// microbench13's ping-pong with the host -> device word in DEVICE memory, written by the CPU through the PCIe BAR (large-BAR systems map VRAM into the
// process), polled by the kernel with agent-scope loads (no PCIe read): is it there, and what does a round trip cost ?  (the mailbox of the
// resident launch).  Variant C additionally streams 170 KB of 16-byte write-through stores to pinned host memory in front of every answer from 166 other
// workgroups... kept simple: ONE workgroup writes 16 KB, to see a read (pinned mailbox) queue behind posted writes and a BAR mailbox not.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <csetjmp>
#include <csignal>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void pong(const uint32_t *mbox, int mbox_dev, uint32_t *ack, float *row, int n, int row_kb) {
    const int lane = threadIdx.x;
    for (uint32_t s = 1; s <= (uint32_t)n; s++) {
        if (lane == 0) {
            if (mbox_dev) while (__hip_atomic_load(mbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != s) {}
            else while (__hip_atomic_load(mbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != s) {}
        }
        __builtin_amdgcn_s_barrier();
        for (int k = 0; k < row_kb; k++) {
            v4f v = {(float)s, 1.f, 2.f, 3.f};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(row + 256 * k + 4 * lane), "v"(v) : "memory");
        }
        if (lane == 0) __hip_atomic_store(ack, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // behind the row stores, not waiting for them
    }
}
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("isLargeBar = %d\n", prop.isLargeBar);
    uint32_t *mbox_h, *mbox_d = nullptr, *mbox_f = nullptr, *ack; float *row;
    hipHostMalloc((void **)&mbox_h, 64, hipHostMallocDefault); hipHostMalloc((void **)&ack, 64, hipHostMallocDefault); hipHostMalloc((void **)&row, 1 << 20, hipHostMallocDefault);
    hipMalloc((void **)&mbox_d, 4096); hipMemset(mbox_d, 0, 4096);
    if (hipExtMallocWithFlags((void **)&mbox_f, 4096, hipDeviceMallocFinegrained) != hipSuccess) mbox_f = nullptr; else hipMemset(mbox_f, 0, 4096);
    hipDeviceSynchronize();
    struct { const char *name; uint32_t *p; int dev; } boxes[3] = {{"pinned host word", mbox_h, 0}, {"hipMalloc word through the BAR", mbox_d, 1}, {"fine-grained device word through the BAR", mbox_f, 1}};
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    for (auto &b : boxes) {
        if (!b.p) { printf("%s: allocation failed\n", b.name); continue; }
        if (sigsetjmp(jb, 1)) { printf("%s: the CPU cannot write it (fault)\n", b.name); continue; }
        *(volatile uint32_t *)b.p = 0;      // faults here if VRAM is not mapped
        for (int row_kb : {0, 1, 16, 170}) {
            const int n = 5000;
            *(volatile uint32_t *)b.p = 0; *ack = 0;
            __atomic_thread_fence(__ATOMIC_SEQ_CST);
            hipLaunchKernelGGL(pong, dim3(1), dim3(64), 0, 0, b.p, b.dev, ack, row, n, row_kb);
            const auto t0 = std::chrono::steady_clock::now();
            bool dead = false;
            for (uint32_t s = 1; s <= (uint32_t)n && !dead; s++) {
                __atomic_store_n(b.p, s, __ATOMIC_RELEASE);
                if (b.dev) __builtin_ia32_sfence();
                const auto ts = std::chrono::steady_clock::now();
                for (uint32_t spin = 0; *(volatile uint32_t *)ack != s; spin++)
                    if ((spin & 0xfffff) == 0xfffff && std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count() > 2.0) { dead = true; break; }
            }
            const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / n;
            if (dead) { printf("%s, %3d KB row: no answer (the kernel does not see the CPU's store)\n", b.name, row_kb); *(volatile uint32_t *)b.p = 0xffffffffu; return 1; }
            hipDeviceSynchronize();
            printf("%s, %3d KB row in front of the answer: round trip %.2f us\n", b.name, row_kb, us);
        }
    }
    return 0;
}
