// Round-4 check behind XP_H16 (kernels_xpipe.hip.h): is a 16-byte granule {3 words, tag}, written with ONE aligned 16-byte plain store by one lane and read with ONE
// 16-byte agent-scope (sc1) load by lanes of OTHER compute units of the same XCD, ever seen torn ?  One writer workgroup keeps rewriting 512 granules with
// {v, v ^ A, v ^ B, v} for v = 1, 2, ...; reader workgroups on the same XCD (HW_REG_XCC_ID) poll them and count every granule whose four words do not belong to one v.
// Also cross-XCD (write-through stores).  An observation, not a proof: the architecture does not promise single-copy atomicity beyond 8 bytes.
//   microbench19 [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));
constexpr u32 A = 0x9E3779B9u, B = 0x7F4A7C15u;

// ctl[0] stop flag, ctl[1] writer ticket, stat[0] reads, stat[1] torn, stat[2] distinct values seen (progress)
__global__ void k(v4u *gran, volatile u32 *ctl, u64 *stat, int writer_xcd, int same_xcd_readers, int sc1_stores) {
    __shared__ int s_role;
    const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);
    if (threadIdx.x == 0) {
        int role = -1;                                   // 0 writer, 1 reader
        if (xcc == writer_xcd && atomicAdd((u32 *)ctl + 1, 1u) == 0u) role = 0;
        else if ((xcc == writer_xcd) == (same_xcd_readers != 0)) role = 1;
        s_role = role;
    }
    __syncthreads();
    const int role = s_role;
    if (role < 0) return;
    const int t = threadIdx.x;                           // 512 threads: granule t
    if (role == 0) {
        for (u32 v = 1; ctl[0] == 0u; v++) {
            v4u g; g.x = v; g.y = v ^ A; g.z = v ^ B; g.w = v;
            if (sc1_stores) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(gran + t), "v"(g) : "memory");
            else gran[t] = g;
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)gran, 0, 512 * 16, 0x00027000);
    u64 reads = 0, torn = 0, changes = 0;
    u32 last = 0;
    while (ctl[0] == 0u) {
        for (int i = 0; i < 64; i++) {
            const v4u g = __builtin_amdgcn_raw_buffer_load_b128(rs, t * 16, 0, 16);      // sc1
            reads++;
            if (g.w != g.x || g.y != (g.x ^ A) || g.z != (g.x ^ B)) { if (g.x | g.y | g.z | g.w) torn++; }
            if (g.x != last) { changes++; last = g.x; }
        }
    }
    atomicAdd(stat + 0, reads); atomicAdd(stat + 1, torn); atomicAdd(stat + 2, changes);
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    v4u *gran; u32 *ctl; u64 *stat;
    CK(hipMalloc(&gran, 512 * 16));
    CK(hipHostMalloc((void **)&ctl, 64, hipHostMallocDefault));
    CK(hipMalloc(&stat, 64));
    for (int mode = 0; mode < 3; mode++) {      // 0: same XCD, plain stores; 1: other XCDs, write-through stores; 2: same XCD, write-through stores
        CK(hipMemset(gran, 0, 512 * 16)); CK(hipMemset(stat, 0, 64));
        ctl[0] = 0; ctl[1] = 0;
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, gran, ctl, stat, 2, mode != 1, mode != 0);
        const auto t0 = clock();
        while ((double)(clock() - t0) / CLOCKS_PER_SEC < secs) {}
        ctl[0] = 1;
        CK(hipDeviceSynchronize());
        u64 h[3];
        CK(hipMemcpy(h, stat, 24, hipMemcpyDeviceToHost));
        printf("%s: %.3g granule reads, %llu value changes seen, %llu torn\n",
               mode == 0 ? "same XCD, plain 16-byte stores, sc1 16-byte loads    " : mode == 1 ? "other XCDs, write-through 16-byte stores, sc1 loads " : "same XCD, write-through 16-byte stores, sc1 loads  ",
               (double)h[0], h[2], h[1]);
    }
    return 0;
}
