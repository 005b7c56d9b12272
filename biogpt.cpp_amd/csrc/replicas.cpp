// Single-process multi-GPU replicas behind the C-ABI (BASELINE north_star: "independent prompts sharded
// across the 8 GPUs of one node via RCCL broadcast of the weights over xGMI, no tensor parallelism, host code stays C++").
//
// Reference call replaced: the ONE biogpt_model_load at examples/main/main.cpp:38 (a C++ caller gets one device there).
// Here: the file is read and repacked once (device devices[0]), the packed weight arena -- one contiguous allocation
// whose layout depends on the 7 header ints only (engine.hip: plan_arena) -- goes to the other devices with ONE
// ncclBroadcast inside a ncclCommInitAll communicator (xGMI is point-to-point: each peer has its own link to the root),
// every device gets its own context (KV cache, scratch, stream, graphs) attached to its copy, and prompt g is served by
// replica g mod n, each replica driven by its own host thread.  No collective on the data path.
//
// RCCL is resolved at run time (dlopen "librccl.so"): the library has no link-time dependency on it, and a process that
// already carries a RCCL (PyTorch ships its own) keeps using that one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/biogpt_hip.h"
#include "host_common.h"

using namespace bg;

namespace {

// the five RCCL entry points used, with rccl.h's signatures (ncclComm_t is an opaque pointer; ncclUint8 = 1)
typedef void *nccl_comm_t;
typedef int (*fn_comm_init_all)(nccl_comm_t *, int, const int *);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef int (*fn_broadcast)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_group)(void);
typedef const char *(*fn_err)(int);

struct Rccl {
    void *lib = nullptr;
    fn_comm_init_all comm_init_all = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_broadcast broadcast = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_err err = nullptr;
    bool open() {
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        comm_init_all = (fn_comm_init_all)dlsym(lib, "ncclCommInitAll");
        comm_destroy = (fn_comm_destroy)dlsym(lib, "ncclCommDestroy");
        broadcast = (fn_broadcast)dlsym(lib, "ncclBroadcast");
        group_start = (fn_group)dlsym(lib, "ncclGroupStart");
        group_end = (fn_group)dlsym(lib, "ncclGroupEnd");
        err = (fn_err)dlsym(lib, "ncclGetErrorString");
        return comm_init_all && comm_destroy && broadcast && group_start && group_end && err;
    }
};

}  // namespace

struct biogpt_hip_replicas {
    std::vector<int> devices;
    std::vector<biogpt_hip_ctx *> ctx;
    std::vector<void *> arenas;     // arenas of the replicas 1 .. n-1 (owned here; replica 0 owns its own)
    double broadcast_seconds = 0.0;
    size_t arena_bytes = 0;
};

#define HIP_TRY_R(ret, expr)                                                                \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) BG_FAIL(ret, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

extern "C" {

void biogpt_hip_replicas_free(biogpt_hip_replicas *r) {
    if (!r) return;
    for (biogpt_hip_ctx *c : r->ctx) biogpt_hip_free(c);
    for (size_t i = 0; i < r->arenas.size(); i++)
        if (r->arenas[i]) { (void)hipSetDevice(r->devices[i + 1]); (void)hipFree(r->arenas[i]); }
    delete r;
}

biogpt_hip_replicas *biogpt_hip_replicas_load(const char *fname, const int *devices, int n_devices, int verbosity) {
    clear_error();
    if (!fname || !devices || n_devices < 1) BG_FAIL(nullptr, "null argument");
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) BG_FAIL(nullptr, "device %d listed twice", devices[i]);
    int n_found = 0;
    if (hipGetDeviceCount(&n_found) != hipSuccess || n_found <= 0) { (void)hipGetLastError(); BG_FAIL(nullptr, "no HIP device available: this engine has no CPU fallback"); }
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= n_found) BG_FAIL(nullptr, "HIP device %d out of range (found %d)", devices[i], n_found);
    std::unique_ptr<biogpt_hip_replicas, void (*)(biogpt_hip_replicas *)> r(new biogpt_hip_replicas(), biogpt_hip_replicas_free);
    r->devices.assign(devices, devices + n_devices);
    biogpt_hip_ctx *root = biogpt_hip_load(fname, devices[0], verbosity);
    if (!root) return nullptr;
    r->ctx.push_back(root);
    biogpt_hip_hparams hp;
    biogpt_hip_get_hparams(root, &hp);
    r->arena_bytes = biogpt_hip_arena_bytes(root);
    if (biogpt_hip_n_tensors(root) == 0) BG_FAIL(nullptr, "empty model (no tensors): nothing to replicate");

    Rccl rccl;
    if (!rccl.open()) BG_FAIL(nullptr, "librccl.so not found or incomplete (%s): multi-GPU replicas need RCCL", dlerror() ? dlerror() : "missing symbols");
    r->arenas.assign((size_t)n_devices - 1, nullptr);
    // streams and communicators live only for the broadcast: released on EVERY way out of this function (the arenas belong to r)
    struct Transient {
        const Rccl &rccl; const std::vector<int> &dev;
        std::vector<hipStream_t> streams; std::vector<nccl_comm_t> comms;
        ~Transient() {
            for (size_t i = 0; i < dev.size(); i++) {
                (void)hipSetDevice(dev[i]);
                if (comms[i]) (void)rccl.comm_destroy(comms[i]);
                if (streams[i]) (void)hipStreamDestroy(streams[i]);
            }
        }
    } tr{rccl, r->devices, std::vector<hipStream_t>((size_t)n_devices, nullptr), std::vector<nccl_comm_t>((size_t)n_devices, nullptr)};
    std::vector<hipStream_t> &streams = tr.streams;
    std::vector<nccl_comm_t> &comms = tr.comms;
    for (int i = 0; i < n_devices; i++) {
        HIP_TRY_R(nullptr, hipSetDevice(devices[i]));
        HIP_TRY_R(nullptr, hipStreamCreateWithFlags(&streams[(size_t)i], hipStreamNonBlocking));
        if (i > 0) HIP_TRY_R(nullptr, hipMalloc(&r->arenas[(size_t)i - 1], r->arena_bytes));
    }
    int rc = rccl.comm_init_all(comms.data(), n_devices, devices);
    if (rc != 0) BG_FAIL(nullptr, "ncclCommInitAll failed: %s", rccl.err(rc));
    const auto t0 = std::chrono::steady_clock::now();
    rc = rccl.group_start();
    for (int i = 0; i < n_devices && rc == 0; i++) {
        void *buf = (i == 0) ? biogpt_hip_arena_ptr(root) : r->arenas[(size_t)i - 1];
        rc = rccl.broadcast(buf, buf, r->arena_bytes, /*ncclUint8*/ 1, /*root*/ 0, comms[(size_t)i], streams[(size_t)i]);
    }
    const int rc_end = rccl.group_end();
    if (rc == 0) rc = rc_end;
    hipError_t sync_err = hipSuccess;
    int sync_dev = -1;
    for (int i = 0; i < n_devices; i++) {
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipStreamSynchronize(streams[(size_t)i]);
        if (e != hipSuccess && sync_err == hipSuccess) { sync_err = e; sync_dev = devices[i]; }
    }
    r->broadcast_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != 0 && sync_err != hipSuccess)
        BG_FAIL(nullptr, "ncclBroadcast of the weight arena failed: %s; and the stream of device %d: %s", rccl.err(rc), sync_dev, hipGetErrorString(sync_err));
    if (sync_err != hipSuccess) BG_FAIL(nullptr, "the broadcast stream of device %d failed: %s", sync_dev, hipGetErrorString(sync_err));
    if (rc != 0) BG_FAIL(nullptr, "ncclBroadcast of the weight arena failed: %s", rccl.err(rc));
    if (verbosity > 0)
        fprintf(stderr, "biogpt_hip_replicas_load: %.1f MiB arena broadcast to %d device(s) in %.2f ms\n", r->arena_bytes / 1048576.0, n_devices,
                r->broadcast_seconds * 1e3);
    for (int i = 1; i < n_devices; i++) {
        biogpt_hip_ctx *c = biogpt_hip_attach(&hp, devices[i], r->arenas[(size_t)i - 1], r->arena_bytes);
        if (!c) return nullptr;
        r->ctx.push_back(c);
        if (biogpt_hip_share_vocab(c, root) != 0) return nullptr;
    }
    return r.release();
}

int biogpt_hip_replicas_count(const biogpt_hip_replicas *r) { return r ? (int)r->ctx.size() : -1; }
biogpt_hip_ctx *biogpt_hip_replicas_ctx(biogpt_hip_replicas *r, int i) {
    if (!r || i < 0 || (size_t)i >= r->ctx.size()) return nullptr;
    return r->ctx[(size_t)i];
}
double biogpt_hip_replicas_broadcast_seconds(const biogpt_hip_replicas *r) { return r ? r->broadcast_seconds : -1.0; }

int biogpt_hip_replicas_generate_greedy(biogpt_hip_replicas *r, const int32_t *prompts, const int32_t *prompt_lens, int32_t n_prompts,
                                        int32_t n_batch, int32_t n_predict, int32_t *out_ids, int32_t *out_counts, double *seconds_out) {
    clear_error();
    if (!r || !prompts || !prompt_lens || !out_ids || n_prompts < 1 || n_predict < 1) BG_FAIL(-1, "bad argument");
    const int n = (int)r->ctx.size();
    std::vector<size_t> offs((size_t)n_prompts + 1, 0);
    for (int g = 0; g < n_prompts; g++) offs[(size_t)g + 1] = offs[(size_t)g] + (size_t)std::max(0, prompt_lens[g]);
    std::vector<int> status((size_t)n, 0);
    std::vector<std::string> errs((size_t)n);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int i = 0; i < n; i++) {
        pool.emplace_back([&, i] {   // one host thread per device; prompts i, i + n, i + 2n, ...
            for (int g = i; g < n_prompts; g += n) {
                const int got = biogpt_hip_generate_greedy(r->ctx[(size_t)i], prompts + offs[(size_t)g], prompt_lens[g], n_batch, n_predict,
                                                           out_ids + (size_t)g * n_predict, nullptr);
                if (got < 0) { status[(size_t)i] = got; errs[(size_t)i] = biogpt_hip_last_error(); return; }
                if (out_counts) out_counts[g] = got;
            }
        });
    }
    for (auto &t : pool) t.join();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < n; i++)
        if (status[(size_t)i] != 0) BG_FAIL(status[(size_t)i], "replica %d (device %d): %s", i, r->devices[(size_t)i], errs[(size_t)i].c_str());
    return n_prompts;
}

}  // extern "C"
