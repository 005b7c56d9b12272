#!/usr/bin/env python3
"""bench.py -- decode tokens/sec of the MI355X-native BioGPT engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W                 (N > 1 without a launcher: re-executes itself under the next line)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W            (N > 1: one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic input: ONE greedy 200-token
continuation of a 4-token prompt per rank (examples/main/main.cpp:91-151 with --top_k 1, -n 200, -b 8:
one N=4 prompt eval, then 199 single-token evals; 200 ids sampled), on the synthetic seeded
BioGPT-base model quantized to Q4_0 (BASELINE.json configs[1]; n_ctx = n_positions = 1024).
Weights, KV cache, logits, arg-max and the token feedback all stay in HBM inside the timed region.
Ranks are independent replicas (weak scaling): rank 0 loads the file and the packed weight arena is
broadcast once over RCCL before the timed region; there is no collective on the data path.

Prints ONE JSON line on rank 0 (see the driver contract); extra keys: roofline, cpu_baseline,
token_roofline, api_loop, long_context.
"""
import argparse
import json
import os
import sys
import time

# multi-process RCCL on this host driver needs dmabuf IPC (see biogpt.cpp_amd/replicas.py); read by the HSA runtime at the first HIP call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy rate
SEED = 0x42494F47       # "BIOG" 


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def ensure_model(pkg, workdir, ftype_name, n_layer):
    """Synthetic seeded BioGPT-base file in the reference's format, quantized by the build's own quantizer."""
    os.makedirs(workdir, exist_ok=True)
    tag = "L%d" % n_layer
    f32 = os.path.join(workdir, "synthetic-%s-f32.bin" % tag)
    out = os.path.join(workdir, "synthetic-%s-%s.bin" % (tag, ftype_name))
    if not os.path.exists(out):
        t0 = time.time()
        if not os.path.exists(f32):
            pkg.write_synthetic(f32 + ".tmp", seed=SEED, n_layer=n_layer)
            os.replace(f32 + ".tmp", f32)
        if ftype_name == "f32":
            return f32
        if ftype_name == "f16":
            pkg.write_synthetic(out + ".tmp", seed=SEED, n_layer=n_layer, ftype=1)
            os.replace(out + ".tmp", out)
            return out
        pkg.quantize_file(f32, out + ".tmp", ftype_name)
        os.replace(out + ".tmp", out)
        log("bench: wrote %s in %.1f s" % (out, time.time() - t0))
    return out


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch_command(n_gpus, argv, port=None):
    """The command `python bench.py --gpus N` turns itself into when no launcher started it (WORLD_SIZE unset): one rank per GPU of this node
    under torch.distributed.run, rendezvous on 127.0.0.1 (the container's hostname may not resolve).  Replaces the ONE biogpt_model_load of
    main.cpp:38 by N replicas.  Returned as a list so that tests can look at it without executing it."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port if port is not None else free_port()), os.path.abspath(__file__)] + list(argv)


def make_prompt(n_vocab, unit):
    import numpy as np
    rng = np.random.default_rng(1000 + unit)
    return [2] + [int(v) for v in rng.integers(4, n_vocab, 3)]   # 4 ids, first is </s> = 2 (biogpt.cpp:859)


def pmc_traffic(model_path):
    """FETCH_SIZE and WRITE_SIZE of dec_xpipe_kernel, one rocprofv3 --pmc pass each (MI355X_MICROARCH.md, HBM: the two do not fit one pass; kernel-trace only beside
    them).  Units: the counters are KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read (the guide's correction: x 2); WRITE_SIZE is uncalibrated and
    reported as it is.  Returns (bytes per launch = 2 x FETCH + WRITE, detail)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    if os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("ROCPROFILER_LIBRARY_CTOR"):
        raise RuntimeError("this run is itself under a profiler")
    here = os.path.dirname(os.path.abspath(__file__))
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="biogpt_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(here, "tools", "pmc_target.py"), model_path, "xpipe"]
            subprocess.run(cmd, timeout=150, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, stdin=subprocess.DEVNULL, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp"), check=False)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if not dbs:
                raise RuntimeError("no rocprofv3 database for " + counter)
            con = sqlite3.connect(dbs[0])
            vals = [r[0] for r in con.execute("select value from counters_collection where counter_name = ? and kernel_name like '%dec_xpipe_kernel%'", (counter,))]
            con.close()
            if not vals:
                raise RuntimeError("no dec_xpipe_kernel dispatch in the " + counter + " pass")
            got[counter] = (sum(vals) / len(vals), len(vals))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch_b = got["FETCH_SIZE"][0] * 1024.0 * 2.0
    write_b = got["WRITE_SIZE"][0] * 1024.0
    detail = {"FETCH_SIZE_KiB": round(got["FETCH_SIZE"][0], 1), "WRITE_SIZE_KiB": round(got["WRITE_SIZE"][0], 1), "dispatches": got["FETCH_SIZE"][1],
              "how": "two rocprofv3 --pmc passes (one counter each, --kernel-trace only) over tools/pmc_target.py <model> xpipe: single-token launches of dec_xpipe_kernel at 104 keys; "
                     "bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 counts a wide coalesced read at half, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB x 1024 (uncalibrated)"}
    return round(fetch_b + write_b, 0), detail


def pmc_chunk_traffic(model_path):
    """FETCH_SIZE of the column-per-XCD chunk launch (dec_xcols_kernel), one rocprofv3 --pmc pass over tools/pmc_target.py <model> chunk (8-token evals at 0 .. 64 keys and at
    296 .. 360 keys): bytes fetched per launch (KiB x 1024 x 2, the gfx950 correction of pmc_traffic), per context variant."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    if os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("ROCPROFILER_LIBRARY_CTOR"):
        raise RuntimeError("this run is itself under a profiler")
    here = os.path.dirname(os.path.abspath(__file__))
    d = tempfile.mkdtemp(prefix="biogpt_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", "FETCH_SIZE", "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(here, "tools", "pmc_target.py"), model_path, "chunk"]
        subprocess.run(cmd, timeout=200, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, stdin=subprocess.DEVNULL, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=False)
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if not dbs:
            raise RuntimeError("no rocprofv3 database")
        con = sqlite3.connect(dbs[0])
        rows = list(con.execute("select kernel_name, value from counters_collection where counter_name = 'FETCH_SIZE' and kernel_name like '%dec_xcols_kernel%'"))
        con.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)
    per = {}
    for name, v in rows:
        k = name.split("(")[0].split("dec_xcols_kernel")[-1]
        per.setdefault(k, []).append(v)
    if not per:
        raise RuntimeError("no dec_xcols_kernel dispatch in the pass")
    return {k: {"fetched_bytes_per_launch": round(sum(v) / len(v) * 1024.0 * 2.0, 0), "dispatches": len(v)} for k, v in per.items()}


def pmc_mfma_busy(model_path, pass_seconds):
    """SQ_VALU_MFMA_BUSY_CYCLES of one 512-column prompt pass (its own rocprofv3 --pmc pass over tools/pmc_target.py <model> prefill, --kernel-trace only beside it), as a
    fraction of the SIMD cycles of the pass: busy / (pass time x 2.4 GHz x 1024 SIMDs).  The counter sums over the SIMDs; one v_mfma_i32_16x16x32_i8 is 16 busy cycles."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    if os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("ROCPROFILER_LIBRARY_CTOR"):
        raise RuntimeError("this run is itself under a profiler")
    here = os.path.dirname(os.path.abspath(__file__))
    d = tempfile.mkdtemp(prefix="biogpt_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(here, "tools", "pmc_target.py"), model_path, "prefill"]
        subprocess.run(cmd, timeout=200, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, stdin=subprocess.DEVNULL, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=False)
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if not dbs:
            raise RuntimeError("no rocprofv3 database")
        con = sqlite3.connect(dbs[0])
        rows = list(con.execute("select kernel_name, value from counters_collection where counter_name = 'SQ_VALU_MFMA_BUSY_CYCLES'"))
        con.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)
    per = {}
    for name, v in rows:
        if v > 0:
            k = name.split("(")[0][-60:]
            per[k] = per.get(k, 0.0) + v
    busy = sum(per.values()) / 2.0      # tools/pmc_target.py runs the pass twice
    if busy <= 0:
        raise RuntimeError("no MFMA dispatch in the pass")
    return {"busy_cycles_per_pass": busy, "frac_of_simd_cycles": round(busy / (pass_seconds * 2.4e9 * 1024), 4),
            "how": "one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass (kernel-trace only beside it) over tools/pmc_target.py <model> prefill; busy cycles of every MFMA kernel of one pass / "
                   "(pass time x 2.4 GHz x 1024 SIMDs); attention runs on the VALU (exact double sums, DESIGN 4.4) and counts as zero"}


def bench_capi(args):
    """`--replicas capi`: the single-process form of SURVEY 8(e) -- biogpt_hip_replicas_load (weights read once, one ncclBroadcast over a communicator from
    ncclCommInitAll) and biogpt_hip_replicas_generate_greedy (a host thread and a stream per device, prompt g on device g mod N).  Same workload, same JSON line
    (value = all devices' tokens / the wall time of the whole call, max over devices by construction)."""
    import numpy as np
    import _pkg
    pkg = _pkg.load()
    pkg.lib()
    n = args.gpus
    path = ensure_model(pkg, args.workdir, args.ftype, args.n_layer)
    t0 = time.time()
    reps = pkg.Replicas(path, list(range(n)))
    t_load = time.time() - t0
    hp = pkg.HParams(**pkg.BIOGPT_BASE)
    n_predict = min(args.n_predict, hp.n_positions - 4)
    def step(k):
        prompts = [make_prompt(hp.n_vocab, k * n + g) for g in range(n)]
        ids, secs = reps.generate_greedy(prompts, n_predict, n_batch=8)
        return ids, secs
    for w in range(args.warmup):
        step(1000 + w)
    t0 = time.perf_counter()
    tokens = 0
    for k in range(args.steps):
        ids, _ = step(args.warmup + k)
        tokens += sum(len(v) for v in ids)
    elapsed = time.perf_counter() - t0
    out = {
        "metric": "decode tokens/sec BioGPT %s n_ctx=%d" % (args.ftype.upper(), hp.n_positions), "value": round(tokens / elapsed, 2), "unit": "tokens/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int8" if args.ftype.startswith("q") else args.ftype, "data": "synthetic",
        "config": {"workload": "greedy %d-token continuation of a 4-token prompt per GPU (main.cpp loop, -b 8, --top_k 1), BioGPT-base %s, F32 KV cache, n_ctx=%d; "
                               "1 step = 1 continuation per device" % (n_predict, args.ftype.upper(), hp.n_positions),
                   "n_layer": args.n_layer, "n_predict": n_predict, "n_prompt": 4, "parallelism": "replicas x%d, ONE process (biogpt_hip_replicas_*)" % n,
                   "weights": "synthetic N(0,0.02^2), seed 0x42494F47, written + quantized by the build's own tools"},
        "load_s": round(t_load, 2),
        "replicas": {"form": "one process, ncclCommInitAll, a host thread and a stream per device (csrc/replicas.cpp)", "ranks_seen": int(reps.count),
                     "broadcast_ms": round(reps.broadcast_seconds * 1e3, 3), "arena_bytes": int(pkg.arena_bytes_for(hp_for(pkg, args)))},
    }
    reps.close()
    try:        # RCCL writes its version banner to the C library's stdout buffer: out with it BEFORE the JSON line, which must be the last line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out))
    sys.stdout.flush()


def hp_for(pkg, args):
    hp = pkg.HParams(**pkg.BIOGPT_BASE)
    hp.n_layer = args.n_layer
    hp.ftype = pkg.FTYPES[args.ftype]
    return hp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ftype", default="q4_0", choices=["f32", "f16", "q4_0", "q4_1", "q5_0", "q5_1", "q8_0"])
    ap.add_argument("--n-predict", type=int, default=200)
    ap.add_argument("--n-layer", type=int, default=24, help="24 = BioGPT-base (anything else is NOT the headline config)")
    ap.add_argument("--workdir", default=os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench"))
    ap.add_argument("--workload", default="decode", choices=["decode", "prefill"],
                    help="decode = headline (configs[1]); prefill = configs[2]: 512-token prompt in chunks of n_batch=8")
    ap.add_argument("--n-prompt", type=int, default=512)
    ap.add_argument("--replicas", default="ranks", choices=["ranks", "capi"],
                    help="N > 1: ranks = one process per GPU under torch.distributed (RCCL), the driver's form; capi = ONE process driving biogpt_hip_replicas_* "
                         "(ncclCommInitAll, a host thread and a stream per device) -- the other form of SURVEY 8(e), for comparison on the same lease")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes behind roofline.traffic")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the bounded cpu_baseline sample")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher (no WORLD_SIZE in the environment): become the launcher.  BIOGPT_BENCH_SELF_LAUNCH=1 takes
    # this branch for N = 1 too (tests on the one-GPU box); =print shows the command and stops.
    if args.replicas == "capi":
        return bench_capi(args)
    forced = os.environ.get("BIOGPT_BENCH_SELF_LAUNCH", "")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or forced in ("1", "print")):
        cmd = self_launch_command(args.gpus, sys.argv[1:])
        if forced == "print":
            print(json.dumps({"self_launch": cmd}))
            return
        log("bench: no launcher in the environment -- starting %d rank(s): %s" % (args.gpus, " ".join(cmd)))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), BIOGPT_BENCH_LAUNCHED="self")
        env.pop("BIOGPT_BENCH_SELF_LAUNCH", None)
        sys.stdout.flush()
        os.execvpe(cmd[0], cmd, env)     # the ranks inherit this stdout: rank 0's JSON line stays the last line on it

    import numpy as np
    import torch
    import _pkg
    pkg = _pkg.load()
    pkg.lib()  # fails loudly if the HIP extension has not been built
    from biogpt_cpp_amd import replicas

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: the launcher's --nproc-per-node must equal --gpus" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("BIOGPT_BENCH_FORCE_DIST") == "1"   # exercise the RCCL path on one GPU (tests)
    if world > 1 or force_dist or os.environ.get("BIOGPT_BENCH_LAUNCHED") == "self":
        dist = replicas.init_process_group("nccl")
        # "did RCCL see N ranks": every rank says which device it bound and what the communicator reports
        log("bench: rank %d / %d bound cuda:%d (%s), %d device(s) visible, backend %s, communicator size %d" % (
            dist.get_rank(), world, torch.cuda.current_device(), torch.cuda.get_device_name(torch.cuda.current_device()),
            torch.cuda.device_count(), dist.get_backend(), dist.get_world_size()))
        force_dist = True

    # ---- model: rank 0 loads the file; the packed arena is broadcast (RCCL over xGMI) ----------------
    t_load0 = time.time()
    bcast_s, bcast_bytes = None, None
    if world == 1 and not force_dist:
        path = ensure_model(pkg, args.workdir, args.ftype, args.n_layer)
        model = pkg.BiogptModel.load(path, device=local_rank)
        arena_t = None
    else:
        hp_list = None
        if rank == 0:
            path = ensure_model(pkg, args.workdir, args.ftype, args.n_layer)
            hp0 = pkg.HParams(**pkg.BIOGPT_BASE)
            hp0.n_layer = args.n_layer
            hp0.ftype = pkg.FTYPES[args.ftype]
            hp_list = [hp0.n_vocab, hp0.n_layer, hp0.n_head, hp0.n_positions, hp0.d_ff, hp0.d_model, hp0.ftype, hp0.n_merges]
        hp_list = replicas.broadcast_hparams(hp_list)
        hp = pkg.HParams(*hp_list)
        nbytes = pkg.arena_bytes_for(hp)
        arena_t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        if rank == 0:
            model = pkg.BiogptModel.load(path, device=local_rank, arena=arena_t.data_ptr(), arena_bytes=nbytes)
        torch.cuda.synchronize()
        tb0 = time.time()
        replicas.broadcast_arena(arena_t, src=0)
        torch.cuda.synchronize()
        tb = time.time() - tb0
        if rank != 0:
            model = pkg.BiogptModel.attach(hp, local_rank, arena_t.data_ptr(), nbytes)
        if rank == 0:
            log("bench: broadcast %.1f MiB arena in %.1f ms" % (nbytes / 2 ** 20, tb * 1e3))
        bcast_s, bcast_bytes = tb, nbytes
    hp = model.hparams
    t_load = time.time() - t_load0

    n_predict = min(args.n_predict, hp.n_positions - 4)
    prefill = args.workload == "prefill"
    n_prompt = min(args.n_prompt, hp.n_positions)

    def run_step(unit):
        if prefill:
            # configs[2]: one 512-token prompt fed as consecutive evals of n_batch = 8 tokens (main.cpp:129-137),
            # no intra-chunk mask (F1); logits stay on the device
            rng = np.random.default_rng(7000 + unit)
            toks = [2] + [int(v) for v in rng.integers(4, hp.n_vocab, n_prompt - 1)]
            if os.environ.get("BIOGPT_BENCH_CHUNK_CALLS"):      # one library call per reference chunk (the biogpt_eval loop)
                for n_past in range(0, n_prompt, 8):
                    model.eval_device(toks[n_past:n_past + 8], n_past)
            else:                                                # same result, several chunks per pass (biogpt_hip_eval_prompt)
                model.eval_prompt(toks, 0, 8, want_logits=False)
            model.synchronize()
            return None, 0.0
        ids, secs = model.generate_greedy(make_prompt(hp.n_vocab, unit), n_predict, n_batch=8)
        return ids, secs

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        run_step(rank * 1000 + w)
    barrier()
    t0 = time.perf_counter()
    last_ids = None
    for k in range(args.steps):
        last_ids, _ = run_step((args.warmup + k) * world + rank)
    torch.cuda.synchronize()
    model.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    report = None
    if dist is not None:
        report = replicas.replica_report(elapsed, args.steps * (n_prompt if prefill else n_predict), torch.cuda.get_device_name(torch.cuda.current_device()), bcast_s, bcast_bytes)
        elapsed = replicas.max_over_ranks(elapsed)

    total_tokens = world * args.steps * (n_prompt if prefill else n_predict)
    value = total_tokens / elapsed
    dump = os.environ.get("BIOGPT_BENCH_DUMP_IDS")   # tests: the last timed continuation of this rank, for the oracle to check
    if dump and not prefill and last_ids is not None:
        unit = (args.warmup + args.steps - 1) * world + rank
        with open("%s.rank%d" % (dump, rank), "w") as f:
            json.dump({"model": path if rank == 0 else None, "prompt": make_prompt(hp.n_vocab, unit), "ids": [int(v) for v in last_ids]}, f)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    out = {
        "metric": ("prefill tokens/sec BioGPT %s n_batch=8 prompt=%d" % (args.ftype.upper(), n_prompt)) if prefill else
                  ("decode tokens/sec BioGPT %s n_ctx=%d" % (args.ftype.upper(), hp.n_positions)),
        "value": round(value, 2),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # the timed steps' own launches: tokens of each multi-token pipelined launch of a continuation (the first token comes out of the prompt's eval) and the step's time per
        # token; the kernel trace of exactly these launches: profiles/rocprofv3_kernel_stats_r6_headline.csv (dominant-kernel time x launches <= ms_per_step from the evidence)
        **({"launch_tokens": model.generate_launches(), "us_per_token_in_launch": round(elapsed / args.steps / max(1, n_predict) * 1e6, 2)} if not prefill else {}),
        "dtype": "int8" if args.ftype.startswith("q") else args.ftype,  # W4/5/8 x A8 integer block dots, f32 scale/accumulate
        "data": "synthetic",
        "config": {
            "workload": "greedy %d-token continuation of a 4-token prompt per GPU (main.cpp loop, -b 8, --top_k 1), "
                        "BioGPT-base %s, F32 KV cache, n_ctx=%d; 1 step = 1 continuation per rank" % (n_predict, args.ftype.upper(), hp.n_positions),
            "n_layer": hp.n_layer, "d_model": hp.d_model, "d_ff": hp.d_ff, "n_vocab": hp.n_vocab,
            "n_predict": n_predict, "n_prompt": 4, "parallelism": "replicas x%d (weights RCCL-broadcast once)" % world,
            "weights": "synthetic N(0,0.02^2), seed 0x42494F47, written + quantized by the build's own tools",
        },
        "load_s": round(t_load, 2),
    }
    if report is not None:
        out["replicas"] = dict(report, form="one process per GPU (torch.distributed, backend nccl = RCCL), one broadcast of the packed weight arena from rank 0")

    if prefill:
        chunk_calls = bool(os.environ.get("BIOGPT_BENCH_CHUNK_CALLS"))
        cols = 8 if chunk_calls else max(8, int(os.environ.get("BIOGPT_HIP_PROMPT_COLS", "512")) // 8 * 8)
        out["config"]["workload"] = ("%d-token prompt = %d reference evals of n_batch=8 tokens (no mask inside an eval, F1), BioGPT-base %s; "
                                     "%s; 1 step = 1 prompt per rank; attention path: %s" % (
                                         n_prompt, (n_prompt + 7) // 8, args.ftype.upper(),
                                         "one library call per eval" if chunk_calls else
                                         "biogpt_hip_eval_prompt: %d columns (%d evals) per pass, each column limited to its own eval's keys -- same logits and KV rows" % (cols, cols // 8),
                                         "register-tiled VALU kernel, products rounded to f32, double accumulation (bit-parity path)"))
        # weights are streamed once per pass: algorithmic bytes per pass = W + KV read of its columns' contexts
        passes = (n_prompt + cols - 1) // cols
        b = sum(pkg.decode_bytes_per_token(hp, min(n_prompt, cols * (k + 1))) for k in range(passes))
        t_prompt = elapsed / args.steps
        out["token_roofline"] = {"bytes_per_prompt": int(b), "passes": passes, "GBps": round(b / t_prompt / 1e9, 1), "frac_of_peak": round(b / t_prompt / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- roofline of the dominant kernel + whole-token figure (N = 1 only) ---------------------------
    if world == 1 and not prefill:
        try:
            reps = 24 * 20
            quant = args.ftype.startswith("q")
            if quant:   # the five launches of a decode layer (csrc/kernels_decode.hip.h), each timed on its own: HIP events around
                        # `reps` back-to-back launches cycling through the 24 layers' REAL arena weights (fc1: 24 x 4096 rows = 57 MB per sweep)
                per = {}
                for name, which in (("qkv", 6), ("attention@104keys", 7), ("out_proj", 8), ("fc1", 9), ("fc2", 10)):
                    sk, bk = model.bench_matvec(which, layer=0, reps=reps)
                    per[name] = {"GBps": round(bk / sk / 1e9, 1), "us": round(sk * 1e6, 3), "bytes": bk, "frac": round(bk / sk / 1e9 / HBM_PEAK_GBS, 4)}
                secs, nbytes = per["fc1"]["us"] * 1e-6, per["fc1"]["bytes"]
                kname = "dec_fc1_kernel<%s> (fc1 %dx%d: LayerNorm + W*A8 mat-vec + GELU + Q8 output, 24 launches/token)" % (args.ftype.upper(), hp.d_ff, hp.d_model)
                method = None
                if model.xpipe_state() == 1:
                    # the decode step of this run is ONE persistent launch for all layers (csrc/kernels_xpipe.hip.h) + the lm_head launch:
                    # that launch is the dominant kernel; the five-launch layer above stays as the fallback's figures
                    sx, bx = model.bench_matvec(11, layer=0, reps=40)
                    per["five_launch_layer_fc1"] = per.pop("fc1")
                    per = {("five_launch_layer_" + k if not k.startswith("five_") else k): v for k, v in per.items()}
                    secs, nbytes = sx, bx
                    kname = ("dec_xpipe_kernel<%s> (all %d layers + final LayerNorm + lm_head of one token at 104 keys in ONE persistent launch, layer l on the 32 compute units "
                             "of XCD l %% 8, weights stationary in registers, hand-offs through the XCD's L2; the headline's 200-token continuation runs the same kernel with "
                             "60-72 tokens per launch)" % (args.ftype.upper(), hp.n_layer))
                    method = ("HIP events on the engine stream around 40 back-to-back replays of this ONE kernel as a single-token launch (hipGraph, as biogpt_eval's step is replayed), real arena weights; "
                              "algorithmic bytes = the four matrices of every layer and the output projection at file density + K / V rows of 104 keys + the new K / V rows + x in / out + the logits row; "
                              "rocprofv3 agreement: profiles/rocprofv3_kernel_stats_r5.csv is taken with BIOGPT_HIP_XPIPE_MULTI=0 BIOGPT_HIP_RESIDENT=0 (every call one token); "
                              "the launch is LATENCY-bound by design: a layer is %.1f us of dependent stages (profiles/xpipe_timeline_r4.txt) on 1/8 of the chip while the "
                              "other XCDs prefetch -- 8 TB/s would move a layer's 7.1 MB in 0.9 us" % (sx * 1e6 / hp.n_layer))
            else:
                secs, nbytes = model.bench_matvec(0, layer=0, reps=reps)        # fc1: LN + mat-vec + GELU (generic kernels)
                secs2, nbytes2 = model.bench_matvec(1, layer=0, reps=reps)
                per = {"fc2": {"GBps": round(nbytes2 / secs2 / 1e9, 1), "us": round(secs2 * 1e6, 3), "bytes": nbytes2, "frac": round(nbytes2 / secs2 / 1e9 / HBM_PEAK_GBS, 4)}}
                kname = "matvec_kernel<%s,LN,GELU> (fc1 %dx%d, 24 launches/token)" % (args.ftype.upper(), hp.d_ff, hp.d_model)
                method = None
                if model.fpipe_launches() >= 0:
                    # float files: a single-token step up to 224 keys is ONE persistent launch for all layers (csrc/kernels_fpipe.hip.h) + the lm_head launch: that launch is
                    # the dominant kernel; the five-launch layer's fc1 / fc2 above stay as the figures of contexts beyond 224 keys
                    sx, bx = model.bench_matvec(13, layer=0, reps=40)
                    per["five_launch_layer_fc1"] = {"GBps": round(nbytes / secs / 1e9, 1), "us": round(secs * 1e6, 3), "bytes": nbytes, "frac": round(nbytes / secs / 1e9 / HBM_PEAK_GBS, 4)}
                    per["five_launch_layer_fc2"] = per.pop("fc2")
                    secs, nbytes = sx, bx
                    kname = ("fpipe_kernel<%s> (all %d layers of one token at 104 keys in ONE persistent launch: 256 workgroups, each holding its rows of the layer's four matrices in "
                             "registers a layer ahead; stage outputs as tagged granules collected by two polling waves per workgroup)" % (args.ftype.upper(), hp.n_layer))
                    method = ("HIP events on the engine stream around 40 back-to-back replays of this ONE kernel as a single-token launch (hipGraph, as the decode step is replayed), real arena weights; "
                              "algorithmic bytes = the four matrices of every layer at file density + K / V rows of 104 keys + the new K / V rows + x in / out; the launch is bound by its "
                              "dependent chain, not by bytes: %.1f us per layer = five all-to-all hand-overs + row dots on one wave per SIMD + two LayerNorms + the attention stage "
                              "(profiles/fpipe_timeline_r6_*.txt)" % (sx * 1e6 / hp.n_layer))
            secs_lm, nbytes_lm = model.bench_matvec(4, layer=0, reps=50)    # lm_head
            per["lm_head"] = {"GBps": round(nbytes_lm / secs_lm / 1e9, 1), "us": round(secs_lm * 1e6, 3), "bytes": nbytes_lm, "frac": round(nbytes_lm / secs_lm / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "HIP events around 50 back-to-back launches on the ONE copy of the matrix the model owns: between launches its 24.6 MB stay in the 256 MB Infinity Cache, so "
                                      "this is a fabric-side rate, not an HBM rate (a pure read of the same bytes takes 2.4 us); rocprofv3's average over the calls of a whole bench run is higher "
                                      "(profiles/rocprofv3_kernel_stats_r5.csv: the first calls after other work are cold)"}
            try:        # the same kernel with the weights NOT cache-resident: a different one of 14 copies of the matrix per launch (344 MB) -- the figure an "HBM roofline" means
                secs_lc, _ = model.bench_matvec(12, layer=0, reps=56)
                per["lm_head_cold"] = {"GBps": round(nbytes_lm / secs_lc / 1e9, 1), "us": round(secs_lc * 1e6, 3), "bytes": nbytes_lm, "frac": round(nbytes_lm / secs_lc / 1e9 / HBM_PEAK_GBS, 4),
                                       "note": "weights cycled through 14 device copies (344 MB > Infinity Cache + L2s): every launch streams its 24.6 MB from HBM"}
            except Exception as e:
                per["lm_head_cold"] = {"error": str(e)[:200]}
            ach = nbytes / secs / 1e9
            out["roofline"] = {
                "bound": "hbm", "kernel": kname,
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                # HBM bytes from PMC counters need their own rocprofv3 --pmc passes: two child processes after the timed work (pmc_traffic below; null when
                # rocprofv3 is not usable here); summaries of the same passes are committed under profiles/ (pmc_*_r4.txt, pmc_attn_tile_r5.txt)
                "traffic": None,
                "bytes_per_launch": nbytes, "us_per_launch": round(secs * 1e6, 3),
                "method": (method if method else
                           "HIP events on the engine stream around %d back-to-back launches of this ONE kernel (hipGraph replays of one sweep over the layers, as the decode step is replayed) cycling through the 24 layers' own "
                           "arena weights (no L2 reuse between launches; the 211 MB arena fits the 256 MB Infinity Cache); model shapes only -- "
                           "a launch moves %.1f MB, which 8 TB/s would move in %.2f us, against a measured ~1.3-1.6 us launch boundary" % (reps, nbytes / 1e6, nbytes / 8e6)),
                "other_kernels": per,
            }
            if quant:
                # north_star's threshold (">= 70 % of the HBM roofline on the Q4_0 single-token decode mat-vec at d_model = 1024") on the MODEL'S OWN bytes: every block-
                # quantized matrix of the arena (24 x {q/k/v, out_proj, fc1, fc2} + lm_head) x a resident Q8 activation of its shape in ONE launch, rows spread over the
                # chip, two copies of the weights in turn so that the Infinity Cache cannot serve them (csrc/kernels_sweep.hip.h; SURVEY 7 "hard parts")
                try:
                    sw_s, sw_b, sw_chk = model.bench_sweep(reps=120)
                    out["matvec_sweep"] = {"what": "matvec_sweep_kernel<%s>: all %d block-quantized matrices of the model (the W of SURVEY 8(d)) as independent mat-vecs in one launch; int8 block "
                                                   "dots, block terms added in block order (the reference's arithmetic); launches alternate between two copies of the weights (2 x %.0f MB > the "
                                                   "256 MB Infinity Cache)" % (args.ftype.upper(), 4 * hp.n_layer + 1, sw_b / 1e6),
                                           "bytes": sw_b, "us": round(sw_s * 1e6, 2), "GBps": round(sw_b / sw_s / 1e9, 1), "frac": round(sw_b / sw_s / 1e9 / HBM_PEAK_GBS, 4),
                                           "max_abs_diff_vs_host_recompute": sw_chk, "method": "HIP events around 120 back-to-back launches (launch gaps included)"}
                    # the same kernel on each shape alone, as many copies of its weights in turn as exceed the Infinity Cache (SURVEY 8(d) words the bar per shape):
                    # "k1024_dxd" = q/k/v + out_proj of every layer = the Q4_0 mat-vec at d_model = 1024 to the letter (biogpt.cpp:705-716,767)
                    shapes = {}
                    for key, which in (("k1024_dxd", 2), ("k4096_fc2", 3), ("k1024_fc1", 4), ("lm_head", 1)):
                        s_s, s_b, s_chk = model.bench_sweep(reps=120, which=which)
                        shapes[key] = {"bytes": s_b, "us": round(s_s * 1e6, 2), "GBps": round(s_b / s_s / 1e9, 1), "frac": round(s_b / s_s / 1e9 / HBM_PEAK_GBS, 4), "max_abs_diff_vs_host_recompute": s_chk}
                    out["matvec_sweep"]["by_shape"] = shapes
                except Exception as e:
                    out["matvec_sweep"] = {"error": str(e)[:200]}
            if args.ftype == "q4_0":
                # NOT a roofline answer for this model (no BASELINE config has such a matrix): the mat-vec kernel body on a
                # 524288 x 1024 synthetic Q4_0 stream (302 MB > L2 + Infinity Cache), to show what the loop sustains per byte
                ss, sb = model.bench_stream(1 << 19, 20, 8)
                out["stream_probe"] = {"what": "matvec_fast_kernel<Q4_0,LN,LOGITS,1024> on 524288 x 1024 synthetic Q4_0 rows; outside every BASELINE config",
                                       "bytes_per_launch": sb, "us_per_launch": round(ss * 1e6, 2), "GBps": round(sb / ss / 1e9, 1)}
            # whole-token: graph replay at fixed context, HIP-event timed
            tok = {}
            for T in (104, 512, 1024):
                s = model.bench_decode(T - 1, reps=30)
                b = pkg.decode_bytes_per_token(hp, T)
                tok["T=%d" % T] = {"us_per_token": round(s * 1e6, 2), "tokens_per_s": round(1.0 / s, 1),
                                   "GBps": round(b / s / 1e9, 1), "frac_of_peak": round(b / s / 1e9 / HBM_PEAK_GBS, 4),
                                   "bytes_per_token": int(b)}
            out["token_roofline"] = tok
            out["xpipe_state"] = model.xpipe_state()   # 1: the pipelined launches are in use (a hand-off time-out or a disturbed launch would have left -1 and the five-launch layer)
            multi = os.environ.get("BIOGPT_HIP_XPIPE_MULTI", "1") != "0"
            out["decode_path"] = (("xcd-pipeline: layers + lm_head + greedy sampler in ONE persistent launch per context bucket (64 / 128 / 192 / 256 keys: a head's K / V rows in its workgroup's "
                                   "registers; 257 .. 512 keys: two workgroups per head; 513 .. 1024 keys: every head's keys spread over 16 helper workgroups of XCD head / 2, csrc/kernels_xlong.hip.h)" if multi else
                                   "xcd-pipeline: layers + lm_head in ONE persistent launch per token") if model.xpipe_state() == 1 else "five launches per layer + lm_head")
            # batched multi-sequence decode on this one GPU (biogpt_hip_generate_greedy_batch): S independent
            # 200-token continuations decoded together, weights read once per step for all S -- NOT the headline
            # (configs[1] is single-stream), reported because it is what the HBM-bound regime of this chip looks like
            if args.ftype.startswith("q"):
                ms = {}
                for S in (8, 32, 64, 256):  # from 48 sequences the chain runs on the int8 matrix cores (kernels_mfma.hip.h)
                    prompts = [make_prompt(hp.n_vocab, 9000 + i) for i in range(S)]
                    model.generate_greedy_batch(prompts, 8, n_batch=8)            # warm-up: allocations + graph capture
                    ids_b, secs_b = model.generate_greedy_batch(prompts, n_predict, n_batch=8)
                    # algorithmic HBM bytes of the run: weights once per step + every sequence's KV read/write and logits row
                    w_bytes = pkg.decode_bytes_per_token(hp, 0) - (2 * hp.n_layer * hp.d_model * 4 + hp.n_vocab * 4)   # weights only
                    run_bytes = sum(w_bytes + S * (pkg.decode_bytes_per_token(hp, 4 + k) - w_bytes) for k in range(1, n_predict + 1))
                    ms["S=%d" % S] = {"tokens_per_s": round(S * n_predict / secs_b, 1), "ms_per_step_all_seqs": round(secs_b / n_predict * 1e3, 4),
                                      "GBps": round(run_bytes / secs_b / 1e9, 1), "frac_of_peak": round(run_bytes / secs_b / 1e9 / HBM_PEAK_GBS, 4)}
                ms["note"] = ("S = 8: every decode step is ONE launch with one sequence per XCD (csrc/kernels_xcols.hip.h, streams mode; each XCD streams all weights: the algorithmic bytes "
                              "of GBps count them once); S >= 32: the launch chain, from 48 sequences on the int8 matrix cores")
                out["multi_stream"] = ms
            # prompt ingestion (configs[2] in short): a 512-token prompt with -b 8 semantics through biogpt_hip_eval_prompt
            rngp = np.random.default_rng(7000)
            ptoks = [2] + [int(v) for v in rngp.integers(4, hp.n_vocab, 511)]
            model.eval_prompt(ptoks, 0, 8, want_logits=False); model.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                model.eval_prompt(ptoks, 0, 8, want_logits=False)
            model.synchronize()
            t_pass = (time.perf_counter() - t1) / 3
            # configs[2] in the driver's line: bytes (weights once + the K / V rows each column reads and writes + activations) and int8 operations of the pass against the two peaks
            pb = pkg.decode_bytes_per_token(hp, 512)
            mat = hp.n_layer * (3 * hp.d_model * hp.d_model + hp.d_model * hp.d_model + 2 * hp.d_ff * hp.d_model)
            iops = 2.0 * 512 * mat + 2.0 * hp.n_vocab * hp.d_model
            out["prompt_pass"] = {"tokens_per_s": round(512 / t_pass, 1), "ms_per_pass": round(t_pass * 1e3, 3),
                                  "roofline": {"bytes_per_pass": int(pb), "GBps": round(pb / t_pass / 1e9, 1), "frac_of_hbm_peak": round(pb / t_pass / 1e9 / HBM_PEAK_GBS, 4),
                                               "int8_ops_per_pass": iops, "TOPS": round(iops / t_pass / 1e12, 1), "int8_peak_TOPS": 3944.0, "frac_of_int8_peak": round(iops / t_pass / 1e12 / 3944.0, 4),
                                               "mfma_busy": None,
                                               "bound": "neither peak: the chain is VALU-bound -- every v_mfma_i32_16x16x32_i8 (16 cycles) is followed by the exact per-block f32 scaling of its 4 outputs per lane "
                                                        "(the reference's arithmetic), and the attention (exact double sums) runs on the VALU"},
                                  "note": "512-token prompt, 64 reference evals of n_batch=8 in one pass (bench.py --workload prefill is the full bench line)"}
            # the reference's OWN prompt loop (main.cpp:129-137): one biogpt_eval per n_batch = 8 tokens, the row copied out every time.  Up to 512 keys such an eval is ONE
            # persistent launch with one column per XCD (csrc/kernels_xcols.hip.h), beyond that the launch chain of kernels_fast.hip.h
            try:
                def chunk_loop(n_tok):
                    t0 = time.perf_counter()
                    for at in range(0, n_tok, 8):
                        model.eval(ptoks[at:at + 8], at)
                    return time.perf_counter() - t0
                chunk_loop(256)
                before = model.chunk_launches()
                t256 = min(chunk_loop(256) for _ in range(3))
                per_loop = (model.chunk_launches() - before) // 3
                t512 = min(chunk_loop(512) for _ in range(2))
                out["prompt_chunk_evals"] = {"tokens_per_s_0_256_keys": round(256 / t256, 1), "ms_per_eval_0_256_keys": round(t256 / 32 * 1e3, 3),
                                             "chunk_launches_per_32_evals": int(per_loop),
                                             "tokens_per_s_512_token_prompt": round(512 / t512, 1), "ms_per_eval_257_512_keys": round((t512 - t256) / 32 * 1e3, 3),
                                             "note": "biogpt_hip_eval per 8-token chunk from Python (ctypes), rows copied to the host; 0 .. 256 keys: the column-per-XCD launch "
                                                     "(every XCD streams all weights: 16.7 us per layer against 9.8 us of pure weight stream and ~ 11 us of dependent chain, HISTORY.md round 4; profiles/xcols_timeline_r4.txt), "
                                                     "257 .. 512 keys: the same launch in its 512-key variant (round 6: the second half of a head's old K / V rows requested at the start of the attention stage; the launch chain it replaces: 0.92 ms per eval); beyond 512 keys the launch chain"}
            except Exception as e:
                out["prompt_chunk_evals"] = {"error": str(e)[:300]}
            # the drop-in API loop as a C++ caller runs it (main.cpp:91-151: one eval call per token, sampler on the host; never
            # `value`): the whole logits row over PCIe + host arg-max, and eval + device top-40 (512 bytes over PCIe)
            pr = make_prompt(hp.n_vocab, 7)
            model.bench_api_loop(pr, 8, 0); model.bench_api_loop(pr, 8, 1)          # warm-up: graph capture, staging buffers
            ids0, s0 = model.bench_api_loop(pr, n_predict, 0)
            ids1, s1 = model.bench_api_loop(pr, n_predict, 1)
            dev_ids, _ = model.generate_greedy(pr, n_predict, n_batch=8)
            out["api_loop"] = {"tokens_per_s": round(n_predict / s0, 1), "frac_of_device_loop": round(n_predict / s0 / value, 3),
                               "ids_match_device_loop": bool((np.asarray(ids0) == np.asarray(dev_ids)).all()),
                               "speculation": model.resident_stats(),
                               "note": "biogpt_eval per token: 170 KB logits row to the host (PCIe) + host arg-max (std::max_element), C++ loop; a greedy caller's next "
                                       "position is started by the resident launch from its own arg-max while the host still reads the row, the call then only confirms the "
                                       "token (speculation.hits of the calls; BIOGPT_HIP_SPEC=0: eval_only_tokens_per_s below is that rate)"}
            model.bench_api_loop(pr, 8, 3)
            ids3, s3 = model.bench_api_loop(pr, n_predict, 3)
            _, s4 = model.bench_api_loop(pr, n_predict, 4)
            out["api_loop_inplace"] = {"tokens_per_s": round(n_predict / s3, 1), "frac_of_device_loop": round(n_predict / s3 / value, 3),
                                       "ids_match_device_loop": bool((np.asarray(ids3) == np.asarray(dev_ids)).all()),
                                       "eval_only_tokens_per_s": round(n_predict / s4, 1),
                                       "note": "biogpt_hip_eval_inplace per token (the row read where the launch wrote it in pinned host memory) + 8-lane host arg-max, C++ loop; "
                                               "eval_only: the same calls without the arg-max (token fixed: what a caller that samples gets, the launch waits for every token). One resident pipelined launch per context bucket serves the calls "
                                               "(BIOGPT_HIP_RESIDENT=0: one launch per call)"}
            # the same loop beyond 256 keys (the long-context launch in its resident form): a 300-token prompt, then 200 single-token calls at 301 .. 500 keys
            rngl = np.random.default_rng(11)
            prl = [2] + [int(v) for v in rngl.integers(4, hp.n_vocab, 299)]
            model.generate_greedy(prl, n_predict, n_batch=len(prl))
            idl, sdl = model.generate_greedy(prl, n_predict, n_batch=len(prl))
            model.bench_api_loop(prl, 8, 0)
            ial, sal = model.bench_api_loop(prl, n_predict, 0)
            out["api_loop_long"] = {"contexts": "301 .. %d keys" % (300 + n_predict), "us_per_token": round(sal / n_predict * 1e6, 1),
                                    "device_loop_us_per_token": round(sdl / n_predict * 1e6, 1), "frac_of_device_loop": round(sdl / sal, 3),
                                    "ids_match_device_loop": bool((np.asarray(ial) == np.asarray(idl)).all()),
                                    "note": "biogpt_eval per token + host arg-max as in api_loop, after a 300-token prompt (both figures include their prompt pass; both feed the "
                                            "prompt as ONE 300-token eval -- the device loop with n_batch = 300 -- so the ids are comparable)"}
            out["api_loop_topk"] = {"tokens_per_s": round(n_predict / s1, 1), "frac_of_device_loop": round(n_predict / s1 / value, 3),
                                    "ids_match_device_loop": bool((np.asarray(ids1) == np.asarray(dev_ids)).all()),
                                    "note": "biogpt_eval_sample-style: biogpt_hip_eval_topk(k = 40) per token, C++ loop, next token = the first id: the resident launch + a one-pass top-40 selection over the pinned row on the host"}
        except Exception as e:  # keep the headline line even if a side measurement fails
            out["roofline_error"] = str(e)
        # configs[3]: the same workload on the other block formats (same synthetic weights, quantized by the build's own quantizer), 3
        # continuations each after one warm-up; each model is its own context on this device -- the pipeline slot is handed over
        # between contexts whenever the holder has synchronised
        if args.ftype == "q4_0" and not os.environ.get("BIOGPT_BENCH_SKIP_TYPES"):
            # ... and the F32 file of configs[0] (README.md:24,45 / main.cpp:160: 4-token prompt, 200 tokens): single-token steps as ONE persistent launch per token (csrc/kernels_fpipe.hip.h)
            for other in ("q5_1", "q8_0", "f32"):
                try:
                    mo = pkg.BiogptModel.load(ensure_model(pkg, args.workdir, other, args.n_layer), device=local_rank)
                    mo.generate_greedy(make_prompt(hp.n_vocab, 1), n_predict, n_batch=8)
                    t1 = time.perf_counter()
                    for k in range(3):
                        mo.generate_greedy(make_prompt(hp.n_vocab, 2 + k), n_predict, n_batch=8)
                    mo.synchronize()
                    dt = (time.perf_counter() - t1) / 3
                    s1k = mo.bench_decode(1023, reps=20)
                    b104, b1024 = pkg.decode_bytes_per_token(mo.hparams, 104), pkg.decode_bytes_per_token(mo.hparams, 1024)
                    out["decode_" + other] = {"tokens_per_s": round(n_predict / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": 3,
                                              "frac_of_peak_T104": round(b104 * n_predict / dt / 1e9 / HBM_PEAK_GBS, 4),
                                              "T=1024": {"us_per_token": round(s1k * 1e6, 2), "tokens_per_s": round(1.0 / s1k, 1),
                                                         "frac_of_peak": round(b1024 / s1k / 1e9 / HBM_PEAK_GBS, 4)},
                                              "xpipe_state": mo.xpipe_state()}
                    if other == "f32":
                        out["decode_f32"]["persistent_launches"] = mo.fpipe_launches()
                    mo.close()
                except Exception as e:
                    out["decode_" + other] = {"error": str(e)}

    # ---- CPU baseline: the oracle (restatement of the reference's ggml CPU path), bounded samples -----
    # Four legs on the GPU box's host cores, each a bounded sample of the SAME workload (whole 200-token continuations of 4-token prompts, eval time only as main.cpp:96-103
    # counts it): the block dots as ggml's AVX2 kernels execute them (intrinsics; bo_opts.assoc = 7) on all cores and at the reference CLI's default -t 4 (biogpt.h:111), the
    # scalar fallbacks (assoc = 0, the association the HIP kernels are bit-identical to) on all cores, and the fp32 file (BASELINE configs[0], README.md:24,45) at -t 4 and all cores.
    if world == 1 and not args.no_cpu_baseline and not prefill:
        try:
            from oracle import oracle as O
            cores = usable_cores()
            simd = 7 if O.have_avx2() else 0
            try:
                cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
            except Exception:
                cpu_model = "unknown"

            def leg(file_path, threads, assoc, budget_s, check_ids=None):
                om = O.OracleModel(file_path, n_threads=threads, assoc=assoc)
                tok, sec, k, first = 0, 0.0, 0, None
                while sec < budget_s and k < 64:
                    ids, secs = om.generate_greedy(make_prompt(hp.n_vocab, 5000 + k), n_predict, n_batch=8)
                    if k == 0:
                        first = list(ids)
                    tok += len(ids); sec += secs; k += 1
                om.close()
                r = {"value": round(tok / sec, 2), "cores": threads, "continuations": k, "eval_s": round(sec, 1)}
                if check_ids is not None:
                    r["ids_match_gpu"] = bool(first == list(check_ids))
                return r

            g_ids, _ = model.generate_greedy(make_prompt(hp.n_vocab, 5000), n_predict, n_batch=8)
            b = args.cpu_seconds
            scalar = leg(path, cores, 0, b / 2, check_ids=g_ids)            # the parity association: its ids ARE the GPU's
            simd_all = leg(path, cores, simd, b / 2) if simd else None
            simd_t4 = leg(path, min(4, cores), simd, b / 4) if simd else leg(path, min(4, cores), 0, b / 4)
            legs = {"scalar": scalar, "simd": simd_all, "threads_4": simd_t4}
            if args.ftype != "f32":
                try:
                    f32_path = ensure_model(pkg, args.workdir, "f32", hp.n_layer)
                    legs["fp32_t4"] = leg(f32_path, min(4, cores), 5 if simd else 0, b / 4)
                    legs["fp32_all"] = leg(f32_path, cores, 5 if simd else 0, b / 4)
                except Exception as e:
                    legs["fp32_error"] = str(e)[:200]
            head = simd_all if simd_all else scalar
            out["cpu_baseline"] = {
                "value": head["value"], "unit": "tokens/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
                "sample": "oracle (C restatement of the reference's ggml CPU path, OpenMP over mat-mul rows), same %s file, greedy %d-token continuations of 4-token prompts; "
                          "value = %s on all cores: %d continuation(s), %.1f s of eval time"
                          % (args.ftype.upper(), n_predict, "block dots / activation conversion with AVX2 + FMA intrinsics in the shape of ggml's AVX2 kernels (assoc 7)" if simd_all else
                             "scalar block dots (no AVX2 on this host)", head["continuations"], head["eval_s"]),
                "ids_match_gpu": scalar.get("ids_match_gpu"),
                **legs,
                "note": "scalar: ggml's scalar fallbacks (the association the HIP kernels reproduce bit for bit); simd / threads_4: the AVX2 shape (its logits differ from the scalar "
                        "mode's by up to ~3e-2 with block-quantized weights, tests/test_oracle_assoc.py); fp32_*: the fp32 file of BASELINE configs[0]; the reference's own published "
                        "figure is 125 tok/s fp16 on an M1 (README.md:56)",
            }
        except Exception as e:
            out["cpu_baseline_error"] = str(e)

    # the PMC passes below are child processes on this device: the pipelined launches belong to ONE process per device (engine_xpipe.inc, xpipe_process_lock), and
    # everything that is timed is done -- this process lets go of its context (and with it of the device's lock file) first
    if world == 1 and not args.no_pmc:
        try:
            model.close()
        except Exception:
            pass
    # ---- roofline.traffic: HBM-side bytes per launch of the dominant kernel from the PMC counters, each counter in its OWN rocprofv3 pass over a small child
    #      process (tools/pmc_target.py: 24 single-token launches of the pipelined kernel at 104 keys), after everything that is timed
    if world == 1 and args.ftype.startswith("q") and not prefill and not args.no_pmc and "roofline" in out and "dec_xpipe" in out["roofline"].get("kernel", ""):
        try:
            t, detail = pmc_traffic(path)
            out["roofline"]["traffic"] = t
            out["roofline"]["traffic_detail"] = detail
        except Exception as e:
            out["roofline"]["traffic_detail"] = {"error": str(e)[:300]}

    if world == 1 and not prefill and not args.no_pmc and isinstance(out.get("prompt_chunk_evals"), dict) and "error" not in out["prompt_chunk_evals"]:
        # fetched / algorithmic bytes of the chunk launch: every XCD streams ALL weights (kernels_xcols.hip.h), so an 8-token eval fetches ~ 8 x the 196 MB a single stream needs
        try:
            tr = pmc_chunk_traffic(path)
            alg = float(pkg.decode_bytes_per_token(hp, 64))
            out["prompt_chunk_evals"]["traffic"] = {k: dict(v, fetched_over_algorithmic=round(v["fetched_bytes_per_launch"] / alg, 2)) for k, v in tr.items()}
            out["prompt_chunk_evals"]["traffic"]["algorithmic_bytes_per_eval"] = alg
            out["prompt_chunk_evals"]["traffic"]["how"] = ("one rocprofv3 --pmc FETCH_SIZE pass (kernel-trace only beside it) over tools/pmc_target.py <model> chunk; KiB x 1024 x 2 (gfx950); "
                                                            "algorithmic = the weights once + K / V rows (SURVEY 8(d), B at 64 keys)")
        except Exception as e:
            out["prompt_chunk_evals"]["traffic"] = {"error": str(e)[:300]}

    if world == 1 and not prefill and not args.no_pmc and isinstance(out.get("prompt_pass"), dict) and "roofline" in out["prompt_pass"]:
        try:
            out["prompt_pass"]["roofline"]["mfma_busy"] = pmc_mfma_busy(path, out["prompt_pass"]["ms_per_pass"] * 1e-3)
        except Exception as e:
            out["prompt_pass"]["roofline"]["mfma_busy"] = {"error": str(e)[:300]}

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL prints its banner through C stdio: flush that first so that the JSON line is the LAST line on stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
