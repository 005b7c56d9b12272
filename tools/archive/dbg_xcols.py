"""Chunk evals (2 .. 8 tokens) through the column-per-XCD launch (kernels_xcols.hip.h) against the launch chain (BIOGPT_HIP_XCOLS=0): logits and K / V rows, then
the time per 8-token eval at 24 layers.   python tools/dbg_xcols.py [ftype ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg  # noqa: E402

pkg = _pkg.load()
d = os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench")
os.makedirs(d, exist_ok=True)
KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)


def opts(g, **env):
    for k, v in env.items():
        os.environ[k] = v
    g.refresh_options()
    for k in env:
        del os.environ[k]


def check(ftype):
    f32, q = os.path.join(d, "xc-L3-f32.bin"), os.path.join(d, "xc-L3-%s.bin" % ftype)
    if not os.path.exists(q):
        if not os.path.exists(f32):
            pkg.write_synthetic(f32, **KW)
        pkg.quantize_file(f32, q, ftype)
    g = pkg.BiogptModel.load(q)
    rng = np.random.default_rng(5)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 300)]
    n_past, bad = 0, 0
    sizes = [8, 2, 3, 8, 5, 7, 8, 4, 6, 8] * 4
    for n in sizes:
        if n_past + n > 256:
            break
        chunk = toks[n_past:n_past + n]
        rows = []
        for arm in ("1", "0"):
            opts(g, BIOGPT_HIP_XCOLS=arm)
            lg = g.eval(chunk, n_past)
            kv = [g.read_kv(w, (2 * KW["n_positions"] + n_past) * KW["d_model"], n * KW["d_model"]) for w in (0, 1)]
            rows.append((lg, kv))
        (la, ka), (lb, kb) = rows
        same = bool((la == lb).all() and (ka[0] == kb[0]).all() and (ka[1] == kb[1]).all())
        if not same:
            bad += 1
            print("%s n_past %3d n %d: logits max diff %g, k %g, v %g   xpipe_state %d" % (ftype, n_past, n, np.abs(la - lb).max(), np.abs(ka[0] - kb[0]).max(), np.abs(ka[1] - kb[1]).max(), g.xpipe_state()))
        n_past += n
    print("%s: %d chunk evals up to %d keys, %d differ, xpipe_state %d" % (ftype, len(sizes), n_past, bad, g.xpipe_state()))
    g.close()


def timing():
    f32, q = os.path.join(d, "synthetic-L24-f32.bin"), os.path.join(d, "synthetic-L24-q4_0.bin")
    if not os.path.exists(q):
        pkg.write_synthetic(f32)
        pkg.quantize_file(f32, q, "q4_0")
        os.remove(f32)
    rng = np.random.default_rng(9)
    toks = [2] + [int(v) for v in rng.integers(4, 42384, 255)]
    for arm in ("1", "0"):
        os.environ["BIOGPT_HIP_XCOLS"] = arm
        m = pkg.BiogptModel.load(q)
        del os.environ["BIOGPT_HIP_XCOLS"]
        for nb in (8, 4, 2):
            for rep in range(3):
                t0 = time.perf_counter()
                for at in range(0, 256, nb):
                    m.eval(toks[at:at + nb], at)
                t = time.perf_counter() - t0
            print("XCOLS=%s: %d evals of %d tokens (0 .. 256 keys): %.3f ms per eval, %.0f prompt tok/s, xpipe_state %d" % (arm, 256 // nb, nb, t / (256 // nb) * 1e3, 256 / t, m.xpipe_state()))
        for rep in range(2):      # the same without the row: biogpt_hip_eval_device + one synchronisation at the end
            t0 = time.perf_counter()
            for at in range(0, 256, 8):
                m.eval_device(toks[at:at + 8], at)
            m.synchronize()
            t = time.perf_counter() - t0
        print("XCOLS=%s: 32 asynchronous evals of 8 tokens: %.3f ms per eval" % (arm, t / 32 * 1e3))
        m.close()


for ft in (sys.argv[1:] or ["q4_0"]):
    check(ft)
timing()
