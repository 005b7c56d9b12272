"""Batched generation of 2 .. 8 sequences with the decode steps as column-per-XCD launches (kernels_xcols.hip.h, streams mode) against the launch chain
(BIOGPT_HIP_XCOLS=0): ids, then tokens/s at 24 layers.   python tools/dbg_streams.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
d = os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench")
os.makedirs(d, exist_ok=True)
KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)


def load(q, arm):
    os.environ["BIOGPT_HIP_XCOLS"] = arm
    m = pkg.BiogptModel.load(q)
    del os.environ["BIOGPT_HIP_XCOLS"]
    return m


def check(ftype):
    f32, q = os.path.join(d, "xc-L3-f32.bin"), os.path.join(d, "xc-L3-%s.bin" % ftype)
    if not os.path.exists(q):
        if not os.path.exists(f32):
            pkg.write_synthetic(f32, **KW)
        pkg.quantize_file(f32, q, ftype)
    rng = np.random.default_rng(3)
    for S, lens, n_predict in ((8, (4, 9, 17, 5, 8, 30, 2, 11), 60), (3, (60, 61, 7), 80), (2, (5, 5), 260), (8, (250,) * 8, 20)):
        prompts = [[2] + [int(v) for v in rng.integers(4, KW["n_vocab"], n - 1)] for n in lens]
        out = []
        for arm in ("1", "0"):
            m = load(q, arm)
            ids, _ = m.generate_greedy_batch(prompts, n_predict, n_batch=8)
            out.append((np.asarray(ids), m.chunk_launches(), m.xpipe_state()))
            m.close()
        same = bool((out[0][0] == out[1][0]).all())
        print("%s S=%d lens %s n_predict %d: ids equal %s, launches %d / %d, state %d / %d" % (ftype, S, lens[:4], n_predict, same, out[0][1], out[1][1], out[0][2], out[1][2]))


def timing():
    f32, q = os.path.join(d, "synthetic-L24-f32.bin"), os.path.join(d, "synthetic-L24-q4_0.bin")
    if not os.path.exists(q):
        pkg.write_synthetic(f32)
        pkg.quantize_file(f32, q, "q4_0")
        os.remove(f32)
    rng = np.random.default_rng(4)
    for S in (8, 4, 2):
        prompts = [[2] + [int(v) for v in rng.integers(4, 42384, 3)] for _ in range(S)]
        for arm in ("1", "0"):
            m = load(q, arm)
            m.generate_greedy_batch(prompts, 8, n_batch=8)
            ids, secs = m.generate_greedy_batch(prompts, 200, n_batch=8)
            print("XCOLS=%s S=%d: %.1f tok/s, %.3f ms per step of all sequences, launches %d" % (arm, S, S * 200 / secs, secs / 200 * 1e3, m.chunk_launches()))
            m.close()


for ft in (sys.argv[1:] or ["q4_0", "q5_1"]):
    check(ft)
timing()
