"""Small workload for rocprofv3 --pmc passes: every kernel of the decode step back to back (graph replays of one sweep
over the layers), plus the lm_head and -- with `prefill` as second argument -- one 512-column prompt pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg
m = _pkg.load()
g = m.BiogptModel.load(sys.argv[1])
if len(sys.argv) > 2 and sys.argv[2] == "prefill":
    rng = np.random.default_rng(7000)
    toks = [2] + [int(v) for v in rng.integers(4, g.hparams.n_vocab, 511)]
    g.eval_prompt(toks, 0, 8, want_logits=False); g.synchronize()
    g.eval_prompt(toks, 0, 8, want_logits=False); g.synchronize()
elif len(sys.argv) > 2 and sys.argv[2] == "long":      # the pipelined launch of a 1024-key step (kernels_xlong.hip.h): graph replays at n_past = 1023
    print("T=1024", round(g.bench_decode(1023, 40) * 1e6, 2), "us per token", flush=True)
elif len(sys.argv) > 2 and sys.argv[2] == "chunk":     # the column-per-XCD chunk launch (kernels_xcols.hip.h): 8-token evals at 0 .. 64 keys and at 296 .. 360 keys
    rng = np.random.default_rng(7000)
    toks = [2] + [int(v) for v in rng.integers(4, g.hparams.n_vocab, 63)]
    for rep in range(3):
        for at in range(0, 64, 8):
            g.eval(toks[at:at + 8], at)
    more = [2] + [int(v) for v in rng.integers(4, g.hparams.n_vocab, 359)]      # ... and at 296 .. 360 keys: the 512-key variant of the launch
    g.eval_prompt(more[:296], 0, 8)
    for rep in range(3):
        for at in range(296, 360, 8):
            g.eval(more[at:at + 8], at)
    print("chunk launches", g.chunk_launches(), flush=True)
elif len(sys.argv) > 2 and sys.argv[2] == "dual":      # the two-workgroups-per-head launch at 400 keys: graph replays at n_past = 399
    print("T=400", round(g.bench_decode(399, 40) * 1e6, 2), "us per token", flush=True)
elif len(sys.argv) > 2 and sys.argv[2] == "sweep":     # matvec_sweep_kernel: every block-quantized matrix of the model in one launch (kernels_sweep.hip.h)
    s, b, c = g.bench_sweep(40)
    print("sweep", round(s * 1e6, 2), "us", b, "bytes", round(b / s / 1e9, 1), "GB/s, check", c, flush=True)
elif len(sys.argv) > 2 and sys.argv[2] == "headline":  # the headline's own work and nothing else: 200-token greedy continuations of 4-token prompts (bench.py's timed step), multi-token launches
    rng = np.random.default_rng(1000)
    for k in range(6):
        pr = [2] + [int(v) for v in rng.integers(4, g.hparams.n_vocab, 3)]
        ids, secs = g.generate_greedy(pr, 200, n_batch=8)
        print("continuation", k, round(secs * 1e3, 3), "ms", g.generate_launches(), flush=True)
elif len(sys.argv) > 2 and sys.argv[2] == "xpipe":     # only the XCD-pipelined single-token launch at 104 keys (bench.py's roofline.traffic pass)
    if g.xpipe_state() == 1:
        s, b = g.bench_matvec(11, 0, 24)
        print(11, round(s * 1e6, 2), "us", b, "bytes", flush=True)
else:
    for which in (6, 7, 8, 9, 10, 4) + ((11,) if g.xpipe_state() == 1 else ()):   # 11: the XCD-pipelined launch (all layers + lm_head)
        s, b = g.bench_matvec(which, 0, 48)
        print(which, round(s * 1e6, 2), "us", b, "bytes", flush=True)
