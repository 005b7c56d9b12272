"""Round 5: the arrangement of profiles/two_contexts_r4.txt, long: context g holds the device's pipeline slot with its resident launch, context u replays its captured
five-launch step beside it (BIOGPT_HIP_GRAPH_CONTENDED=1 switches round 4's detour to eager launches off).  Same tokens for both, so every row of u must equal g's,
bit for bit.  Prints wrong rows and what the lineage check (biogpt_hip_lineage_stats) found.   usage: python tools/soak_two_contexts_r5.py [calls] [pause_us]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg
pkg = _pkg.load()
KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)
d = "/tmp/dbg_topk"; os.makedirs(d, exist_ok=True)
f32, q = d + "/f32.bin", d + "/q4_0.bin"
if not os.path.exists(q):
    pkg.write_synthetic(f32, seed=77, **KW); pkg.quantize_file(f32, q, "q4_0")
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 600
pause = float(sys.argv[2]) * 1e-6 if len(sys.argv) > 2 else 0.0
g = pkg.BiogptModel.load(q)
os.environ["BIOGPT_HIP_GRAPH_CONTENDED"] = "1"
u = pkg.BiogptModel.load(q)
del os.environ["BIOGPT_HIP_GRAPH_CONTENDED"]
rng = np.random.default_rng(101)
wrong = wrong_prev = 0
done = 0
while done < calls:
    prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 40)]
    g.eval_device(prompt, 0); u.eval_device(prompt, 0)
    tok, prev = 77, None
    for n_past in range(41, 41 + min(200, calls - done)):
        a = g.eval([tok], n_past)
        if pause: time.sleep(pause)
        b = u.eval([tok], n_past) if (n_past & 3) else None
        if b is None:
            vals, ids = u.eval_topk([tok], n_past, 1)
            ok = int(ids[0]) == int(a.argmax())
        else:
            ok = bool((a == b).all())
            if not ok and prev is not None and (b == prev).all(): wrong_prev += 1
        wrong += 0 if ok else 1
        prev = a
        tok = int(a.argmax()); done += 1
try:
    st = u.lineage_stats()
except Exception:
    st = None
print("%d calls per context (pause %g us between the two): %d rows of the second context differ from the first's (%d of them = the previous call's row); lineage check: %s"
      % (done, pause * 1e6, wrong, wrong_prev, st))
g.close(); u.close()
