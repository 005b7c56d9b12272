"""Debug: greedy ids across 257 keys with the two-workgroup 512-key variant on / off (same library, BIOGPT_HIP_XPIPE_DUAL)."""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    sys.path.insert(0, root)
    import numpy as np, _pkg
    pkg = _pkg.load()
    q = os.path.join(os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench"), "synthetic-L24-q4_0.bin")
    m = pkg.BiogptModel.load(q)
    rng = np.random.default_rng(23)
    prompt = [2] + [int(v) for v in rng.integers(4, m.hparams.n_vocab, int(sys.argv[2]) - 1)]
    ids, _ = m.generate_greedy(prompt, int(sys.argv[3]), n_batch=8)
    print(json.dumps([int(v) for v in ids]))
    sys.exit(0)
for npr, ng in ((257, 3), (258, 3), (259, 3), (262, 3), (268, 3), (272, 3), (280, 3), (300, 3)):
    out = []
    for dual in ("0", "1"):
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", str(npr), str(ng)], env=dict(os.environ, BIOGPT_HIP_XPIPE_DUAL=dual), capture_output=True, text=True)
        try: out.append(json.loads(o.stdout.strip().splitlines()[-1]))
        except Exception: print(o.stderr[-500:]); out.append([])
    diff = [i for i in range(min(len(out[0]), len(out[1]))) if out[0][i] != out[1][i]]
    print("prompt", npr, "gen", ng, "first diffs at", diff[:6], "ref", out[0][:16], "dual", out[1][:16])
