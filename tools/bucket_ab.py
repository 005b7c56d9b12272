"""Per-token time of multi-token launches inside ONE context bucket, for several builds: python tools/bucket_ab.py LIB [LIB ...]
prompt lengths 4 / 66 / 130 / 194 + 58 generated tokens = the 64- / 128- / 192- / 256-key variants of dec_xpipe_kernel."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    sys.path.insert(0, root)
    import numpy as np
    import _pkg
    pkg = _pkg.load()
    q = os.path.join(os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench"), "synthetic-L24-%s.bin" % os.environ.get("BUCKET_AB_FTYPE", "q4_0"))
    m = pkg.BiogptModel.load(q)
    rng = np.random.default_rng(5)
    out = {}
    for n_prompt in (4, 66, 130, 194):
        pr = [2] + [int(v) for v in rng.integers(4, m.hparams.n_vocab, n_prompt - 1)]
        m.generate_greedy(pr, 58)
        ts = []
        for _ in range(5):
            _, s58 = m.generate_greedy(pr, 58)
            _, s2 = m.generate_greedy(pr, 2)
            ts.append((s58 - s2) / 56 * 1e6)      # the prompt pass and the first token cancel
        out["%d..%d keys" % (n_prompt + 3, n_prompt + 58)] = round(sorted(ts)[2], 2)
    print(json.dumps(out))
    sys.exit(0)
for rep in range(2):
    for l in sys.argv[1:]:
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm"], env=dict(os.environ, BIOGPT_HIP_LIB=os.path.join(root, l)), capture_output=True, text=True)
        print(l, o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-300:], flush=True)
