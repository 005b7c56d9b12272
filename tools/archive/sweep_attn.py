import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import _pkg
    m = _pkg.load()
    g = m.BiogptModel.load(sys.argv[2])
    out = {}
    for n_past in (40, 103, 200, 400, 1000):
        s, b = g.bench_matvec(5, n_past, 240)
        out[n_past] = round(s * 1e6, 2)
    print(json.dumps(out))
    sys.exit(0)
for cfg in [dict()] + [dict(BIOGPT_HIP_DBG=str(d)) for d in (1, 2, 3, 4, 8, 16, 31)]:
    r = subprocess.run([sys.executable, __file__, "child", sys.argv[1]], env=dict(os.environ, **cfg), capture_output=True, text=True)
    print(cfg, r.stdout.strip() or r.stderr[-300:], flush=True)
