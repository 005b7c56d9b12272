"""Ablation sweep of the mat-vec kernel on the GPU box (profiling aid, not part of the product)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import _pkg
    m = _pkg.load()
    g = m.BiogptModel.load(sys.argv[2])
    out = {}
    for which, name in ((0, "fc1"), (1, "fc2"), (2, "qkv"), (3, "o"), (4, "lm")):
        s, b = g.bench_matvec(which, 0, 240 if which != 4 else 40)
        out[name] = round(s * 1e6, 2)
    print(json.dumps(out))
    sys.exit(0)
path = sys.argv[1]
configs = [dict()] + [dict(BIOGPT_HIP_DBG=str(d)) for d in (1, 2, 8, 16, 27)] + \
          [dict(BIOGPT_HIP_MAX_WGS=str(t)) for t in (512, 256, 128, 64)] + \
          [dict(BIOGPT_HIP_MAX_WGS=str(t), BIOGPT_HIP_MV_WAVES="2") for t in (512, 256, 128)] + \
          [dict(BIOGPT_HIP_TREE_REDUCE="1")]
for cfg in configs:
    env = dict(os.environ, **cfg)
    r = subprocess.run([sys.executable, __file__, "child", path], env=env, capture_output=True, text=True)
    print(cfg, r.stdout.strip() or r.stderr[-300:], flush=True)
