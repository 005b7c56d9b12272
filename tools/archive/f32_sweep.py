"""F32 / F16 decode step vs the generic mat-vec launch shape (env switches are read at load): python tools/f32_sweep.py [ftype]"""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ft = sys.argv[1] if len(sys.argv) > 1 else "f32"
code = "import sys; sys.path.insert(0, %r); import _pkg; m=_pkg.load(); g=m.BiogptModel.load(%r, verbosity=0); g.generate_greedy([2,5,6,7],8); print(round(g.bench_decode(103, reps=50)*1e6,1))"
path = os.path.join(os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench"), "synthetic-L24-%s.bin" % ft if ft != "f32" else "synthetic-L24-f32.bin")
for env in ({}, {"BIOGPT_HIP_MAX_WGS": "256"}, {"BIOGPT_HIP_MAX_WGS": "512"}, {"BIOGPT_HIP_MAX_WGS": "2048"}, {"BIOGPT_HIP_MV_WAVES": "8"}, {"BIOGPT_HIP_MV_WAVES": "2"},
            {"BIOGPT_HIP_MV_WAVES": "8", "BIOGPT_HIP_MAX_WGS": "512"}, {"BIOGPT_HIP_TARGET_WGS": "512"}, {"BIOGPT_HIP_TARGET_WGS": "1024"}):
    r = subprocess.run([sys.executable, "-c", code % (root, path)], env=dict(os.environ, **env), capture_output=True, text=True)
    print(env, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
