"""Quick A/B of several builds of the library (interleaved processes, same box), round 4:
    python tools/ab_quick.py [--reps R] [--points 103,300,1023] LIB[:ENV=VALUE,...] [LIB ...]
Each arm (own process, BIOGPT_HIP_LIB): the headline workload (200-token greedy continuation of a 4-token prompt, 10 continuations after a warm-up) and the
graph-replayed single-token step at the given n_past points.  Prints one line per arm and run; the ids of the continuation are compared between the arms."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    sys.path.insert(0, root)
    import time
    import numpy as np
    import _pkg
    pkg = _pkg.load()
    d = os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench")
    os.makedirs(d, exist_ok=True)
    ft = os.environ.get("AB_FTYPE", "q4_0")
    f32, q = os.path.join(d, "synthetic-L24-f32.bin"), os.path.join(d, "synthetic-L24-%s.bin" % ft)
    if not os.path.exists(q):
        if not os.path.exists(f32):
            pkg.write_synthetic(f32)
        pkg.quantize_file(f32, q, ft)
    m = pkg.BiogptModel.load(q)
    points = [int(v) for v in sys.argv[2].split(",") if v]
    rng = np.random.default_rng(1000)
    prompt = [2] + [int(v) for v in rng.integers(4, m.hparams.n_vocab, 3)]
    ids, _ = m.generate_greedy(prompt, 200)
    t0 = time.perf_counter()
    for _ in range(10):
        m.generate_greedy(prompt, 200)
    m.synchronize()
    dt = (time.perf_counter() - t0) / 10
    out = {"value": round(200 / dt, 1), "ids": [int(v) for v in ids], "xpipe": m.xpipe_state()}
    for n_past in points:
        out["T=%d" % (n_past + 1)] = round(m.bench_decode(n_past, reps=100) * 1e6, 2)
    print(json.dumps(out))
    sys.exit(0)

args = sys.argv[1:]
reps, points = 2, "103,300,1023"
while args and args[0].startswith("--"):
    if args[0] == "--reps":
        reps = int(args[1])
    elif args[0] == "--points":
        points = args[1]
    args = args[2:]
libs = args
ids0 = None
res = {l: [] for l in libs}
for r in range(reps):
    for l in libs:
        lib, _, extra = l.partition(":")      # an arm is LIB or LIB:NAME=VALUE,NAME=VALUE (environment of that arm)
        env = dict(os.environ, BIOGPT_HIP_LIB=os.path.join(root, lib))
        for kv in extra.split(","):
            if "=" in kv:
                env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", points], env=env, capture_output=True, text=True)
        try:
            dct = json.loads(o.stdout.strip().splitlines()[-1])
        except Exception:
            print(l, "FAILED", o.stderr[-800:])
            continue
        ids = dct.pop("ids")
        if ids0 is None:
            ids0 = ids
        dct["ids_equal_first_arm"] = ids == ids0
        res[l].append(dct)
        print(l, json.dumps(dct), flush=True)
print("---- summary (means)")
for l in libs:
    if res[l]:
        keys = [k for k in res[l][0] if k.startswith("T=") or k == "value"]
        print(l, {k: round(sum(x[k] for x in res[l]) / len(res[l]), 2) for k in keys}, "ids equal:", all(x["ids_equal_first_arm"] for x in res[l]))
