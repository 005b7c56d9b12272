"""A/B of two builds of the library on the headline workload (interleaved runs, same box): python tools/ab_value.py LIB_A LIB_B [reps]
Each arm runs in its own process (BIOGPT_HIP_LIB), 3 x (1 warm-up + 10 continuations), alternating."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {l: [] for l in libs}
for r in range(reps):
    for l in libs:
        env = dict(os.environ, BIOGPT_HIP_LIB=os.path.join(root, l), BIOGPT_BENCH_SKIP_TYPES="1")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "1", "--no-cpu-baseline", "--no-pmc"], env=env, capture_output=True, text=True)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res[l].append((d["value"], d["token_roofline"]["T=104"]["us_per_token"], d["api_loop"]["tokens_per_s"]))
for l in libs:
    print(l, " value", [v[0] for v in res[l]], " T=104 us", [v[1] for v in res[l]], " api_loop", [v[2] for v in res[l]])
