"""Eager (no hipGraph) single-token evals at a fixed context, for rocprofv3 kernel traces of the decode chain:
    BIOGPT_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -- python tools/eager_decode_target.py MODEL N_PAST [REPS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
m = _pkg.load()
g = m.BiogptModel.load(sys.argv[1], verbosity=0)
n_past, reps = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 20
for _ in range(reps):
    g.eval_device([7], n_past)
g.synchronize()
print("done", n_past, reps)
