"""PMC target (tools): three 58-token greedy generations after a 4-token prompt = multi-token launches of the 64-key variant of dec_xpipe_kernel only.
usage: rocprofv3 --pmc COUNTER --kernel-trace -d DIR -o p -- python tools/pmc_bucket_target.py   (BIOGPT_HIP_XPIPE_AS_RES=1: through the RES instantiations)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
m = pkg.BiogptModel.load(os.path.join(os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench"), "synthetic-L24-q4_0.bin"), verbosity=0)
for _ in range(3):
    ids, s = m.generate_greedy([2, 100, 200, 300], 58)
print("us/token", s / 58 * 1e6)
