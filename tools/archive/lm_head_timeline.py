"""Stage stamps of the stand-alone lm_head launch (profiling build): BIOGPT_HIP_DBG=32 BIOGPT_HIP_LIB=.../libbiogpt_hip_prof.so python tools/lm_head_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
d = os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench")
m = pkg.BiogptModel.load(os.path.join(d, "synthetic-L24-%s.bin" % (sys.argv[1] if len(sys.argv) > 1 else "q4_0")), verbosity=0)
s, b = m.bench_matvec(4, 0, 50)
print("lm_head %.2f us per launch" % (s * 1e6))
