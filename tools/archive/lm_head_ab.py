"""The stand-alone lm_head launch: lm_stream_kernel against matvec_fast_kernel<PRO_LN, EPI_LOGITS> (BIOGPT_HIP_LM_STREAM=0), bench_matvec(4), per format."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import _pkg
    pkg = _pkg.load()
    d = os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench")
    for t in ("q4_0", "q5_1", "q8_0"):
        f = os.path.join(d, "synthetic-L24-%s.bin" % t)
        if not os.path.exists(f):
            continue
        m = pkg.BiogptModel.load(f, verbosity=0)
        best = min(m.bench_matvec(4, 0, 50)[0] for _ in range(5))
        nb = m.bench_matvec(4, 0, 50)[1]
        print("%s lm_head %6.2f us per launch  %5.0f GB/s  %.1f %% of 8 TB/s" % (t, best * 1e6, nb / best / 1e9, nb / best / 8e12 * 100), flush=True)
        m.close()
else:
    for v in ("1", "0"):
        print("BIOGPT_HIP_LM_STREAM=" + v, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, BIOGPT_HIP_LM_STREAM=v))
