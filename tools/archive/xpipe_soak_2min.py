import sys, time; sys.path.insert(0, '.')
import _pkg
m=_pkg.load(); g=m.BiogptModel.load("/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin")
pr=[2,100,200,300]
want,_=g.generate_greedy(pr,200)
bad=0; t0=time.time(); n=0; worst=0
while time.time()-t0 < 120:
    got,secs=g.generate_greedy(pr,200); n+=1; worst=max(worst,secs)
    bad+=int(list(got)!=list(want))
print("generations", n, "mismatches", bad, "state", g.xpipe_state(), "mean tok/s %.1f" % (n*200/(time.time()-t0)), "slowest call %.2f ms" % (worst*1e3))
