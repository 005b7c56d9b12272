timeout 900 python - <<'PY'
import os, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import _pkg
m=_pkg.load()
KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)
q='/tmp/dbg/q4_0.bin'
if not os.path.exists(q):
    os.makedirs('/tmp/dbg', exist_ok=True); m.write_synthetic('/tmp/dbg/f32.bin', **KW); m.quantize_file('/tmp/dbg/f32.bin', q, 'q4_0')
rng = np.random.default_rng(17)
toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 40)]
os.environ["BIOGPT_HIP_XPIPE"]="0"; g5=m.BiogptModel.load(q); del os.environ["BIOGPT_HIP_XPIPE"]
ref={}
for n in range(8): ref[n]=g5.eval([toks[n]], n)
def run(name, seq):
    g=m.BiogptModel.load(q)
    out=[]
    for op in seq:
        if op=="r": g.refresh_options(); out.append("r")
        elif isinstance(op, tuple): g.eval_device([toks[i] for i in range(op[0], op[1])], op[0]); g.synchronize(); out.append("p")
        else: out.append("%d:%.3g" % (op, float(np.abs(g.eval([toks[op]], op)-ref[op]).max())))
    print(name, " ".join(out), flush=True); g.close()
run("V1 same position thrice", [0,0,0])
run("V2 refresh between", [0,"r",1])
run("V3 prompt 0..4 then 5,6", [(0,5),5,6])
run("V4 0,1,2", [0,1,2])
run("V5 0,1 refresh 2,3", [0,1,"r",2,3])
run("V6 prompt 0..1 then 1.. (rewrite)", [(0,2),1,2])
PY
