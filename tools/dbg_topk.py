import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg
pkg = _pkg.load()
KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)
d = "/tmp/dbg_topk"; os.makedirs(d, exist_ok=True)
f32, q = d + "/f32.bin", d + "/q4_0.bin"
if not os.path.exists(q):
    pkg.write_synthetic(f32, seed=77, **KW); pkg.quantize_file(f32, q, "q4_0")
k = 1
mode = sys.argv[1] if len(sys.argv) > 1 else "topk"
g = pkg.BiogptModel.load(q)
for kv in os.environ.get('UENV','').split(','):
    if '=' in kv: os.environ[kv.split('=')[0]] = kv.split('=')[1]
u = pkg.BiogptModel.load(q)
for kv in os.environ.get('UENV','').split(','):
    if '=' in kv: os.environ.pop(kv.split('=')[0])
rng = np.random.default_rng(101)
prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 245)]
g.eval_device(prompt, 0); u.eval_device(prompt, 0)
toks, rows_g, rows_u = [77], [], []
for n_past in range(246, 256):
    tok = toks[-1]
    if mode == "topk":
        vg, ig = g.eval_topk([tok], n_past, k); vu, iu = u.eval_topk([tok], n_past, k)
        rows_g.append(int(ig[0])); rows_u.append(int(iu[0]))
    else:
        a = g.eval([tok], n_past); b = u.eval([tok], n_past); dl = u.read_logits(); print('   u: host row argmax', int(b.argmax()), 'device row argmax', int(dl.argmax()), flush=True)
        rows_g.append(int(a.argmax())); rows_u.append(int(b.argmax())); full_u = globals().setdefault('full_u', []); full_u.append(b)
    toks.append(rows_g[-1])
D=1024; P=1024
kv_u = {(l,pos): u.read_kv(0, (l*P+pos)*D, D) for l in (0,2) for pos in (245,246,247,248)}
kv_g = {(l,pos): g.read_kv(0, (l*P+pos)*D, D) for l in (0,2) for pos in (245,246,247,248)}
g.close(); u.close()
os.environ["BIOGPT_HIP_RESIDENT"] = "0"; os.environ["BIOGPT_HIP_XPIPE"] = "0"
r = pkg.BiogptModel.load(q)
r.eval_device(prompt, 0)
for i, n_past in enumerate(range(246, 256)):
    tr_row = r.eval([toks[i]], n_past)
    t = int(tr_row.argmax())
    if mode != "topk":
        prev = globals().get('prev_row')
        print("   u row vs truth row: %.3g   u row vs PREVIOUS truth row: %s" % (float(np.abs(full_u[i] - tr_row).max()), "n/a" if prev is None else "%.3g" % float(np.abs(full_u[i] - prev).max())))
        prev_row = tr_row
    print(n_past, "truth", t, "g", rows_g[i], "u", rows_u[i], "" if t == rows_g[i] == rows_u[i] else "  <-- MISMATCH", flush=True)

for key in sorted(kv_u):
    tr = r.read_kv(0, (key[0]*P+key[1])*D, D)
    print("K row layer %d pos %d: u max|d| %.3g  g max|d| %.3g  (|truth| max %.3g)" % (key[0], key[1], float(np.abs(kv_u[key]-tr).max()), float(np.abs(kv_g[key]-tr).max()), float(np.abs(tr).max())))
