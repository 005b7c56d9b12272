"""Single-token decode of FLOAT weight files (F32 / F16, biogpt.cpp:160-165) through the persistent launch of csrc/kernels_fpipe.hip.h -- all layers in ONE launch, weights
register-resident a layer ahead, stage inputs collected by a polling wave -- against (a) the five-launch layer it replaces (BIOGPT_HIP_FPIPE=0) bit for bit, logits and the
appended K / V rows, and (b) the oracle within the contract; across its context limit (224 keys), the device-resident greedy loop, and the full 24-layer F16 model."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-3
KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)


@pytest.fixture(scope="module")
def files(pkg, tmp_path_factory):
    from modelfile_py import read_model, write_model
    d = tmp_path_factory.mktemp("fpipe")
    f32, f16 = str(d / "f32.bin"), str(d / "f16.bin")
    pkg.write_synthetic(f32, seed=91, **KW)
    hp, vocab, merges, tensors = read_model(f32)           # convert.py --use-f16: the 2-D "*.weight" tensors as float16, ftype 1
    for t in tensors:
        if len(t["ne"]) == 2 and t["name"].endswith(".weight") and t["type"] == 0:
            t["raw"] = np.frombuffer(t["raw"], dtype=np.float32).astype(np.float16).tobytes()
            t["type"] = 1
    write_model(f16, dict(hp, ftype=1), vocab, merges, tensors)
    return {"f32": f32, "f16": f16}


def _kv(g, n_past):
    out = []
    for l in (0, KW["n_layer"] - 1):
        out += [g.read_kv(w, (l * KW["n_positions"] + n_past) * KW["d_model"], KW["d_model"]) for w in (0, 1)]
    return out


@pytest.mark.parametrize("name", ["f32", "f16"])
def test_persistent_float_decode_equals_the_five_launch_layer_and_the_oracle(pkg, oracle, files, monkeypatch, name):
    monkeypatch.setenv("BIOGPT_HIP_FPIPE", "0")
    ref = pkg.BiogptModel.load(files[name])
    monkeypatch.setenv("BIOGPT_HIP_FPIPE", "1")
    g = pkg.BiogptModel.load(files[name])
    monkeypatch.delenv("BIOGPT_HIP_FPIPE")
    assert ref.fpipe_launches() == -1
    if g.fpipe_launches() < 0:
        pytest.skip("the persistent float-weight launch is not available on this device")
    o = oracle.OracleModel(files[name], n_threads=16)
    rng = np.random.default_rng(7)
    ctx = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 239)]
    worst, exact, steps = 0.0, 0, 0

    def step(tok, n_past):
        nonlocal worst, exact, steps
        lg, lr, lo = g.eval([tok], n_past), ref.eval([tok], n_past), o.eval([tok], n_past)
        assert (lg == lr).all(), "%s n_past %d: persistent launch != five-launch layer (max diff %g)" % (name, n_past, np.abs(lg - lr).max())
        for a, b in zip(_kv(g, n_past), _kv(ref, n_past)):
            assert (a == b).all(), (name, n_past)
        worst = max(worst, float(np.abs(lg - lo).max()))
        exact += int((lg == lo).all())
        steps += 1
        assert int(lg.argmax()) == int(lo.argmax())

    # a prompt chunk, then token by token: the first positions, the 64-key border ...
    for m in (g, ref, o):
        m.eval(ctx[:8], 0)
    for n_past in range(8, 20):
        step(ctx[n_past], n_past)
    for m in (g, ref, o):
        m.eval(ctx[20:60], 20)
    for n_past in range(60, 70):
        step(ctx[n_past], n_past)
    # ... and across the launch's context limit (224 keys): 225 keys and more take the five-launch layer
    for m in (g, ref, o):
        m.eval(ctx[70:216], 70)
    for n_past in range(216, 232):
        step(ctx[n_past], n_past)
    print("%s: %d single-token steps, worst |diff| vs oracle %.2e, %d bit-identical; %d of them through the persistent launch" % (name, steps, worst, exact, g.fpipe_launches()))
    assert worst <= ATOL
    assert g.fpipe_launches() == 12 + 10 + 8          # n_past 8 .. 19, 60 .. 69, 216 .. 223: contexts of up to 224 keys
    g.close(); ref.close()


@pytest.mark.parametrize("name", ["f32", "f16"])
def test_float_generation_in_the_device_loop(pkg, oracle, files, name):
    """generate_greedy (captured steps: embedding launch + the persistent launch + lm_head + sampler per token) == the oracle's ids, and twice the same."""
    g = pkg.BiogptModel.load(files[name])
    prompt = [2, 7548, 1171, 32924, 17]
    ids, _ = g.generate_greedy(prompt, 40, n_batch=8)
    ref, _ = oracle.OracleModel(files[name], n_threads=16).generate_greedy(prompt, 40, n_batch=8)
    assert list(ids) == list(ref)
    ids2, _ = g.generate_greedy(prompt, 40, n_batch=8)
    assert list(ids2) == list(ref)
    assert g.fpipe_launches() != 0
    g.close()


def test_full_model_f32_24_layers(pkg, oracle, tmp_path_factory, monkeypatch):
    """configs[0]'s file format at BioGPT-base size (24 layers, 1.4 GB of F32 weights, README.md:24,45 / main.cpp:160): teacher-forced single-token steps around 100 and 200 keys
    through the persistent launch == the five-launch layer bit for bit, the oracle within the contract, the same argmax."""
    d = tmp_path_factory.mktemp("fpipe24")
    path = str(d / "f32-24.bin")
    kw = dict(KW, n_layer=24)
    pkg.write_synthetic(path, seed=92, **kw)
    monkeypatch.setenv("BIOGPT_HIP_FPIPE", "0")
    ref = pkg.BiogptModel.load(path)
    monkeypatch.delenv("BIOGPT_HIP_FPIPE")
    g = pkg.BiogptModel.load(path)
    if g.fpipe_launches() < 0:
        pytest.skip("the persistent float-weight launch is not available on this device")
    o = oracle.OracleModel(path, n_threads=16)
    rng = np.random.default_rng(11)
    ctx = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 211)]
    worst, exact, steps = 0.0, 0, 0
    for lo_, hi_ in ((0, 96), (104, 196)):
        for m in (g, ref, o):
            m.eval(ctx[lo_:hi_], lo_)
        for n_past in range(hi_, hi_ + 8):
            lg, lr, lo = g.eval([ctx[n_past]], n_past), ref.eval([ctx[n_past]], n_past), o.eval([ctx[n_past]], n_past)
            assert (lg == lr).all(), "n_past %d: persistent launch != five-launch layer (max diff %g)" % (n_past, np.abs(lg - lr).max())
            worst = max(worst, float(np.abs(lg - lo).max()))
            exact += int((lg == lo).all())
            steps += 1
            assert int(lg.argmax()) == int(lo.argmax())
    print("F32 x 24 layers: %d steps, worst |diff| vs oracle %.2e, %d bit-identical" % (steps, worst, exact))
    assert worst <= ATOL and g.fpipe_launches() == 16
    g.close(); ref.close()


def test_a_persistent_float_launch_that_never_completes_is_repeated_on_five_launches(pkg, files, monkeypatch, capfd):
    """BIOGPT_HIP_FPIPE_FAULT=1: in the context's first persistent launch one workgroup withholds its out_proj rows of layer 0 -- what a launch whose workgroups are not all
    resident looks like.  The launch must drain on its bounded spins, the call must be repeated transparently on the five-launch layer with the undisturbed logits and K / V
    rows, and the context must stay off the persistent launch."""
    ref = pkg.BiogptModel.load(files["f32"])
    if ref.fpipe_launches() < 0:
        pytest.skip("the persistent float-weight launch is not available on this device")
    prompt = [2, 100, 200, 300, 400, 500]
    ref.eval(prompt, 0)
    want = ref.eval([7], 6)
    want_kv = _kv(ref, 6)
    want2 = ref.eval([9], 7)
    ref.close()
    monkeypatch.setenv("BIOGPT_HIP_FPIPE_FAULT", "1")
    g = pkg.BiogptModel.load(files["f32"])
    monkeypatch.delenv("BIOGPT_HIP_FPIPE_FAULT")
    assert g.fpipe_launches() == 0
    g.eval(prompt, 0)
    got = g.eval([7], 6)
    assert (got == want).all()
    for a, b in zip(_kv(g, 6), want_kv):
        assert (a == b).all()
    assert g.fpipe_launches() == -1
    assert "five-launch layer" in capfd.readouterr().err
    assert (g.eval([9], 7) == want2).all()
    g.close()
