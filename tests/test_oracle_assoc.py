"""What "within 1e-3 of the reference's logits" can mean: the reference's arithmetic is ggml's CPU backend, whose f32
association and activation rounding differ between its scalar fallbacks and its SIMD kernels, so the
reference itself is not one function of its inputs.  The oracle restates both shapes (biogpt_oracle.h, bo_opts.assoc);
this test measures the envelope between them on the 3-layer BioGPT-base-width model the GPU tests use.  The HIP kernels
are compared with assoc = 0 (bit-identical); DESIGN.md section 3 quotes the numbers printed here."""
import numpy as np
import pytest

KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)


@pytest.fixture(scope="module")
def files(pkg, tmp_path_factory):
    d = tmp_path_factory.mktemp("assoc")
    f32 = str(d / "f32.bin")
    pkg.write_synthetic(f32, **KW)                    # host-side writer / quantizer of the build: no GPU needed
    out = {"f32": f32}
    for name in ("q4_0", "q8_0"):
        out[name] = str(d / (name + ".bin"))
        pkg.quantize_file(f32, out[name], name)
    return out


def _envelope(oracle, path, assoc, steps=24):
    a = oracle.OracleModel(path, n_threads=8, assoc=0)
    b = oracle.OracleModel(path, n_threads=8, assoc=assoc)
    rng = np.random.default_rng(11)
    prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 7)]
    la, lb = a.eval(prompt, 0), b.eval(prompt, 0)
    worst, agree, n_past = float(np.abs(la - lb).max()), int(la.argmax() == lb.argmax()), 8
    for _ in range(steps):                            # teacher-forced with the scalar mode's greedy ids
        t = int(la.argmax())
        la, lb = a.eval([t], n_past), b.eval([t], n_past)
        worst = max(worst, float(np.abs(la - lb).max()))
        agree += int(la.argmax() == lb.argmax())
        n_past += 1
    return worst, agree, steps + 1, float(np.abs(la).max())


@pytest.mark.parametrize("name", ["q4_0", "q8_0", "f32"])
def test_envelope_between_ggml_association_modes(oracle, files, name):
    rows = []
    for assoc, what in ((1, "8-lane fma dots"), (2, "nearest-even activations, id = 127/amax"), (3, "both (AVX2 shape)")):
        worst, agree, n, scale = _envelope(oracle, files[name], assoc)
        rows.append((assoc, worst, agree, n))
        print("%s  assoc %d (%s): max |dlogit| vs scalar mode %.3e (logit scale %.2f), arg-max equal %d/%d" % (name, assoc, what, worst, scale, agree, n))
    # the modes are the same model: the envelope is a fraction of the logit scale, and not zero (they do differ)
    assert all(0.0 < w < 0.15 for a, w, _, _ in rows if not (name == "f32" and a == 2))
    if name == "f32":
        assert rows[1][1] == 0.0                      # no activation quantization with float weights
        assert rows[0][1] < 5e-3
    # quantized weights: even the pure association change exceeds the 1e-3 logit contract of north_star, because a
    # 1-ulp difference can flip an int8 code of a downstream Q8 activation block -- documented, not asserted as a bar


@pytest.mark.parametrize("name", ["q4_0", "q8_0", "f32"])
def test_avx2_intrinsics_equal_their_scalar_emulation(oracle, files, name):
    """bench.py's SIMD cpu_baseline runs the AVX2 shape with real intrinsics (bo_opts.assoc = 7); the scalar emulation of the same
    shape (assoc = 3) is what this file measures the envelope with.  Same arithmetic, so the logits must be equal bit for bit."""
    if not oracle.have_avx2():
        pytest.skip("this CPU has no AVX2 + FMA")
    a = oracle.OracleModel(files[name], n_threads=8, assoc=3)
    b = oracle.OracleModel(files[name], n_threads=8, assoc=7)
    rng = np.random.default_rng(5)
    prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 10)]
    la, lb = a.eval(prompt[:8], 0), b.eval(prompt[:8], 0)
    assert (la == lb).all()
    la, lb = a.eval(prompt[8:], 8), b.eval(prompt[8:], 8)
    assert (la == lb).all()
    n_past = len(prompt)
    for _ in range(6):
        t = int(la.argmax())
        la, lb = a.eval([t], n_past), b.eval([t], n_past)
        assert (la == lb).all()
        n_past += 1
    for which in (0, 1):
        assert (a.kv(which) == b.kv(which)).all()


@pytest.mark.parametrize("name", ["q4_1", "q5_0", "q5_1"])
def test_avx2_intrinsics_equal_their_scalar_emulation_other_formats(oracle, pkg, files, tmp_path, name):
    if not oracle.have_avx2():
        pytest.skip("this CPU has no AVX2 + FMA")
    path = str(tmp_path / (name + ".bin"))
    pkg.quantize_file(files["f32"], path, name)
    a = oracle.OracleModel(path, n_threads=8, assoc=3)
    b = oracle.OracleModel(path, n_threads=8, assoc=7)
    la, lb = a.eval([2, 77, 4000, 911, 12, 30000], 0), b.eval([2, 77, 4000, 911, 12, 30000], 0)
    assert (la == lb).all()
    la, lb = a.eval([int(la.argmax())], 6), b.eval([int(lb.argmax())], 6)
    assert (la == lb).all()
