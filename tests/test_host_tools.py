"""CPU-side tests of the product library (no GPU, no compute kernels): the C-ABI surface, the
file->file quantizer (quantize.cpp:8-135 + biogpt.cpp:459-621 replacement), the synthetic model writer
and the loader's failure cases that precede device selection."""
import ctypes
import os
import re
import struct

import numpy as np
import pytest

from conftest import ROOT, has_gpu
from modelfile_py import read_model, write_model


def test_c_abi_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "biogpt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(biogpt_hip_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in pkg.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    raw = ctypes.CDLL(pkg.LIB_PATH)
    for name in sorted(declared):
        assert getattr(raw, name) is not None
    assert b"gfx950" in pkg.lib().biogpt_hip_version()


def test_library_contains_gfx950_code_object(pkg):
    blob = open(pkg.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"matvec_kernel" in blob and b"attn_kernel" in blob


@pytest.mark.parametrize("name,ft", [("q4_0", 2), ("q4_1", 3), ("q5_0", 8), ("q5_1", 9), ("q8_0", 7)])
@pytest.mark.parametrize("src", ["f32", "f16"])
def test_quantize_file_matches_oracle_bytes(pkg, oracle, tiny_models, tmp_path, name, ft, src):
    mine, ref = str(tmp_path / "mine.bin"), str(tmp_path / "ref.bin")
    pkg.quantize_file(tiny_models[src], mine, name)
    oracle.quantize_file(tiny_models[src], ref, ft)
    assert open(mine, "rb").read() == open(ref, "rb").read()
    hp, vocab, merges, tensors = read_model(mine)
    hp0, vocab0, merges0, tensors0 = read_model(tiny_models[src])
    assert hp["ftype"] == ft and vocab == vocab0 and merges == merges0   # quantize.cpp:43-122
    qtype = {2: 2, 3: 3, 8: 6, 9: 7, 7: 8}[ft]
    for t, t0 in zip(tensors, tensors0):  # selection rule biogpt.cpp:523
        is_mat = "weight" in t["name"] and len(t["ne"]) == 2
        assert t["type"] == (qtype if is_mat else t0["type"]) and t["ne"] == t0["ne"]


def test_quantize_file_rejects_bad_types(pkg, tiny_models, tmp_path):
    for bad in (0, 1, 4, 5, 6, 10):
        with pytest.raises(pkg.BiogptError):
            pkg.quantize_file(tiny_models["f32"], str(tmp_path / "x.bin"), bad)
    with pytest.raises(pkg.BiogptError):   # already-quantized input (biogpt.cpp:526-528)
        pkg.quantize_file(tiny_models["q4_0"], str(tmp_path / "x.bin"), "q8_0")
    with pytest.raises(pkg.BiogptError):
        pkg.quantize_file(str(tmp_path / "missing.bin"), str(tmp_path / "x.bin"), "q4_0")


def test_synthetic_writer(pkg, oracle, tmp_path):
    kw = dict(n_vocab=96, n_layer=2, n_head=2, n_positions=32, d_ff=128, d_model=64, n_merges=11)
    a, b, c = (str(tmp_path / n) for n in ("a.bin", "b.bin", "c.bin"))
    pkg.write_synthetic(a, seed=7, **kw)
    pkg.write_synthetic(b, seed=7, **kw)
    pkg.write_synthetic(c, seed=8, **kw)
    assert open(a, "rb").read() == open(b, "rb").read() != open(c, "rb").read()
    hp, vocab, merges, tensors = read_model(a)
    assert hp["ftype"] == 0 and len(vocab) == 96 and len(merges) == 11 and len(tensors) == 37
    by = {t["name"]: np.frombuffer(t["raw"], dtype=np.float32) for t in tensors}
    w = by["biogpt.layers.1.fc1.weight"]
    assert abs(w.mean()) < 2e-3 and abs(w.std() - 0.02) < 1.5e-3
    assert abs(by["biogpt.layers.0.self_attn_layer_norm.weight"].mean() - 1.0) < 0.02
    emb = by["biogpt.embed_tokens.weight"].reshape(96, 64)
    assert (emb[1] == 0).all() and np.abs(emb[0]).max() > 0
    pos = [t for t in tensors if t["name"] == "biogpt.embed_positions.weight"][0]
    assert pos["ne"] == [64, 34]
    m = oracle.OracleModel(a)   # the oracle accepts what the product writes
    lg = m.eval(np.array([2, 5, 9], dtype=np.int32), 0)
    assert np.isfinite(lg).all()
    f16 = str(tmp_path / "h.bin")
    pkg.write_synthetic(f16, seed=7, ftype=1, **kw)
    assert read_model(f16)[0]["ftype"] == 1 and oracle.OracleModel(f16).n_tensors == 37


def _expect_load_error(pkg, path, needle):
    with pytest.raises(pkg.BiogptError) as e:
        pkg.BiogptModel.load(path)
    assert needle in str(e.value), str(e.value)


def test_loader_failures_before_device_selection(pkg, tiny_models, tmp_path):
    good = open(tiny_models["f32"], "rb").read()
    _expect_load_error(pkg, str(tmp_path / "nope.bin"), "failed to open")           # biogpt.cpp:35-38
    p = str(tmp_path / "bad.bin")
    open(p, "wb").write(b"\x00\x01\x02\x03" + good[4:])
    _expect_load_error(pkg, p, "bad magic")                                          # biogpt.cpp:44-47
    open(p, "wb").write(good[:32] + struct.pack("<i", 999) + good[36:])
    _expect_load_error(pkg, p, "bad vocab size")                                     # biogpt.cpp:76-80
    open(p, "wb").write(good[:28] + struct.pack("<i", 5) + good[32:])
    _expect_load_error(pkg, p, "bad ftype")                                          # biogpt.cpp:161-165
    open(p, "wb").write(good[:len(good) - 100])
    _expect_load_error(pkg, p, "wrong size")                                         # truncated payload


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback(pkg, tiny_models):
    _expect_load_error(pkg, tiny_models["q4_0"], "no HIP device")


def test_algorithmic_bytes_match_survey(pkg):
    hp = pkg.HParams(**pkg.BIOGPT_BASE)
    hp.ftype = 2
    w = pkg.decode_bytes_per_token(hp, 0) - 2 * 24 * 1024 * 4 - 42384 * 4
    assert int(w) == 195569792                      # W(Q4_0)
    assert pkg.decode_bytes_per_token(hp, 1024) - pkg.decode_bytes_per_token(hp, 0) == 196608 * 1024
    assert 209 < pkg.arena_bytes_for(hp) / 2 ** 20 < 213   # tensor bytes 210.35 MiB + padding + tables


def test_pass_kernels_use_no_scratch(pkg, tmp_path):
    """The many-column kernels sit at their register bound (128 / 168 VGPRs) on purpose; a single spilled register gives a launch a
    scratch allocation and costs it its dispatch rate (measured: 60 us instead of 20).  Checked where it can be checked without a GPU:
    the kernel descriptors of the built objects (private_segment_fixed_size of every matmul_mfma_kernel / attn_tile_kernel / lnq_kernel)."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(llvm + "/clang-offload-bundler") and shutil.which("objcopy")):
        pytest.skip("no clang-offload-bundler / objcopy in this image")
    pkg.build()
    seen = 0
    # (the float-weight persistent launch, kernels_fpipe.hip.h: a scratch reload in a computing wave stands in its vector-memory queue behind a layer of weight requests)
    for obj, pat in (("mfma_tu.o", r"matmul_mfma_kernel"), ("engine.o", r"attn_tile_kernel|lnq_kernel"), ("fpipe_tu.o", r"fpipe_kernel")):
        path = os.path.join(ROOT, "biogpt.cpp_amd", "csrc", "obj", obj)
        assert os.path.exists(path), path
        fat, co = str(tmp_path / (obj + ".fatbin")), str(tmp_path / (obj + ".co"))
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([llvm + "/llvm-readelf", "--notes", co], text=True)
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name and re.search(pat, name):
                assert int(m.group(1)) == 0, "%s uses %s bytes of scratch per lane" % (name, m.group(1))
                seen += 1
    assert seen >= 32
