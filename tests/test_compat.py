"""The C++ source-level boundary (include/biogpt_compat.h): reference-shaped programs compile against it
and run on the engine."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, has_gpu

REF_MAIN = "/root/reference/examples/main/main.cpp"
INC = os.path.join(ROOT, "include", "compat")


def _build_driver(pkg, tmp_path):
    exe = str(tmp_path / "compat_driver")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", INC, os.path.join(ROOT, "tests", "compat_driver.cpp"),
                           "-L", libdir, "-lbiogpt_hip", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference sources only exist in the build container")
def test_reference_main_compiles_unmodified_against_compat_headers():
    # syntax + semantic check of the reference's own CLI source against OUR headers (no reference header is used)
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-I", INC, REF_MAIN])


def test_driver_links_and_fails_loudly_without_model(pkg, tmp_path):
    exe = _build_driver(pkg, tmp_path)
    r = subprocess.run([exe, str(tmp_path / "missing.bin"), "4", "1", "2", "5"], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to open" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["q4_0", "q5_1"])
def test_driver_greedy_ids_match_oracle(pkg, oracle, tiny_models, tmp_path, name):
    exe = _build_driver(pkg, tmp_path)
    prompt = [2, 17, 45, 300, 9, 128, 64, 255, 31, 7, 199]        # 11 ids -> chunks of 8 + 3 (n_batch = 8)
    r = subprocess.run([exe, tiny_models[name], "24", "1"] + [str(t) for t in prompt], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ids = [int(t) for t in r.stdout.split()]
    ref, _ = oracle.OracleModel(tiny_models[name], n_threads=2).generate_greedy(prompt, 24, n_batch=8)
    assert ids == list(ref)
    assert "vocab 320 tokens, 7 merges, n_loaded 37" in r.stderr


def test_params_parse_flags(pkg, tmp_path):
    exe = _build_driver(pkg, tmp_path)
    r = subprocess.run([exe, "--flags", "--bogus", "1"], capture_output=True, text=True)
    assert r.returncode == 0 and "unknown argument: --bogus" in r.stderr and "--top_k N" in r.stderr   # biogpt.cpp:1011-1015
    r = subprocess.run([exe, "--flags", "-m"], capture_output=True, text=True)
    assert r.returncode == 2 and "missing value" in r.stderr
    r = subprocess.run([exe, "--flags", "-m", str(tmp_path / "nope.bin"), "-n", "3", "--top_k", "1", "-p", "2 5"], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to open" in r.stderr


@pytest.mark.gpu
def test_flags_drive_generation(pkg, oracle, tiny_models, tmp_path):
    exe = _build_driver(pkg, tmp_path)
    r = subprocess.run([exe, "--flags", "-m", tiny_models["q4_1"], "-n", "12", "--top_k", "1", "-b", "4", "-p", "2 17 45 300 9 128"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref, _ = oracle.OracleModel(tiny_models["q4_1"]).generate_greedy([2, 17, 45, 300, 9, 128], 12, n_batch=4)
    assert [int(t) for t in r.stdout.split()] == list(ref)


@pytest.mark.gpu
def test_sampler_top_k_top_p_is_seeded_and_bounded(pkg, tiny_models, tmp_path):
    exe = _build_driver(pkg, tmp_path)
    a = subprocess.run([exe, tiny_models["q8_0"], "16", "40", "2", "17"], capture_output=True, text=True)
    b = subprocess.run([exe, tiny_models["q8_0"], "16", "40", "2", "17"], capture_output=True, text=True)
    assert a.returncode == 0 and a.stdout == b.stdout          # mt19937(7): reproducible
    ids = [int(t) for t in a.stdout.split()]
    assert len(ids) == 16 and all(0 <= t < 320 for t in ids)
