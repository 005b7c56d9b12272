import os
import sys

import numpy as np
import pytest

try:  # torch ships its own HIP runtime: load it BEFORE libbiogpt_hip.so pulls in /opt/rocm's copy, otherwise a test
    import torch  # noqa: F401  (that touches torch.cuda late in the process finds "No HIP GPUs")
except ImportError:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure, oracle/)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def pkg():
    """The product package (biogpt.cpp_amd), C-ABI library built if necessary."""
    import _pkg
    m = _pkg.load()
    m.build()
    return m


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def tiny_models(tmp_path_factory, oracle):
    """tiny_f32/f16 fixtures (written by the reference's convert.py) + all five quantizations of the
    f32 file produced by the ORACLE quantizer.  Returns {ftype_name: path}."""
    d = tmp_path_factory.mktemp("tiny")
    out = {"f32": os.path.join(GOLDEN, "tiny_f32.bin"), "f16": os.path.join(GOLDEN, "tiny_f16.bin")}
    for name, ft in (("q4_0", 2), ("q4_1", 3), ("q5_0", 8), ("q5_1", 9), ("q8_0", 7)):
        p = str(d / ("tiny_%s.bin" % name))
        oracle.quantize_file(out["f32"], p, ft)
        out[name] = p
    return out


def has_gpu():
    try:
        import _pkg
        m = _pkg.load()
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False
