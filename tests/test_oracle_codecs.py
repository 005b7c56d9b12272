"""Known-answer tests that pin the ORACLE's block codecs and scalar helpers.
Hand-computed blocks: values chosen so that the scale is exactly 1.0 and every code is an integer."""
import numpy as np
import pytest


def _u16(b, off=0):
    return int(b[off]) | (int(b[off + 1]) << 8)


def test_fp16_matches_numpy_all_halfs(oracle):
    halfs = np.arange(65536, dtype=np.uint16)
    ref = halfs.view(np.float16).astype(np.float32)
    got = np.array([oracle.fp16_to_fp32(int(h)) for h in halfs], dtype=np.float32)
    ok = (got == ref) | (np.isnan(got) & np.isnan(ref))
    assert ok.all()


def test_fp32_to_fp16_matches_numpy_rne(oracle):
    rng = np.random.default_rng(0)
    xs = np.concatenate([
        rng.standard_normal(4000).astype(np.float32) * 3,
        (rng.standard_normal(2000) * 1e-5).astype(np.float32),           # fp16 subnormal range
        (rng.standard_normal(1000) * 1e-8).astype(np.float32),           # underflow to zero
        (rng.standard_normal(1000) * 7e4).astype(np.float32),            # around overflow
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25,
                  2.0 ** -14, 1.0009765625, 1.00048828125, 1.00146484375, np.inf, -np.inf], dtype=np.float32)])
    with np.errstate(over="ignore"):
        ref = xs.astype(np.float16).view(np.uint16)
    got = np.array([oracle.fp32_to_fp16(float(x)) for x in xs], dtype=np.uint16)
    assert (got == ref).all()


def test_q4_0_known_block(oracle):
    x = np.concatenate([np.arange(16) - 8, 7 - np.arange(16)]).astype(np.float32)  # extreme value -8 -> d = 1
    b = oracle.quantize(oracle.TYPE_Q4_0, x, 32)
    assert len(b) == 18 and _u16(b) == 0x3C00
    assert list(b[2:]) == [j | ((15 - j) << 4) for j in range(16)]
    assert (oracle.dequantize_row(oracle.TYPE_Q4_0, b, 32) == x).all()


def test_q4_0_zero_block_and_positive_extreme(oracle):
    b = oracle.quantize(oracle.TYPE_Q4_0, np.zeros(32, np.float32), 32)
    assert _u16(b) == 0x8000 and list(b[2:]) == [0x88] * 16  # d = 0.0f / -8 = -0.0f
    # extreme value +8 -> d = -1: codes are 8 - x, and +8 itself clamps... to 0 (8 - 8)
    x = np.zeros(32, np.float32); x[3] = 8.0; x[4] = -7.0
    b = oracle.quantize(oracle.TYPE_Q4_0, x, 32)
    assert _u16(b) == 0xBC00  # -1.0
    assert (b[2 + 3] & 0xF) == 0 and (b[2 + 4] & 0xF) == 15 and (b[2] & 0xF) == 8
    assert (oracle.dequantize_row(oracle.TYPE_Q4_0, b, 32) == x).all()


def test_q4_1_known_block(oracle):
    x = np.concatenate([np.arange(16), 15 - np.arange(16)]).astype(np.float32) + 2.0  # min 2, max 17 -> d = 1
    b = oracle.quantize(oracle.TYPE_Q4_1, x, 32)
    assert len(b) == 20 and _u16(b, 0) == 0x3C00 and _u16(b, 2) == 0x4000
    assert list(b[4:]) == [j | ((15 - j) << 4) for j in range(16)]
    assert (oracle.dequantize_row(oracle.TYPE_Q4_1, b, 32) == x).all()


def test_q5_0_known_block(oracle):
    x = (np.arange(32) - 16).astype(np.float32)  # extreme -16 -> d = 1, codes 0..31
    b = oracle.quantize(oracle.TYPE_Q5_0, x, 32)
    assert len(b) == 22 and _u16(b) == 0x3C00
    qh = int.from_bytes(bytes(b[2:6]), "little")
    codes = np.arange(32)
    assert qh == sum(((int(c) >> 4) & 1) << j for j, c in enumerate(codes))
    assert list(b[6:]) == [(int(codes[j]) & 0xF) | ((int(codes[j + 16]) & 0xF) << 4) for j in range(16)]
    assert (oracle.dequantize_row(oracle.TYPE_Q5_0, b, 32) == x).all()


def test_q5_1_known_block(oracle):
    x = np.arange(32).astype(np.float32)[::-1].copy() - 3.0  # min -3, max 28 -> d = 1
    b = oracle.quantize(oracle.TYPE_Q5_1, x, 32)
    assert len(b) == 24 and _u16(b, 0) == 0x3C00 and _u16(b, 2) == 0xC200
    codes = (x + 3).astype(int)
    qh = int.from_bytes(bytes(b[4:8]), "little")
    assert qh == sum(((int(c) >> 4) & 1) << j for j, c in enumerate(codes))
    assert (oracle.dequantize_row(oracle.TYPE_Q5_1, b, 32) == x).all()


def test_q8_0_known_block_and_tie_rounding(oracle):
    x = np.zeros(32, np.float32)
    x[0] = 127.0  # d = 1
    x[1:5] = [0.5, -0.5, 2.5, -2.5]  # roundf: half AWAY from zero (nearest-even would give 0,0,2,-2)
    x[5:8] = [100.0, -126.0, 1.49]
    b = oracle.quantize(oracle.TYPE_Q8_0, x, 32)
    assert len(b) == 34 and _u16(b) == 0x3C00
    q = np.frombuffer(bytes(b[2:]), dtype=np.int8)
    assert list(q[:8]) == [127, 1, -1, 3, -3, 100, -126, 1]


def test_q8_1_fields(oracle):
    x = np.zeros(32, np.float32); x[0] = 254.0; x[1] = -3.0; x[17] = 20.0  # d = 2
    b = oracle.quantize(oracle.TYPE_Q8_1, x, 32)
    assert len(b) == 40
    d, s = np.frombuffer(bytes(b[:8]), dtype=np.float32)
    q = np.frombuffer(bytes(b[8:]), dtype=np.int8)
    assert d == 2.0 and list(q[:2]) == [127, -2] and q[17] == 10  # -1.5 rounds away from zero
    assert s == (127 - 2 + 10) * 2.0


@pytest.mark.parametrize("tname", ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0"])
def test_vec_dot_is_integer_block_dot(oracle, tname):
    """W*A8 semantics: dot = sum_blocks sumi * d_w * d_x (+ m_w * s_x), NOT dequant-to-f32."""
    t = getattr(oracle, "TYPE_" + tname)
    rng = np.random.default_rng(5)
    k = 256
    w = rng.standard_normal(k).astype(np.float32)
    x = rng.standard_normal(k).astype(np.float32) * 2
    wq = oracle.quantize(t, w, k)
    got = oracle.vec_dot(t, wq, x)
    asym = tname.endswith("_1")
    xa = oracle.quantize(oracle.TYPE_Q8_1 if asym else oracle.TYPE_Q8_0, x, k)
    wdq = oracle.dequantize_row(t, wq, k).astype(np.float64)
    bb = 40 if asym else 34
    ref = 0.0
    for b in range(k // 32):
        blk = bytes(xa[b * bb:(b + 1) * bb])
        if asym:
            d = float(np.frombuffer(blk[:4], np.float32)[0]); q = np.frombuffer(blk[8:], np.int8)
        else:
            d = float(np.frombuffer(blk[:2], np.float16)[0]); q = np.frombuffer(blk[2:], np.int8)
        ref += float(np.dot(wdq[b * 32:(b + 1) * 32], q.astype(np.float64) * d))
    assert abs(got - ref) < 1e-4 * max(1.0, abs(ref))
    # and it is measurably different from the f32-activation product for a typical row
    f32 = float(np.dot(wdq, x.astype(np.float64)))
    assert abs(got - f32) > 1e-6


def test_tables(oracle):
    xs = np.array([-4.0, -1.0, -0.5, 0.0, 0.3, 1.0, 2.5, 6.0], dtype=np.float32)
    for x in xs:
        xh = np.float32(np.float16(x))
        g = 0.5 * xh * (1.0 + np.tanh(np.float32(0.7978845608) * xh * (1.0 + np.float32(0.044715) * xh * xh)))
        assert abs(oracle.lib().bo_gelu_table(float(x)) - float(np.float16(g))) <= abs(float(np.float16(g))) * 2e-3 + 1e-7
        e = np.exp(-abs(xh))
        assert abs(oracle.lib().bo_exp_table(-abs(float(x))) - float(np.float16(e))) <= float(e) * 2e-3
