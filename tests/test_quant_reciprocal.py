"""Exact division in the quantisers.  (1) the trellis kernels' integer reciprocal (types.h DevQuant::mul / ::sh, k_trellis.hip): (a << sh) * mul >> 32
on the 24-bit multiplier must be a / d for every dividend the kernels can meet (a = |coefficient| + d / 2 < 2^17) and every divisor d = 8 q of an
8-bit table.  (2) the pixel kernels' scalar quantiser (k_pixel.hip quant_one): one f32 fused multiply-add, exhaustively (tests/quant_fma_check.c)."""
import os
import subprocess
import tempfile

import numpy as np


def test_fma_quantiser_every_16_bit_table_value_and_every_coefficient():
    src = os.path.join(os.path.dirname(__file__), "quant_fma_check.c")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "qcheck")
        subprocess.check_call(["gcc", "-O2", "-o", exe, src, "-lm"])
        out = subprocess.check_output([exe], timeout=600).decode()
    assert out.strip().endswith("bad=0"), out


def recip(d):
    lg = int(d).bit_length() - 1
    P = max(25, lg + 18)
    return (1 << P) // d + 1, 32 - P


def test_every_dividend_and_every_8_bit_divisor():
    a = np.arange(1 << 17, dtype=np.uint64)
    for q in range(1, 256):
        d = 8 * q
        mul, sh = recip(d)
        assert mul < (1 << 24) and 0 <= sh <= 7
        x = a << np.uint64(sh)
        assert int(x.max()) < (1 << 24)
        got = (x * np.uint64(mul)) >> np.uint64(32)
        assert np.array_equal(got, a // np.uint64(d)), q


def test_larger_divisors_up_to_the_limit():
    a = np.arange(1 << 17, dtype=np.uint64)
    for d in (2041, 4095, 4096, 8191, 8192, 12345, (1 << 14) - 1):
        mul, sh = recip(d)
        assert mul < (1 << 24) and 0 <= sh <= 7
        assert np.array_equal(((a << np.uint64(sh)) * np.uint64(mul)) >> np.uint64(32), a // np.uint64(d)), d
