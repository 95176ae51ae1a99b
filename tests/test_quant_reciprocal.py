"""The quantiser's exact division (types.h DevQuant::mul / ::sh, k_pixel.hip quant_one): (a << sh) * mul >> 32 on the 24-bit multiplier must be
a / d for every dividend the kernels can meet (a = |coefficient| + d / 2 < 2^17) and every divisor d = 8 q of an 8-bit table."""
import numpy as np


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
