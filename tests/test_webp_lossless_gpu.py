"""Lossless WebP output on the MI355X: the cases of tests/test_webp_lossless_emul.py through the product library."""
import pytest

import test_webp_lossless_emul as E
from _util import product_api

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: libcaesium_hip has no CPU path"
    return a


def test_lossless_webp_round_trips_through_libwebp(api, reference_samples):
    E.test_emul_lossless_webp_round_trips_through_libwebp(api, reference_samples)


def test_lossless_webp_sizes_are_sane(api):
    E.test_emul_lossless_webp_sizes_are_sane(api)


def test_jpeg_to_lossless_webp_and_resize(api):
    E.test_emul_jpeg_to_lossless_webp_and_resize(api)


def test_lossless_webp_failures_stay_per_file(api):
    E.test_emul_lossless_webp_failures_stay_per_file(api)


def test_device_writes_the_oracles_bytes(api):
    """every case above already compares the device's file with the oracle's statement of the coder (check_vp8l -> oracle_vp8l: oracle/png_oracle.c
    cso_vp8l_encode); here one larger picture, against the oracle and against the emulation build"""
    import test_webp_decode_emul as D
    from gen_synth import synth_rgb
    from _util import emul_api
    src = D.lossless_of(synth_rgb(9, 150, 90, texture=20.0))
    p = E.params(webp_lossless=True)
    out = api.compress_in_memory(src, p)
    assert out == E.oracle_vp8l(out)
    assert out == emul_api().compress_in_memory(src, p)


def test_png_to_lossless_webp(api):
    E.test_emul_png_to_lossless_webp(api)
