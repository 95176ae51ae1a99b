"""WebP inputs on the MI355X: the cases of tests/test_webp_decode_emul.py through the product library (decoder pinned to libwebp via Pillow)."""
import pytest

import test_webp_decode_emul as E
from _util import product_api

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: libcaesium_hip has no CPU path"
    return a


def test_reference_samples_decode_like_libwebp(api, reference_samples):
    E.test_emul_reference_samples_decode_like_libwebp(api, reference_samples)


def test_synthetic_files_decode_like_libwebp(api):
    E.test_emul_synthetic_files_decode_like_libwebp(api)


def test_lossless_files_decode_like_libwebp(api):
    E.test_emul_lossless_files_decode_like_libwebp(api)


def test_damaged_lossless_streams_fail_alone(api):
    E.test_emul_damaged_lossless_streams_fail_alone(api)


def test_damaged_and_unsupported_inputs_fail_alone(api):
    E.test_emul_damaged_and_unsupported_inputs_fail_alone(api)


def test_compress_and_convert_from_webp(api, reference_samples):
    E.test_emul_compress_and_convert_from_webp(api, reference_samples)


def test_compress_to_size_on_webp(api):
    E.test_emul_compress_to_size_on_webp(api)


def test_webp_metadata_carried_over(api, reference_samples):
    E.test_emul_webp_metadata_carried_over(api, reference_samples)


def test_transparent_files_decode_like_libwebp(api):
    E.test_emul_transparent_files_decode_like_libwebp(api)


def test_transparent_sources_keep_their_alpha(api):
    E.test_emul_transparent_sources_keep_their_alpha(api)


def test_damaged_transparent_files_fail_alone(api):
    E.test_emul_damaged_transparent_files_fail_alone(api)
