"""JPEG in, WebP out on the device, through the C ABI, against the oracle (oracle/vp8enc_oracle.c = libwebp's encoder, pinned to WebPEncode byte for byte, + the
JPEG / resize oracle in front of it): file bytes; configs[3] shape (1080p JPEG, --long-edge 1500, -q 85) included; and against libwebp itself where the box has one."""
import io

import numpy as np
import pytest

from _util import oracle_jpeg_to_webp, package, product_api
from gen_synth import synth_jpeg
from test_webp_emul import check, webp_cases

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: libcaesium_hip has no CPU path"
    return a


def test_convert_equals_oracle(api):
    check(api, webp_cases(), 85)
    check(api, webp_cases(), 30)
    check(api, webp_cases()[:4], 100)


def test_every_quality_class_and_shape(api):
    import test_webp_emul as E
    E.test_every_quality_class_and_shape(api)


def test_statistics_books_overflow_in_order(api):
    import test_webp_emul as E
    E.test_statistics_books_overflow_in_order(api)
    check(api, [("noisy_1280x720", synth_jpeg(13, 1280, 720, texture=100))], 90)


def test_boolean_coder_in_pieces(api, monkeypatch):
    import test_webp_emul as E
    E.test_boolean_coder_in_pieces(api, monkeypatch)


def test_device_file_is_libwebp_s_file(api):
    """the whole claim in one place: RGB pixels -> this library's PNG -> WebP conversion on the device == WebPEncode of the libwebp on this box on the same pixels"""
    from libwebp_pin import libwebp_encode, libwebps
    from gen_synth import synth_rgb
    libs = libwebps()
    if not libs:
        pytest.skip("no libwebp with the encoder API on this box")
    pkg = package()
    for seed, (w, h), q in ((0, (1500, 844), 85), (1, (640, 481), 60), (2, (97, 61), 100)):
        rgb = np.ascontiguousarray(synth_rgb(seed, w, h))
        b = io.BytesIO()
        PIL.fromarray(rgb).save(b, "PNG")
        out = api.convert_in_memory(b.getvalue(), pkg.default_parameters(webp_quality=q), 3)
        for ver, W in libs:
            assert out == libwebp_encode(W, rgb, q), (seed, w, h, q, ver)


def test_convert_with_resize(api):
    check(api, webp_cases()[:3], 85, width=60)
    check(api, webp_cases()[1:4], 75, height=40)


def test_config4_shape(api):
    """1920x1080 JPEG -> --long-edge 1500 (1500x844) -> WebP q85"""
    cases = webp_cases(big=True)[-1:]
    pkg = package()
    out = api.batch_convert([cases[0][1]] * 3, pkg.default_parameters(webp_quality=85, width=1500), 3)
    want = oracle_jpeg_to_webp(cases[0][1], 85, 1500, 0)
    assert all(o == want for o in out)
    im = PIL.open(io.BytesIO(out[0]))
    assert im.size == (1500, 844)


def test_refusals(api):
    from test_webp_emul import test_entry_point_and_refusals
    test_entry_point_and_refusals(api)
