"""No stage of the PNG, WebP and conversion paths may depend on the order in which the workgroups of a launch run: the emulation build
runs every launch back to front here (the JPEG path has the same check in test_pipeline_emul.py)."""
import ctypes

import pytest

from _util import emul_api, png_cases


@pytest.fixture()
def api():
    a = emul_api()
    a.L.csh_emul_set_reverse.argtypes = [ctypes.c_int]
    a.L.csh_emul_set_reverse(1)
    yield a
    a.L.csh_emul_set_reverse(0)


def test_png_paths_back_to_front(api):
    import test_png_emul as T
    import test_png_lossy_emul as PL
    import test_png_resize_emul as PR
    cases = png_cases()
    T.check_batch(api, cases[:10] + [c for c in cases if c[0].startswith(("palette_", "adam7_P", "adam7_RGB_33"))], 3)
    PL.check_lossy(api, PL.lossy_cases()[:3])
    assert PR.check(api, cases[:8], True, level=1, width=40) >= 6


def test_webp_and_conversions_back_to_front(api):
    import test_jpeg_png_emul as JP
    import test_png_webp_emul as PW
    import test_webp_emul as W
    W.check(api, W.webp_cases(), 85)
    assert PW.check(api, png_cases()[:8], 70) >= 4
    JP.check(api, W.webp_cases()[:3], True, level=2)
    JP.check(api, W.webp_cases()[:3], False)


def test_webp_inputs_back_to_front(api):
    """the decoders' wave fronts (reconstruction, loop filter, VP8L predictor, alpha unfilter) and the per-lane output stages"""
    import test_webp_decode_emul as WD
    WD.test_emul_synthetic_files_decode_like_libwebp(api)
    WD.test_emul_lossless_files_decode_like_libwebp(api)
    WD.test_emul_transparent_files_decode_like_libwebp(api)

