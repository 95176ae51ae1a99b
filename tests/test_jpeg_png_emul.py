"""JPEG in, PNG out (convert_in_memory to PNG): the JPEG decode (and resize) stages leave the pixels in device memory, the PNG coder takes
them from there.  Kernel sources compiled for the CPU, against the oracle's statement (decode, then the PNG path over a file of those
pixels); the same cases run on the device in test_zzz_jpeg_png_gpu.py."""
import io

import pytest

from _util import emul_api, oracle_jpeg_to_png, package
from test_webp_emul import webp_cases

PIL = pytest.importorskip("PIL.Image")
PNG = 1


@pytest.fixture(scope="module")
def api():
    return emul_api()


def check(api, cases, lossless, level=3, width=0, height=0):
    p = package().default_parameters(png_optimize=lossless, png_optimization_level=level, width=width, height=height)
    outs = api.batch_convert([c[1] for c in cases], p, PNG)
    for (name, src), out in zip(cases, outs):
        assert not isinstance(out, Exception), (name, out)
        assert out == oracle_jpeg_to_png(src, lossless, level, width, height), name
        im = PIL.open(io.BytesIO(out))
        im.load()
        assert im.format == "PNG"


def test_lossless_target_equals_oracle(api):
    check(api, webp_cases(), True)
    check(api, webp_cases()[:3], True, level=1)


def test_quantising_target_equals_oracle(api):
    check(api, webp_cases(), False)
    # -q reaches the quantiser of the target
    cases = webp_cases()[:2]
    p = package().default_parameters(png_optimize=False, png_optimization_level=1, png_quality=25)
    outs = api.batch_convert([c[1] for c in cases], p, PNG)
    assert outs == [oracle_jpeg_to_png(c[1], False, 1, quality=25) for c in cases]
    assert all(len(o) < len(oracle_jpeg_to_png(c[1], False, 1)) for o, c in zip(outs, cases))


def test_pixels_survive_a_lossless_target(api):
    """the PNG holds exactly the RGB the JPEG decodes to (Pillow's libjpeg agrees with the oracle's decode on 4:4:4 input)"""
    import numpy as np
    name, src = webp_cases()[1]
    out = api.batch_convert([src], package().default_parameters(png_optimize=True), PNG)[0]
    a = np.asarray(PIL.open(io.BytesIO(out)).convert("RGB")).astype(int)
    b = np.asarray(PIL.open(io.BytesIO(src)).convert("RGB")).astype(int)
    assert a.shape == b.shape and np.abs(a - b).max() <= 1


def test_resize_in_front(api):
    check(api, webp_cases()[:2], True, width=60)
    check(api, webp_cases()[1:3], False, height=40)


def test_mixed_batch_and_failures(api):
    from oracle import oracle as O
    from _util import oracle_jpeg_to_webp, png_cases
    cases = webp_cases()
    p = package().default_parameters(png_optimize=True, png_optimization_level=2)
    png = png_cases()[0][1]
    blobs = [cases[0][1], b"junk", cases[1][1][:200], png, cases[2][1]]
    outs = api.batch_convert(blobs, p, PNG)
    assert outs[0] == oracle_jpeg_to_png(blobs[0], True, 2) and outs[4] == oracle_jpeg_to_png(blobs[4], True, 2)
    assert outs[1].code == 10200 and isinstance(outs[2], Exception) and outs[3].code == 10407
    assert api.convert_in_memory(blobs[0], p, PNG) == outs[0]


def test_damaged_jpegs_convert_like_the_oracle_or_fail(api):
    from test_pipeline_emul import fuzzed_blobs
    blobs = fuzzed_blobs(13, 20, True)
    outs = api.batch_convert(blobs, package().default_parameters(png_optimize=True, png_optimization_level=1), PNG)
    decoded = 0
    for b, o in zip(blobs, outs):
        try:
            want = oracle_jpeg_to_png(b, True, 1)
        except Exception:
            want = None
        if want is None:
            assert isinstance(o, Exception)
        else:
            assert o == want
            decoded += 1
    assert decoded >= 3
