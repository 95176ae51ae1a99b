"""JPEG in, WebP out (convert_in_memory to WebP): the kernel sources compiled for the CPU against the oracle (= libwebp's bytes: tests/test_oracle_webp.py); the same cases
run on the device in test_webp_gpu.py."""
import io

import numpy as np
import pytest

from _util import emul_api, oracle_jpeg_to_webp, package, png_cases
from gen_synth import synth_jpeg

PIL = pytest.importorskip("PIL.Image")
WEBP, JPEG, PNG, TIFF = 3, 0, 1, 4


@pytest.fixture(scope="module")
def api():
    return emul_api()


def webp_cases(big=False):
    cases = [("420_160x96", synth_jpeg(1, 160, 96, texture=10)), ("444_97x61", synth_jpeg(2, 97, 61, subsampling=0, texture=5)), ("flat_64x48", synth_jpeg(3, 64, 48)),
             ("422_50x34", synth_jpeg(4, 50, 34, subsampling=1, texture=8)), ("prog_104x72", synth_jpeg(5, 104, 72, progressive=True, texture=6)), ("tiny_9x5", synth_jpeg(6, 9, 5, texture=3))]
    import io as _io
    g = PIL.open(_io.BytesIO(synth_jpeg(7, 80, 60, texture=6))).convert("L")
    b = _io.BytesIO(); g.save(b, "JPEG", quality=90)
    cases.append(("grey_80x60", b.getvalue()))
    if big:
        cases.append(("420_1920x1080", synth_jpeg(8, 1920, 1080)))
    return cases


def check(api, cases, quality, width=0, height=0):
    pkg = package()
    p = pkg.default_parameters(webp_quality=quality, jpeg_quality=quality, width=width, height=height)
    outs = api.batch_convert([c[1] for c in cases], p, WEBP)
    for (name, src), out in zip(cases, outs):
        assert not isinstance(out, Exception), (name, out)
        assert out == oracle_jpeg_to_webp(src, quality, width, height), name
        im = PIL.open(io.BytesIO(out))
        im.load()
        assert im.format == "WEBP"


def test_convert_equals_oracle(api):
    check(api, webp_cases(), 85)
    check(api, webp_cases()[:3], 30)


def test_every_quality_class_and_shape(api):
    """the coder back end takes its decisions from (bit, probability) pairs written down ahead of it (k_webp_decisions / k_webp_hdr / k_webp_bool: a 64-bit
    accumulator flushed every fourth decision, carries added before the byte is written); the encoder in front of it is libwebp's (k_vp8enc.hip).  Busy and
    empty pictures, both macroblock kinds, one-macroblock-wide and -high ones, values of every size class (q 100 gives long extra-bit runs and switches the
    chroma error diffusion off), several statistics chunks (200 x 150 = 130 macroblocks: the cost tables are rebuilt after 96)"""
    cases = webp_cases() + [("busy_200x150", synth_jpeg(9, 200, 150, texture=90)), ("tall_24x200", synth_jpeg(10, 24, 200, texture=40)), ("wide_333x17", synth_jpeg(11, 333, 17, texture=20))]
    for q in (100, 99, 75, 5):
        assert check(api, cases, q) is None


def test_statistics_books_overflow_in_order(api):
    """libwebp keeps its token statistics in 16 + 16 bits and halves a slot when 65534 events are reached: WHICH events came before matters.  A noisy picture
    makes several slots pass that point (and k_vp8_loop's chunk_stats recount them in macroblock order); the file must still be the oracle's = libwebp's"""
    check(api, [("noisy_640x400", synth_jpeg(12, 640, 400, texture=120))], 95)


def test_boolean_coder_in_pieces(api, monkeypatch):
    """a partition is coded in pieces of 2048 decisions from the ranges a scan over all 128 possible ones finds (k_bool_scan / _chain / _code / _merge); with pieces of
    1, 3 and 61 decisions most pieces complete fewer than two bytes, some none: what they hold travels through them into later pieces, carries ripple back"""
    cases = webp_cases()[:4] + [("busy_200x150", synth_jpeg(9, 200, 150, texture=90))]
    for seg in ("1", "3", "61"):
        monkeypatch.setenv("CSH_TEST_BOOL_SEG", seg)
        check(api, cases, 85)
        check(api, cases[:2], 100)
    monkeypatch.delenv("CSH_TEST_BOOL_SEG")


def test_convert_with_resize(api):
    check(api, webp_cases()[:2], 85, width=60)
    check(api, webp_cases()[1:3], 75, height=40)


def test_entry_point_and_refusals(api):
    pkg = package()
    src = webp_cases()[0][1]
    p = pkg.default_parameters(webp_quality=85)
    assert api.convert_in_memory(src, p, WEBP) == oracle_jpeg_to_webp(src, 85)
    png = dict(png_cases())["RGBA_97x61"]   # transparency: an extended file with an ALPH chunk (test_png_webp_emul.py)
    outs = api.batch_convert([src, png, b"junk", src], p, WEBP)
    assert outs[0] == oracle_jpeg_to_webp(src, 85) and outs[3] == outs[0]
    assert [getattr(o, "code", 0) for o in outs] == [0, 0, 10200, 0] and outs[1][12:16] == b"VP8X"
    with pytest.raises(Exception) as e:
        api.convert_in_memory(src, p, JPEG)
    assert e.value.code == 10407
    with pytest.raises(Exception) as e:
        api.convert_in_memory(src, p, TIFF)   # JPEG -> PNG is built: test_jpeg_png_emul.py
    assert e.value.code == 10201
    out = api.convert_in_memory(src, pkg.default_parameters(webp_quality=85, webp_lossless=True), WEBP)   # lossless WebP from a JPEG: tests/test_webp_lossless_emul.py
    assert out[8:16] == b"WEBPVP8L"
    out = api.convert_in_memory(dict(png_cases())["RGB_97x61"], pkg.default_parameters(webp_lossless=True), WEBP)   # and from an opaque PNG
    assert out[8:16] == b"WEBPVP8L"
    out = api.convert_in_memory(dict(png_cases())["RGBA_97x61"], pkg.default_parameters(webp_lossless=True), WEBP)   # alpha stays (tests/test_webp_lossless_emul.py)
    assert out[8:16] == b"WEBPVP8L"


def test_damaged_jpegs_convert_like_the_oracle_or_fail(api):
    """the JPEG side of a conversion sees the same damaged inputs as the JPEG path: whatever decodes must give the oracle's WebP"""
    from test_pipeline_emul import fuzzed_blobs
    pkg = package()
    blobs = fuzzed_blobs(11, 24, True)
    outs = api.batch_convert(blobs, pkg.default_parameters(webp_quality=75), WEBP)
    decoded = 0
    for b, o in zip(blobs, outs):
        try:
            want = oracle_jpeg_to_webp(b, 75)
        except Exception:
            want = None
        if want is None:
            assert isinstance(o, Exception)
        else:
            assert o == want
            decoded += 1
    assert decoded >= 4
