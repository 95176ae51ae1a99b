"""PNG in, JPEG out (convert_in_memory to JPEG): the PNG decode stages hand 8-bit grey / RGB pixels, device to device, to the JPEG path's
resize and encoder.  Kernel sources compiled for the CPU, against the oracle's statement; the same cases run on the device in
test_zzzz_png_jpeg_gpu.py."""
import io

import numpy as np
import pytest

from _util import emul_api, oracle_png_to_jpeg, package, png_cases

PIL = pytest.importorskip("PIL.Image")
JPEG = 0


@pytest.fixture(scope="module")
def api():
    return emul_api()


def check(api, cases, quality=80, width=0, height=0, subsampling=0, baseline=False):
    from oracle import oracle as O
    p = package().default_parameters(jpeg_quality=quality, width=width, height=height, jpeg_chroma_subsampling=subsampling, jpeg_progressive=not baseline)
    outs = api.batch_convert([c[1] for c in cases], p, JPEG)
    done = 0
    for (name, src), out in zip(cases, outs):
        try:
            want = oracle_png_to_jpeg(src, quality, width, height, subsampling or 420, 0 if baseline else 1)
        except O.PngError as e:
            assert isinstance(out, Exception) and out.code == e.code, (name, out, e.code)
            continue
        assert not isinstance(out, Exception), (name, out)
        assert out == want, name
        im = PIL.open(io.BytesIO(out))
        im.load()
        assert im.format == "JPEG"
        done += 1
    return done


def test_every_png_format_converts_like_the_oracle(api):
    cases = png_cases()
    assert check(api, cases) == len(cases)
    assert check(api, cases[:8], quality=35) == 8


def test_encoder_parameters_apply(api):
    cases = [c for c in png_cases() if c[0] in ("RGB_97x61", "RGBA_97x61", "RGB_200x150_3chunks", "L_97x61")]
    assert check(api, cases, 90, subsampling=444) == 4
    assert check(api, cases, 70, subsampling=422) == 4
    assert check(api, cases, 80, baseline=True) == 4


def test_resize_in_front(api):
    cases = [c for c in png_cases() if c[0] in ("RGB_97x61", "P_97x61", "I;16_97x61", "RGB_200x150_3chunks", "adam7_RGB_33x21")]
    assert check(api, cases, 80, width=50) == 5
    assert check(api, cases[:3], 80, width=120, height=40) == 3


def test_pixels_are_the_png(api):
    """a semantic anchor outside the oracle: at q100 4:4:4 the JPEG decodes (Pillow) to within a few levels of the PNG's pixels"""
    src = dict(png_cases())["RGB_200x150_3chunks"]
    out = api.batch_convert([src], package().default_parameters(jpeg_quality=100, jpeg_chroma_subsampling=444), JPEG)[0]
    a = np.asarray(PIL.open(io.BytesIO(out)).convert("RGB")).astype(int)
    b = np.asarray(PIL.open(io.BytesIO(src)).convert("RGB")).astype(int)
    assert a.shape == b.shape and np.abs(a - b).mean() < 1.5


def test_mixed_batch_and_failures(api):
    from gen_synth import synth_jpeg
    from test_png_emul import damaged_pngs
    from oracle import oracle as O
    cases = dict(png_cases())
    jpg = synth_jpeg(3, 64, 48)
    blobs = [cases["RGB_97x61"], jpg, b"junk", cases["LA_97x61"]] + damaged_pngs(9, 16)
    p = package().default_parameters(jpeg_quality=75)
    outs = api.batch_convert(blobs, p, JPEG)
    assert outs[0] == oracle_png_to_jpeg(blobs[0], 75) and outs[3] == oracle_png_to_jpeg(blobs[3], 75)
    assert outs[1].code == 10407 and outs[2].code == 10200
    assert api.convert_in_memory(blobs[0], p, JPEG) == outs[0]
    for b, o in zip(blobs[4:], outs[4:]):
        try:
            want = oracle_png_to_jpeg(b, 75)
        except O.PngError:
            want = None
        if want is None:
            assert isinstance(o, Exception)
        else:
            assert o == want
