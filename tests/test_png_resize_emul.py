"""PNG sources with --width / --height: decode, Lanczos3 over the decoded samples, the PNG coder over the result.  Kernel sources compiled
for the CPU, against the oracle's statement; the same cases run on the device in test_zzz_png_resize_gpu.py."""
import io

import numpy as np
import pytest

from _util import emul_api, oracle_png_resized, package, png_cases

PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def api():
    return emul_api()


def check(api, cases, lossless, level=3, width=0, height=0):
    from oracle import oracle as O
    p = package().default_parameters(png_optimize=lossless, png_optimization_level=level, width=width, height=height)
    outs = api.cs_batch_compress([c[1] for c in cases], p)
    done = 0
    for (name, src), out in zip(cases, outs):
        try:
            want = oracle_png_resized(src, lossless, level, width, height)
        except O.PngError as e:
            assert isinstance(out, Exception) and out.code == e.code, (name, out, e.code)
            continue
        assert not isinstance(out, Exception), (name, out)
        assert out == want, name
        im = PIL.open(io.BytesIO(out))
        im.load()
        done += 1
    return done


def test_every_case_resizes_like_the_oracle_or_is_refused(api):
    cases = png_cases()
    assert check(api, cases, True, level=1, height=30) == len(cases)   # every format; L_1x300 becomes 1x30, RGBA_300x2 4500x30
    assert check(api, cases[:6], False, width=25) >= 4


def test_expansion_before_the_resize(api):
    """palette (with and without tRNS), 1-bit, grey and RGB with a transparent colour: resized as RGB(A) / grey(+alpha) 8-bit images"""
    cases = dict(png_cases())
    names = ["P_97x61", "1_97x61", "reduce_blocked_by_trns", "adam7_P_40x17", "adam7_1_37x11"]
    pick = [(k, cases[k]) for k in names]

    def resave(name, **kw):
        b = io.BytesIO()
        PIL.open(io.BytesIO(cases[name])).save(b, "PNG", **kw)
        return b.getvalue()
    pick.append(("P_with_trns", resave("P_97x61", transparency=bytes(range(0, 250, 10)))))   # 25 palette entries with their own alpha
    pick.append(("L_with_trns", resave("L_97x61", transparency=120)))
    pick.append(("1bit_with_trns", resave("1_97x61", transparency=1)))
    assert b"tRNS" in pick[-1][1] and b"tRNS" in pick[-2][1] and b"tRNS" in pick[-3][1]
    assert check(api, pick, True, level=2, width=31) == len(pick)
    out = api.cs_batch_compress([pick[-2][1]], package().default_parameters(png_optimize=True, width=31))[0]
    assert PIL.open(io.BytesIO(out)).mode in ("RGBA", "LA", "P")   # grey 120 was transparent: the result carries alpha


def test_sixteen_bit_images_stay_sixteen_bit(api):
    """image-rs resamples L16 / Rgb16 at 16 bits; the coder may then narrow what fits 8 bits (reduce_i16_narrow does, noise does not)"""
    from test_png_webp_emul import make_png
    rng = np.random.default_rng(8)
    rgb16 = make_png(37, 29, 16, 2, rng.integers(0, 65536, (29, 37, 3)).astype(">u2").tobytes())
    la16 = make_png(21, 33, 16, 4, rng.integers(0, 65536, (33, 21, 2)).astype(">u2").tobytes())
    cases = dict(png_cases())
    pick = [("rgb16_noise", rgb16), ("la16_noise", la16), ("I;16_97x61", cases["I;16_97x61"]), ("reduce_i16_narrow", cases["reduce_i16_narrow"]), ("adam7_I;16_12x20", cases["adam7_I;16_12x20"])]
    assert check(api, pick, True, level=2, width=19) == 5
    assert check(api, pick, True, level=1, width=60, height=50) == 5
    out = api.cs_batch_compress([rgb16], package().default_parameters(png_optimize=True, width=19))[0]
    assert out[24] == 16   # IHDR bit depth


def test_sixteen_bit_images_with_a_colour_key(api):
    """16-bit grey / RGB with a tRNS chunk: the png crate's EXPAND hands image-rs La16 / Rgba16 -- the key becomes a 16-bit alpha sample (0 at the key, 65535
    elsewhere), the picture is resampled at 16 bits with it and written as grey + alpha / RGBA"""
    from test_png_webp_emul import make_png
    rng = np.random.default_rng(9)
    g = rng.integers(0, 4, (23, 31)).astype(">u2") * 21845            # four levels: the key hits a quarter of the pixels
    rgb = rng.integers(0, 2, (19, 27, 3)).astype(">u2") * 65535       # eight colours
    grey_key = make_png(31, 23, 16, 0, g.tobytes(), extra=[(b"tRNS", (21845).to_bytes(2, "big"))])
    rgb_key = make_png(27, 19, 16, 2, rgb.tobytes(), extra=[(b"tRNS", b"\xff\xff\x00\x00\xff\xff")])
    no_hit = make_png(31, 23, 16, 0, g.tobytes(), extra=[(b"tRNS", (7).to_bytes(2, "big"))])   # a key no pixel has: the alpha is opaque everywhere (and may be dropped again)
    pick = [("grey16_key", grey_key), ("rgb16_key", rgb_key), ("grey16_key_unused", no_hit)]
    assert check(api, pick, True, level=2, width=17) == 3
    assert check(api, pick, True, level=1, width=40, height=30) == 3
    out = api.cs_batch_compress([rgb_key], package().default_parameters(png_optimize=True, width=17))[0]
    im = PIL.open(io.BytesIO(out))
    assert im.size == (17, 12) and im.mode in ("RGBA", "RGBA;16B") and out[24] == 16 and out[25] == 6   # 16-bit RGBA in the file
    bad = make_png(8, 8, 16, 2, bytes(8 * 8 * 6), extra=[(b"tRNS", b"\0\1")])   # a key of the wrong length
    r = api.cs_batch_compress([bad], package().default_parameters(png_optimize=True, width=4))[0]
    assert isinstance(r, Exception)


def test_sizes_and_shapes(api):
    cases = dict(png_cases())
    pick = [(k, cases[k]) for k in ("RGB_97x61", "RGBA_97x61", "L_97x61", "LA_97x61", "RGB_200x150_3chunks")]
    assert check(api, pick, True, level=2, width=150) == 5            # enlarging
    assert check(api, pick, True, level=2, width=33, height=77) == 5  # both given: exact, aspect not kept
    assert check(api, pick[:2], True, level=2, width=97) == 2         # same size: a copy
    out = api.cs_batch_compress([pick[1][1]], package().default_parameters(png_optimize=True, width=50))[0]
    assert PIL.open(io.BytesIO(out)).size == (50, 31)


def test_result_is_close_to_pillows_lanczos(api):
    """a semantic anchor outside the oracle: Pillow's own Lanczos resize of the same image (different arithmetic, same filter)"""
    cases = dict(png_cases())
    src = cases["RGB_200x150_3chunks"]
    out = api.cs_batch_compress([src], package().default_parameters(png_optimize=True, png_optimization_level=1, width=100))[0]
    a = np.asarray(PIL.open(io.BytesIO(out)).convert("RGB")).astype(int)
    b = np.asarray(PIL.open(io.BytesIO(src)).convert("RGB").resize((100, 75), PIL.LANCZOS)).astype(int)
    assert a.shape == b.shape and np.abs(a - b).mean() < 1.0


def test_mixed_batch_with_jpegs_and_damage(api):
    from _util import oracle_resized
    from gen_synth import synth_jpeg
    from test_png_emul import damaged_pngs
    cases = dict(png_cases())
    jpg = synth_jpeg(3, 120, 90, texture=5)
    from test_png_webp_emul import make_png
    wide_trns = make_png(20, 10, 16, 0, bytes(range(200)) * 2, extra=[(b"tRNS", b"\0\7")])   # 16-bit grey with a transparent level: resized as La16
    blobs = [cases["RGB_97x61"], jpg, wide_trns, b"junk", cases["P_97x61"], cases["I;16_97x61"]] + damaged_pngs(5, 12)
    p = package().default_parameters(png_optimize=True, png_optimization_level=1, jpeg_quality=80, width=48)
    outs = api.cs_batch_compress(blobs, p)
    assert outs[0] == oracle_png_resized(blobs[0], True, 1, 48, 0)
    assert outs[1] == oracle_resized(jpg, 48, 0)
    assert outs[2] == oracle_png_resized(wide_trns, True, 1, 48, 0) and outs[3].code == 10200
    assert outs[4] == oracle_png_resized(blobs[4], True, 1, 48, 0) and outs[5] == oracle_png_resized(blobs[5], True, 1, 48, 0)
    from oracle import oracle as O
    for b, o in zip(blobs[6:], outs[6:]):
        try:
            want = oracle_png_resized(b, True, 1, 48, 0)
        except O.PngError:
            want = None
        if want is None:
            assert isinstance(o, Exception)
        else:
            assert o == want
