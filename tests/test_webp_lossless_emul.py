"""Lossless WebP OUTPUT (webp.lossless / --lossless on WebP files, and JPEG -> WebP conversions with it): the device's VP8L coder.  Its bytes
are not libwebp's (parity unpinned: libwebp's lossless coder is a serial search); what is pinned is the format's own invariant, checked with
libwebp itself (through Pillow): the file decodes to EXACTLY the pixels that went in -- and this repo's VP8L decoder agrees."""
import io
import os

import numpy as np
import pytest
from PIL import Image

from _util import emul_api, package
from gen_synth import synth_jpeg, synth_rgb
import test_webp_decode_emul as D


@pytest.fixture(scope="module")
def api():
    return emul_api()


def params(**kw):
    return package().default_parameters(**kw)


def check_vp8l(blob, want_rgb):
    assert blob[:4] == b"RIFF" and blob[8:16] == b"WEBPVP8L" and int.from_bytes(blob[4:8], "little") == len(blob) - 8 and len(blob) % 2 == 0
    im = Image.open(io.BytesIO(blob))
    got = np.asarray(im.convert("RGB"))
    assert got.shape == want_rgb.shape and np.array_equal(got, want_rgb)
    assert blob == oracle_vp8l(blob)


def oracle_vp8l(blob):
    """the bytes the oracle's statement of the VP8L coder (oracle/png_oracle.c cso_vp8l_encode) makes of the pixels `blob` decodes to (libwebp's reading): the
    device's / the emulation's file must BE that -- the coder is checked against a statement of its own, not against another build of its source.
    (A grey picture coded from one channel and from three equal ones is the same stream: subtract-green leaves the same residuals.)"""
    from oracle import oracle as O
    alpha = (int.from_bytes(blob[21:25], "little") >> 28) & 1
    px = np.asarray(Image.open(io.BytesIO(blob)).convert("RGBA" if alpha else "RGB"))
    return O.vp8l_encode(px.tobytes(), px.shape[1], px.shape[0], 4 if alpha else 3)


def sources():
    """lossless and lossy WebP sources: photographic, flat, tiny, off the 16 x 16 block grid, one pixel wide / high"""
    out = []
    for i, (w, h, tex) in enumerate([(64, 48, 0.0), (101, 67, 20.0), (17, 9, 40.0), (1, 1, 0.0), (320, 240, 30.0), (1, 37, 5.0), (53, 1, 5.0), (16, 16, 10.0), (33, 31, 60.0)]):
        out.append(D.lossless_of(synth_rgb(200 + i, w, h, texture=tex)))
    out.append(D.lossless_of(np.full((40, 30, 3), 77, np.uint8)))
    out.append(D.webp_of(7, 97, 61, 70, texture=20.0))
    return out


def test_emul_lossless_webp_round_trips_through_libwebp(api, reference_samples):
    srcs = sources() + [open(os.path.join(reference_samples, "w0.webp"), "rb").read()]
    outs = api.cs_batch_compress(srcs, params(webp_lossless=True))
    for src, out in zip(srcs, outs):
        assert isinstance(out, bytes), out
        check_vp8l(out, D.libwebp_rgb(src))
    # and this repo's own decoder reads them back
    for src, got in zip(srcs, api.webp_decode(list(outs))):
        assert np.array_equal(got, D.libwebp_rgb(src))


def test_emul_lossless_webp_sizes_are_sane(api):
    """no search for backward references: somewhat larger than libwebp's file, far smaller than the pixels"""
    rgb = synth_rgb(31, 320, 240, texture=5.0)   # measured: 8 % over libwebp on noisy content, 13 % at texture 2; smooth synthetic gradients (where
    src = D.lossless_of(rgb)                     # libwebp's backward references and colour cache pay) come out up to three times libwebp's size
    out = api.compress_in_memory(src, params(webp_lossless=True))
    assert len(out) < rgb.size * 0.7 and len(out) < len(src) * 1.25


def test_emul_jpeg_to_lossless_webp_and_resize(api):
    from oracle import oracle as O
    src = synth_jpeg(4, 120, 88, texture=30)
    out = api.convert_in_memory(src, params(webp_lossless=True), 3)
    want = np.asarray(Image.open(io.BytesIO(api.convert_in_memory(src, params(png_optimize=True), 1))).convert("RGB"))   # the same decode, through the PNG row
    check_vp8l(out, want)
    small = api.convert_in_memory(src, params(webp_lossless=True, width=60), 3)
    want = np.asarray(Image.open(io.BytesIO(api.convert_in_memory(src, params(png_optimize=True, width=60), 1))).convert("RGB"))
    check_vp8l(small, want)
    lossless_src = D.lossless_of(synth_rgb(5, 90, 70, texture=10.0))
    resized = api.compress_in_memory(lossless_src, params(webp_lossless=True, height=35))
    im = Image.open(io.BytesIO(resized))
    assert im.size == (45, 35)


def test_emul_lossless_webp_failures_stay_per_file(api):
    good = D.lossless_of(synth_rgb(6, 40, 30, texture=20.0))
    alpha = D.lossless_of(np.dstack([synth_rgb(7, 32, 24), np.full((24, 32), 128, np.uint8)]), "RGBA")
    outs = api.cs_batch_compress([good, good[:50], alpha, good], params(webp_lossless=True))
    assert [isinstance(o, Exception) for o in outs] == [False, True, False, False]   # the picture with transparency keeps it (tests/test_webp_decode_emul.py)
    check_vp8l(outs[0], D.libwebp_rgb(good)); check_vp8l(outs[3], D.libwebp_rgb(good))


def test_emul_png_to_lossless_webp(api):
    """--format webp --lossless over PNG sources: the PNG row's decode, the VP8L coder behind it.  8-bit and palette / low-depth opaque pictures must
    come back as exactly the pixels libpng (Pillow) reads, alpha channel / tRNS included (RGBA in libwebp's reading); a size resizes opaque pictures first."""
    from _util import png_cases
    from test_png_webp_emul import extra_cases
    cases = dict(png_cases())
    cases.update(dict(extra_cases()))
    exact = ["RGB_97x61", "L_97x61", "P_97x61", "1_97x61", "RGB_flat_64x48", "RGB_200x150_3chunks", "RGB_stored_input", "L_level1_input", "RGB_1x1", "L_1x300",
             "grey2_70x45", "grey4_70x45", "short_plte_70x45"]
    alpha = ["RGBA_97x61", "LA_97x61", "RGBA_300x2", "reduce_blocked_by_trns"]   # an alpha channel or a tRNS chunk: kept, as the picture's alpha
    outs = api.batch_convert([cases[n] for n in exact + alpha + ["I;16_97x61", "rgb16_70x45"]], params(webp_lossless=True), 3)
    for name, out in zip(exact, outs):
        assert isinstance(out, bytes), (name, out)
        check_vp8l(out, np.asarray(Image.open(io.BytesIO(cases[name])).convert("RGB")))
    for name, out in zip(alpha, outs[len(exact):]):
        assert isinstance(out, bytes) and out[8:16] == b"WEBPVP8L", (name, out)
        got = Image.open(io.BytesIO(out))
        assert got.mode == "RGBA", name
        assert np.array_equal(np.asarray(got), np.asarray(Image.open(io.BytesIO(cases[name])).convert("RGBA"))), name
        assert out == oracle_vp8l(out), name
    for name, out in zip(["I;16_97x61", "rgb16_70x45"], outs[len(exact) + len(alpha):]):   # 16-bit samples are narrowed (the lossy PNG -> WebP path pins the rule)
        assert isinstance(out, bytes) and Image.open(io.BytesIO(out)).size == Image.open(io.BytesIO(cases[name])).size, name
    # the same pixels as the lossy conversion's source: a JPEG made from the PNG at 4:4:4 q100 is not exact, so compare with PNG -> PNG resize instead
    small = api.convert_in_memory(cases["RGB_200x150_3chunks"], params(webp_lossless=True, width=80), 3)
    assert Image.open(io.BytesIO(small)).size == (80, 60)
    assert api.convert_in_memory(cases["RGB_97x61"], params(webp_lossless=True), 3) == outs[0]
    # pictures with transparency resize as well (the PNG row's own Lanczos passes over the interleaved samples): the pixels of the PNG -> PNG resize, which the oracle pins
    for name in alpha:
        small = api.convert_in_memory(cases[name], params(webp_lossless=True, width=40), 3)
        as_png = api.compress_in_memory(cases[name], params(png_optimize=True, width=40))
        got, want = Image.open(io.BytesIO(small)), Image.open(io.BytesIO(as_png))
        assert got.size == want.size and got.size[0] == 40 and np.array_equal(np.asarray(got.convert("RGBA")), np.asarray(want.convert("RGBA"))), name
