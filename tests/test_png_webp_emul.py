"""PNG in, WebP out (convert_in_memory to WebP with a PNG source): the PNG decode stages feeding the VP8 encoder, kernel sources compiled
for the CPU, against the oracle's statement (cso_png_to_webp); the same cases run on the device in test_zz_png_webp_gpu.py."""
import io

import pytest

from _util import emul_api, package, png_cases

PIL = pytest.importorskip("PIL.Image")
WEBP = 3


@pytest.fixture(scope="module")
def api():
    return emul_api()


def make_png(w, h, depth, ctype, pixels, extra=()):
    """a PNG file from raw rows (filter 0 on every row)"""
    import zlib

    def chunk(t, d):
        return len(d).to_bytes(4, "big") + t + d + zlib.crc32(t + d).to_bytes(4, "big")
    rb = len(pixels) // h
    raw = b"".join(b"\0" + pixels[y * rb:(y + 1) * rb] for y in range(h))
    ihdr = w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([depth, ctype, 0, 0, 0])
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + b"".join(chunk(t, d) for t, d in extra) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def extra_cases():
    """formats png_cases has no opaque example of: 16-bit truecolour, 2- and 4-bit grey, a short palette"""
    import numpy as np

    from gen_synth import synth_rgb
    rgb = synth_rgb(61, 70, 45, texture=4.0)
    wide = (rgb.astype(">u2") * 251 + 77)
    cases = [("rgb16_70x45", make_png(70, 45, 16, 2, wide.tobytes()))]
    g = rgb[:, :, 1]
    for depth in (2, 4):
        per = 8 // depth
        rows = []
        for y in range(45):
            v = (g[y] >> (8 - depth)).astype(np.uint32)
            v = np.concatenate([v, np.zeros((-len(v)) % per, np.uint32)]).reshape(-1, per)
            rows.append(bytes(int(sum(int(v[i, k]) << (8 - depth - k * depth) for k in range(per))) for i in range(v.shape[0])))
        cases.append((f"grey{depth}_70x45", make_png(70, 45, depth, 0, b"".join(rows))))
    idx = (g >> 5).astype(np.uint8)   # indices 0..7 with a five-entry PLTE: 5, 6, 7 decode as black
    plte = bytes(range(40, 55))
    cases.append(("short_plte_70x45", make_png(70, 45, 8, 3, idx.tobytes(), extra=[(b"PLTE", plte)])))
    return cases


def check(api, cases, quality, width=0, height=0):
    from _util import oracle_png_to_webp
    from oracle import oracle as O
    p = package().default_parameters(webp_quality=quality, width=width, height=height)
    outs = api.batch_convert([c[1] for c in cases], p, WEBP)
    done = 0
    for (name, src), out in zip(cases, outs):
        try:
            want = oracle_png_to_webp(src, quality, width, height)
        except O.PngError as e:
            if e.code == 10201 and not isinstance(out, Exception) and out[12:16] == b"VP8X":   # transparency: no oracle for the ALPH bytes, libwebp judges them
                check_alpha(api, name, src, quality, width, height)
                done += 1
                continue
            assert isinstance(out, Exception) and out.code == e.code, (name, out, e.code)
            continue
        assert not isinstance(out, Exception), (name, out)
        assert out == want, name
        im = PIL.open(io.BytesIO(out))
        im.load()
        assert im.format == "WEBP" and (width or height or im.size == PIL.open(io.BytesIO(src)).size)
        done += 1
    return done


def test_png_sources_equal_oracle(api):
    cases = png_cases()
    assert check(api, cases, 85) >= len(cases) // 2
    assert check(api, extra_cases(), 60) == 4


def riff_chunks(blob):
    assert blob[:4] == b"RIFF" and blob[8:12] == b"WEBP" and int.from_bytes(blob[4:8], "little") == len(blob) - 8
    at, out = 12, []
    while at < len(blob):
        n = int.from_bytes(blob[at + 4:at + 8], "little")
        out.append((blob[at:at + 4], blob[at + 8:at + 8 + n]))
        at += 8 + n + (n & 1)
    assert at == len(blob)
    return out


def check_alpha(api, name, src, quality, width=0, height=0):
    """a transparent PNG -> lossy WebP: extended format, the VP8 frame is the oracle's frame of the colour samples, the ALPH chunk (the device's VP8L coder over
    the alpha plane: no oracle, libwebp is the judge) gives back exactly the source's alpha"""
    import numpy as np

    from _util import png_expand8, png_resized_pixels
    from oracle import oracle as O
    out = api.convert_in_memory(src, package().default_parameters(webp_quality=quality, width=width, height=height), WEBP)
    P = O.png_decode(src)
    if width or height:
        pix, ctype, depth = png_resized_pixels(P, width, height)
        if depth == 16:
            pix = ((pix.astype(np.uint32) + 128) // 257).astype(np.uint8)
    elif P.im.depth == 16:
        v = P.rows().reshape(P.im.height, P.im.width, P.im.channels, 2).astype(np.uint32)
        pix = ((((v[..., 0] << 8) | v[..., 1]) + 128) // 257).astype(np.uint8)
    else:
        pix, ctype = png_expand8(P)
    colour, alpha = pix[:, :, :-1], pix[:, :, -1]
    rgb = np.ascontiguousarray(np.repeat(colour, 3, axis=2) if colour.shape[2] == 1 else colour)
    chunks = riff_chunks(out)
    assert [c[0] for c in chunks] == [b"VP8X", b"ALPH", b"VP8 "], name
    vp8x = chunks[0][1]
    h, w = alpha.shape
    assert vp8x[0] == 0x10 and int.from_bytes(vp8x[4:7], "little") == w - 1 and int.from_bytes(vp8x[7:10], "little") == h - 1
    want = O.webp_encode_rgb(rgb, quality)
    assert chunks[2][1] == riff_chunks(want)[0][1], name
    assert chunks[1][1][0] == 1
    got = PIL.open(io.BytesIO(out))
    assert got.mode == "RGBA" and got.size == (w, h), name
    got = np.asarray(got)
    assert np.array_equal(got[:, :, 3], alpha), name
    assert np.array_equal(got[:, :, :3], np.asarray(PIL.open(io.BytesIO(want)).convert("RGB"))), name


def test_transparency_becomes_an_alph_chunk(api):
    cases = dict(png_cases())
    for name in ("RGBA_97x61", "LA_97x61", "RGBA_300x2", "reduce_blocked_by_trns"):
        check_alpha(api, name, cases[name], 80)
    check_alpha(api, "RGBA_97x61 resized", cases["RGBA_97x61"], 70, width=40)
    # in one call with opaque pictures: the order stays, only the transparent ones are extended files
    outs = api.batch_convert([cases["RGB_97x61"], cases["RGBA_97x61"], cases["L_97x61"]], package().default_parameters(webp_quality=80), WEBP)
    assert [o[12:16] for o in outs] == [b"VP8 ", b"VP8X", b"VP8 "]


def test_resize_in_front(api):
    """--format webp --long-edge N over PNG sources (configs[3] shape): everything opaque resizes (16-bit at 16 bits, narrowed afterwards)"""
    cases = png_cases()
    assert check(api, cases, 85, width=50) >= 8
    pick = [c for c in cases if c[0] in ("RGB_97x61", "L_97x61", "RGB_200x150_3chunks")]
    assert check(api, pick, 70, height=100) == 3
    wide_trns = make_png(20, 10, 16, 0, bytes(range(200)) * 2, extra=[(b"tRNS", b"\0\7")])
    outs = api.batch_convert([wide_trns], package().default_parameters(webp_quality=80, width=40), WEBP)
    assert outs[0].code == 10201


def test_mixed_sources_keep_their_order(api):
    from _util import oracle_jpeg_to_webp
    from gen_synth import synth_jpeg
    from oracle import oracle as O
    cases = dict(png_cases())
    jpg = synth_jpeg(5, 88, 56, texture=6)
    blobs = [cases["L_97x61"], jpg, b"junk", cases["P_97x61"], jpg, cases["adam7_RGB_53x37"] if "adam7_RGB_53x37" in cases else cases["RGB_1x1"]]
    outs = api.batch_convert(blobs, package().default_parameters(webp_quality=70), WEBP)
    assert outs[0] == O.png_to_webp(blobs[0], 70) and outs[3] == O.png_to_webp(blobs[3], 70) and outs[5] == O.png_to_webp(blobs[5], 70)
    assert outs[1] == oracle_jpeg_to_webp(jpg, 70) and outs[4] == outs[1]
    assert outs[2].code == 10200


def test_damaged_pngs_convert_like_the_oracle_or_fail(api):
    """the PNG side of a conversion sees the same damaged inputs as the PNG path: what the oracle decodes gives the oracle's WebP, what it
    refuses is refused"""
    from oracle import oracle as O
    from test_png_emul import damaged_pngs
    blobs = damaged_pngs(23, 40)
    outs = api.batch_convert(blobs, package().default_parameters(webp_quality=75), WEBP)
    decoded = 0
    for k, (b, o) in enumerate(zip(blobs, outs)):
        transparent = False
        try:
            want = O.png_to_webp(b, 75)
        except O.PngError as e:
            want, transparent = None, e.code == 10201   # a picture with transparency that still decodes: the oracle has no ALPH coder, libwebp reads the file
        if transparent:
            assert o[12:16] == b"VP8X" and PIL.open(io.BytesIO(o)).mode == "RGBA", k
        elif want is None:
            assert isinstance(o, Exception), k
        else:
            assert o == want, k
            decoded += 1
    assert decoded >= 4
