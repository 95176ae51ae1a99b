"""Pins the CPU oracle of the lossless PNG row (oracle/png_oracle.c) against what can be pinned here: libpng (through
Pillow) for the decode side, zlib for the validity of the coder's streams.  Byte parity with oxipng/libdeflate is
unpinned (neither is available); see the oracle's header."""
import io
import zlib

import numpy as np
import pytest

from gen_synth import synth_png
from oracle import oracle as O

PIL = pytest.importorskip("PIL.Image")

MODES = ["RGB", "RGBA", "L", "LA", "P", "1", "I;16"]


def pil_pixels(data):
    im = PIL.open(io.BytesIO(data))
    im.load()
    return im.mode, np.asarray(im)


@pytest.mark.parametrize("mode", MODES)
def test_decode_equals_libpng(mode):
    for level in (0, 1, 6, 9):   # 0: stored blocks; small inputs at 1 use the fixed code
        data = synth_png(3, 97, 61, mode, compress_level=level)
        P = O.png_decode(data)
        im = PIL.open(io.BytesIO(data))
        rows = P.rows()
        if mode == "1":
            ref = np.packbits(np.asarray(im), axis=1)
        elif mode == "I;16":
            ref = np.asarray(im).astype(">u2").view(np.uint8).reshape(im.size[1], -1)
        else:
            ref = np.asarray(im).reshape(im.size[1], -1)
        assert np.array_equal(rows, ref), (mode, level)


def test_inflate_equals_zlib_on_every_block_type():
    rng = np.random.default_rng(1)
    blobs = [b"", b"a", bytes(rng.integers(0, 256, 70000, dtype=np.uint8)), b"abc" * 30000, bytes(rng.integers(0, 4, 50000, dtype=np.uint8))]
    for b in blobs:
        for level in (0, 1, 6, 9):
            z = zlib.compress(b, level)
            assert O.inflate_zlib(z, len(b)) == b
        c = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
        z = c.compress(b) + c.flush()
        assert O.inflate_zlib(z, len(b)) == b


def test_inflate_refuses_damage():
    z = bytearray(zlib.compress(b"hello world" * 100, 9))
    with pytest.raises(O.PngError):
        O.inflate_zlib(bytes(z[:10]), 1100)
    bad = bytearray(z); bad[0] = 0x79
    with pytest.raises(O.PngError):
        O.inflate_zlib(bytes(bad), 1100)


def test_coder_streams_are_valid_zlib():
    rng = np.random.default_rng(2)
    blobs = [b"", b"x", b"ab" * 5, bytes(rng.integers(0, 256, 40000, dtype=np.uint8)), b"\0" * 100000, bytes(rng.integers(0, 3, 70001, dtype=np.uint8)),
             bytes(np.repeat(rng.integers(0, 256, 3000, dtype=np.uint8), 11))]
    for b in blobs:
        z = O.deflate_zlib(b)
        assert zlib.decompress(z) == b
        assert O.inflate_zlib(z, len(b)) == b


@pytest.mark.parametrize("mode", MODES)
def test_optimised_file_keeps_the_pixels(mode):
    data = synth_png(5, 120, 80, mode)
    for level in (1, 3, 6):
        out, chosen = O.png_optimize(data, level)
        assert len(out) <= len(data)
        a, b = PIL.open(io.BytesIO(data)), PIL.open(io.BytesIO(out))
        if mode == "I;16":
            assert a.mode == b.mode and np.array_equal(np.asarray(a), np.asarray(b))
        else:   # a reduction (P2) may change the colour type, never what the pixels mean
            assert np.array_equal(np.asarray(a.convert("RGBA")), np.asarray(b.convert("RGBA")))
        if chosen < 0:
            assert out == data
        else:
            assert chosen in O.png_trials(level)


def test_filter_strategies_rebuild_the_rows():
    data = synth_png(6, 64, 48, "RGB")
    P = O.png_decode(data)
    rows = P.rows()
    for s in range(10):
        stream, choice = P.filtered(s)
        assert stream.reshape(48, -1)[:, 0].tolist() == choice.tolist()
        if s < 5:
            assert set(choice.tolist()) == {s}
        # a PNG assembled from this stream decodes to the same rows
        raw = stream.tobytes()
        png = data[:33] + len(zlib.compress(raw)).to_bytes(4, "big") + b"IDAT" + zlib.compress(raw)
        png += zlib.crc32(b"IDAT" + zlib.compress(raw)).to_bytes(4, "big") + b"\0\0\0\0IEND\xaeB`\x82"
        assert np.array_equal(np.asarray(PIL.open(io.BytesIO(png))).reshape(48, -1), rows)


def test_metadata_policy():
    from PIL import PngImagePlugin
    im = PIL.open(io.BytesIO(synth_png(7, 40, 30, "RGB")))
    info = PngImagePlugin.PngInfo()
    info.add_text("Comment", "hello")
    b = io.BytesIO()
    im.save(b, "PNG", pnginfo=info, dpi=(300, 300), compress_level=1)
    data = b.getvalue()
    stripped, _ = O.png_optimize(data, 3, keep_metadata=False)
    kept, _ = O.png_optimize(data, 3, keep_metadata=True)
    assert b"tEXt" not in stripped and b"pHYs" in stripped
    assert b"tEXt" in kept and b"pHYs" in kept
    assert PIL.open(io.BytesIO(kept)).info.get("Comment") == "hello"


def test_reductions_keep_what_the_pixels_mean():
    from _util import png_cases
    want = {"reduce_rgba_opaque": ("RGB", 2), "reduce_rgb_grey": ("L", 4), "reduce_rgba_grey_opaque": ("L", 6), "reduce_la_opaque": ("L", 2), "reduce_i16_narrow": ("L", 1),
            "reduce_rgba_nearly_opaque": ("RGBA", 0), "reduce_rgb_nearly_grey": ("P", 8),   # not grey, but at most 256 distinct pixels: indexed
            "greydepth_bw_L": ("1", 32), "greydepth_bw_RGB": ("1", 36), "greydepth_16_levels": ("L", 32), "greydepth_4_levels_RGBA": ("L", 38),
            "palette_rgb_few": ("P", 8), "palette_rgba_translucent": ("P", 8), "palette_two_colours": ("P", 8), "palette_257_colours": ("RGB", 0), "reduce_blocked_by_trns": ("RGB", 0)}
    for name, data in png_cases():
        if name not in want:
            continue
        P = O.png_decode(data)
        assert P.reduce() == want[name][1], name
        out, _ = O.png_optimize(data, 2)
        a, b = PIL.open(io.BytesIO(data)), PIL.open(io.BytesIO(out))
        assert b.mode == want[name][0], name
        if a.mode == "I;16":
            assert np.array_equal(np.asarray(a) >> 8, np.asarray(b))
        else:
            assert np.array_equal(np.asarray(a.convert("RGBA")), np.asarray(b.convert("RGBA"))), name


def palette_with_duplicates():
    """8-bit indexed files whose palettes list colours twice, carry translucent entries behind opaque ones, and have entries no pixel uses"""
    from test_png_webp_emul import make_png
    rng = np.random.default_rng(12)
    cases = []
    # 40 entries: 0-9 distinct, 10-19 repeat 0-9, 20-29 distinct again but unused, 30-39 distinct; alpha: entries 5, 15 (a duplicate pair) and 33 translucent
    pal = [tuple(int(v) for v in rng.integers(0, 256, 3)) for _ in range(40)]
    for k in range(10):
        pal[10 + k] = pal[k]
    trns = [255] * 40
    trns[5] = trns[15] = 80
    trns[33] = 0
    idx = rng.choice(np.array([k for k in range(40) if not 20 <= k < 30]), size=(31, 45)).astype(np.uint8)
    plte = b"".join(bytes(c) for c in pal)
    cases.append(("dupes_translucent", make_png(45, 31, 8, 3, idx.tobytes(), extra=[(b"PLTE", plte), (b"tRNS", bytes(trns[:34]))])))
    # same colour, different alpha: NOT duplicates
    pal2 = [(10, 20, 30), (10, 20, 30), (200, 100, 50), (200, 100, 50)]
    idx2 = rng.integers(0, 4, (9, 14)).astype(np.uint8)
    cases.append(("same_rgb_other_alpha", make_png(14, 9, 8, 3, idx2.tobytes(), extra=[(b"PLTE", b"".join(bytes(c) for c in pal2)), (b"tRNS", bytes([255, 128, 7, 7]))])))
    # all of a full-size palette used, but half of it twice over: 256 -> 128 entries
    pal3 = [tuple(int(v) for v in rng.integers(0, 256, 3)) for _ in range(128)] * 2
    idx3 = rng.integers(0, 256, (40, 40)).astype(np.uint8)
    cases.append(("full_palette_twice", make_png(40, 40, 8, 3, idx3.tobytes(), extra=[(b"PLTE", b"".join(bytes(c) for c in pal3))])))
    return cases


def test_indexed_palettes_lose_duplicates_and_put_translucent_entries_first():
    """P2, round 4: among the entries an indexed image uses, one per distinct (red, green, blue, alpha); those that are not opaque in front, so that tRNS is as
    short as it can be; the picture means what it meant"""
    for name, data in palette_with_duplicates():
        P = O.png_decode(data)
        assert P.reduce() == 64, name
        out, _ = O.png_optimize(data, 2)
        a, b = PIL.open(io.BytesIO(data)), PIL.open(io.BytesIO(out))
        assert b.mode == "P" and np.array_equal(np.asarray(a.convert("RGBA")), np.asarray(b.convert("RGBA"))), name
        plte = out[out.index(b"PLTE") + 4:][:int.from_bytes(out[out.index(b"PLTE") - 4:out.index(b"PLTE")], "big")]
        trns = b""
        if b"tRNS" in out:
            trns = out[out.index(b"tRNS") + 4:][:int.from_bytes(out[out.index(b"tRNS") - 4:out.index(b"tRNS")], "big")]
        n = len(plte) // 3
        cols = [(plte[3 * k], plte[3 * k + 1], plte[3 * k + 2], trns[k] if k < len(trns) else 255) for k in range(n)]
        assert len(set(cols)) == n, name                                            # no colour twice
        assert all(c[3] != 255 for c in cols[:len(trns)]) and all(c[3] == 255 for c in cols[len(trns):]), name   # tRNS holds the translucent entries and nothing else
        used = set(np.asarray(b).ravel().tolist())
        assert used == set(range(n)), name                                          # every entry is pointed at
    want = {"dupes_translucent": (20, 2), "same_rgb_other_alpha": (3, 2), "full_palette_twice": (128, 0)}   # (entries, tRNS bytes)
    for name, data in palette_with_duplicates():
        out, _ = O.png_optimize(data, 2)
        n = int.from_bytes(out[out.index(b"PLTE") - 4:out.index(b"PLTE")], "big") // 3
        nt = int.from_bytes(out[out.index(b"tRNS") - 4:out.index(b"tRNS")], "big") if b"tRNS" in out else 0
        assert (n, nt) == want[name], (name, n, nt)


def test_refusals():
    data = synth_png(8, 40, 30, "RGB")
    with pytest.raises(O.PngError):
        O.png_optimize(data[:100], 3)
    bad = bytearray(data); bad[20] ^= 1   # IHDR crc
    with pytest.raises(O.PngError):
        O.png_optimize(bytes(bad), 3)
    # an animated PNG: recognised, refused (not on this path)
    apng = data[:33] + (8).to_bytes(4, "big") + b"acTL" + bytes(8) + zlib.crc32(b"acTL" + bytes(8)).to_bytes(4, "big") + data[33:]
    with pytest.raises(O.PngError) as e:
        O.png_optimize(apng, 3)
    assert e.value.code == 10201


def test_adam7_inputs_decode_like_libpng():
    from _util import adam7_png
    for mode, w, h in [("RGB", 33, 21), ("RGBA", 9, 9), ("L", 5, 3), ("P", 40, 17), ("1", 37, 11), ("I;16", 12, 20), ("LA", 2, 1), ("RGB", 1, 1)]:
        plain = synth_png(9, w, h, mode)
        inter = adam7_png(plain)
        ref = PIL.open(io.BytesIO(inter))
        ref.load()   # libpng accepts the hand-made interlaced file
        assert ref.info.get("interlace") == 1
        assert np.array_equal(O.png_decode(inter).rows(), O.png_decode(plain).rows()), (mode, w, h)
        out, _ = O.png_optimize(inter, 2)
        got = PIL.open(io.BytesIO(out))
        assert out == inter or not got.info.get("interlace")   # a tiny file may come back unchanged ("already optimised")
        if mode != "I;16":
            assert np.array_equal(np.asarray(got.convert("RGBA")), np.asarray(ref.convert("RGBA")))


@pytest.mark.parametrize("rel", ["p0.png", "level_1_0/level_2_0/p2.png"])
def test_reference_samples_decode_like_libpng_and_recode_losslessly(reference_samples, rel):
    """the reference's own PNG samples (tests/golden/reference_samples): the oracle decodes them to libpng's rows, and what it writes for
    them under --lossless decodes, in libpng, to the same pixels -- never larger than the input"""
    import os
    data = open(os.path.join(reference_samples, rel), "rb").read()
    im = PIL.open(io.BytesIO(data)); im.load()
    assert np.array_equal(O.png_decode(data).rows(), np.asarray(im).reshape(im.size[1], -1))
    for level in (1, 3):
        out = O.png_optimize(data, level, False)[0]
        back = PIL.open(io.BytesIO(out)); back.load()
        assert len(out) <= len(data) and np.array_equal(np.asarray(back.convert(im.mode)), np.asarray(im))


def test_deflate_parse_is_close_to_libdeflate_on_the_same_filtered_bytes():
    """P4 measured against what oxipng -o3 links (libdeflate, level 11 / 12), when the system has it: the oracle's (= the device's) IDAT stream inflated and
    packed again by libdeflate -- same filtered bytes, only the parse differs.  tools/png_parse_gap.py prints the table (640 x 360: 1.000-1.003 x libdeflate-11 on
    textured pictures, 1.019 x on the smooth one -- 1.05 x before round 6's parse --, 9 % under zlib-9)."""
    import ctypes as C
    import ctypes.util
    import zlib
    name = ctypes.util.find_library("deflate")
    if not name:
        pytest.skip("no system libdeflate")
    D = C.CDLL(name)
    D.libdeflate_alloc_compressor.restype = C.c_void_p
    D.libdeflate_zlib_compress.restype = C.c_size_t
    D.libdeflate_zlib_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    from gen_synth import synth_png
    for seed, tex, bound in ((40, 3.0, 1.005), (41, 0.5, 1.03)):   # (round 6, with the min-cost-path parse: 0.999 and 0.969 at this size; 1.000 and 1.019 for the 640 x 360 pictures of tools/png_parse_gap.py)
        out, _ = O.png_optimize(synth_png(seed, 320, 240, "RGB", texture=tex), 3)
        at, z = 8, b""
        while at < len(out):
            n = int.from_bytes(out[at:at + 4], "big")
            if out[at + 4:at + 8] == b"IDAT":
                z += out[at + 8:at + 8 + n]
            at += 12 + n
        raw = zlib.decompress(z)
        comp = D.libdeflate_alloc_compressor(12)
        buf = C.create_string_buffer(len(raw) + 1024)
        n12 = D.libdeflate_zlib_compress(comp, raw, len(raw), buf, len(raw) + 1024)
        D.libdeflate_free_compressor(C.c_void_p(comp))
        assert n12 and len(z) <= bound * n12, (tex, len(z), n12)
        assert len(z) < len(zlib.compress(raw, 9))
