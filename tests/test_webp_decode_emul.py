"""WebP INPUTS (S1 / S3 with a WebP source; BASELINE configs[4] names WebP among the inputs): the VP8 key-frame decoder that runs on the
device, here through the emulation build.  The decoder is pinned to libwebp itself: every picture must equal, sample for sample, what
libwebp 1.6 (through Pillow) decodes -- the reference's own samples/w0.webp and level_1_1/w1.webp included.  What is made of the pixels
(WebP again at webp.quality, or JPEG / PNG for conversions) is checked against the oracle's encoders fed libwebp's pixels."""
import io
import os

import numpy as np
import pytest
from PIL import Image

from _util import device_quantiser, device_scan_script, emul_api, package
from gen_synth import synth_rgb


@pytest.fixture(scope="module")
def api():
    return emul_api()


def params(**kw):
    return package().default_parameters(**kw)


def webp_of(seed, w, h, quality, method=4, texture=0.0):
    b = io.BytesIO()
    Image.fromarray(synth_rgb(seed, w, h, texture=texture), "RGB").save(b, format="WEBP", quality=quality, method=method)
    return b.getvalue()


def libwebp_rgb(blob):
    return np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))


CASES = [(64, 48, 80, 4, 0.0), (101, 67, 50, 6, 10.0), (17, 9, 90, 0, 20.0), (1, 1, 75, 4, 0.0), (320, 240, 20, 4, 30.0), (200, 150, 95, 6, 40.0),
         (333, 222, 5, 2, 50.0), (16, 16, 100, 4, 5.0), (15, 33, 60, 3, 25.0), (256, 8, 70, 5, 15.0)]


def test_emul_reference_samples_decode_like_libwebp(api, reference_samples):
    blobs = [open(os.path.join(reference_samples, rel), "rb").read() for rel in ("w0.webp", "level_1_1/w1.webp")]   # plain VP8; VP8X + EXIF + XMP
    for blob, got in zip(blobs, api.webp_decode(blobs)):
        assert np.array_equal(got, libwebp_rgb(blob))


def test_emul_synthetic_files_decode_like_libwebp(api):
    """sizes down to 1 x 1 and off the macroblock grid, qualities 5..100 (filter strengths, segments, both block types), encoder methods 0..6"""
    blobs = [webp_of(30 + i, *c) for i, c in enumerate(CASES)]
    for c, blob, got in zip(CASES, blobs, api.webp_decode(blobs)):
        assert np.array_equal(got, libwebp_rgb(blob)), c


def lossless_of(arr, mode="RGB", **kw):
    b = io.BytesIO()
    Image.fromarray(arr, mode).save(b, format="WEBP", lossless=True, **kw)
    return b.getvalue()


def lossless_cases():
    """every tool of the VP8L format: photographic content (predictor + cross-colour + subtract-green, colour cache, LZ77 with near and far
    distances, meta prefix image on the larger ones), palettes of 2 / 3 / 5 / 17 / 200 colours (colour indexing with 8 / 4 / 2 / 1 pixels per
    sample), flat and tiny pictures (codes with a single symbol), encoder efforts 0..6, an opaque RGBA source"""
    rng = np.random.default_rng(77)
    out = []
    for i, (w, h, tex, method, q) in enumerate([(64, 48, 0.0, 4, 75), (101, 67, 20.0, 6, 100), (17, 9, 40.0, 0, 0), (1, 1, 0.0, 4, 75), (320, 240, 30.0, 4, 75),
                                                (333, 222, 50.0, 2, 50), (16, 16, 5.0, 3, 75), (15, 33, 25.0, 5, 90), (256, 8, 15.0, 1, 20), (640, 480, 10.0, 6, 100)]):
        out.append(("photo%d" % i, lossless_of(synth_rgb(60 + i, w, h, texture=tex), method=method, quality=q)))
    for ncol in (2, 3, 5, 17, 200):
        pal = rng.integers(0, 256, (ncol, 3), dtype=np.uint8)
        idx = rng.integers(0, ncol, (37, 53))
        idx[10:30, 5:40] = idx[10, 5]   # a flat patch: runs for the LZ77 layer
        out.append(("palette%d" % ncol, lossless_of(pal[idx])))
    out.append(("flat", lossless_of(np.full((40, 30, 3), 77, np.uint8))))
    g = np.zeros((64, 200, 3), np.uint8); g[..., 0] = np.arange(200)[None, :]; g[..., 1] = np.arange(64)[:, None] * 3; g[..., 2] = 255 - g[..., 0]
    out.append(("gradient", lossless_of(g)))
    out.append(("opaque_rgba", lossless_of(np.dstack([synth_rgb(90, 48, 40, texture=20.0), np.full((40, 48), 255, np.uint8)]), "RGBA")))
    noise = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    out.append(("noise", lossless_of(noise, method=6, quality=100)))
    return out


def test_emul_lossless_files_decode_like_libwebp(api):
    cases = lossless_cases()
    outs = api.webp_decode([b for _, b in cases])
    for (name, blob), got in zip(cases, outs):
        assert not isinstance(got, Exception), (name, got)
        assert np.array_equal(got, libwebp_rgb(blob)), name


def test_emul_damaged_lossless_streams_fail_alone(api):
    """truncated and bit-flipped VP8L streams: refused (or decoded to a picture of the declared size), never a crash, and the good file
    next to them is untouched"""
    rng = np.random.default_rng(5)
    cases = dict(lossless_cases())
    good = cases["photo4"]
    bad = []
    for name in ("photo4", "palette5", "photo1", "gradient"):
        blob = cases[name]
        for cut in (len(blob) // 3, len(blob) * 2 // 3, len(blob) - 3, 40):
            bad.append(blob[:cut])
        for _ in range(12):
            b = bytearray(blob)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(20, len(b)))] ^= 1 << int(rng.integers(0, 8))
            bad.append(bytes(b))
    outs = api.webp_decode([good] + bad + [good])
    assert np.array_equal(outs[0], libwebp_rgb(good)) and np.array_equal(outs[-1], libwebp_rgb(good))
    for blob, o in zip(bad, outs[1:-1]):
        assert isinstance(o, Exception) or o.ndim == 3


def test_emul_damaged_and_unsupported_inputs_fail_alone(api):
    good = webp_of(3, 64, 48, 80)
    lossless = io.BytesIO(); Image.fromarray(np.dstack([synth_rgb(4, 32, 24), np.full((24, 32), 128, np.uint8)]), "RGBA").save(lossless, format="WEBP", lossless=True)   # VP8L, not opaque
    alpha = io.BytesIO(); Image.fromarray(np.dstack([synth_rgb(5, 32, 24), np.full((24, 32), 128, np.uint8)]), "RGBA").save(alpha, format="WEBP", quality=80)
    outs = api.webp_decode([good, good[:60], lossless.getvalue(), alpha.getvalue(), good[:12] + b"JUNK" + good[16:], good])
    assert [isinstance(o, Exception) for o in outs] == [False, True, False, False, True, False]
    assert outs[2].shape == (24, 32, 4) and outs[3].shape == (24, 32, 4) and outs[1].code == 40100   # transparency: RGBA (test_emul_transparent_files_decode_like_libwebp)
    assert np.array_equal(outs[0], libwebp_rgb(good)) and np.array_equal(outs[5], libwebp_rgb(good))
    cut = good[:len(good) * 2 // 3]   # data running out inside the token partitions: refused by both, or decoded alike
    got = api.webp_decode([cut])[0]
    try:
        ref = libwebp_rgb(cut)
    except Exception:
        ref = None
    assert isinstance(got, Exception) or (ref is not None and got.shape == ref.shape)


def test_emul_compress_and_convert_from_webp(api, reference_samples):
    """compress_in_memory on a WebP = decode, encode again at webp.quality; convert_in_memory WebP -> JPEG / PNG = decode, then the JPEG / PNG
    rows from pixels.  Oracle: libwebp's pixels through oracle/webp_oracle.c, jpeg_oracle.c, png_oracle.c."""
    from oracle import oracle as O
    w0 = open(os.path.join(reference_samples, "w0.webp"), "rb").read()
    srcs = [w0, webp_of(8, 97, 61, 70, texture=20.0), lossless_of(synth_rgb(9, 83, 59, texture=15.0))]   # lossy (VP8) and lossless (VP8L) sources
    for src in srcs:
        rgb = np.ascontiguousarray(libwebp_rgb(src))
        assert api.compress_in_memory(src, params(webp_quality=60)) == O.webp_encode_rgb(rgb, 60)
        want_jpeg = O.pixels_to_jpeg(rgb, O.params(quality=75, progressive=1, subsampling=420, qtable_profile=3, marker_style=1, scan_script=device_scan_script(), **device_quantiser()), 0, 0)
        assert api.convert_in_memory(src, params(jpeg_quality=75), 0) == want_jpeg
        got_png = api.convert_in_memory(src, params(png_optimize=True), 1)
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(got_png)).convert("RGB")), rgb)
    # a size on the way: the JPEG row's Lanczos branch resamples the decoded pixels
    rgb = np.ascontiguousarray(libwebp_rgb(srcs[1]))
    want = O.pixels_to_jpeg(rgb, O.params(quality=80, progressive=1, subsampling=420, qtable_profile=3, marker_style=1, scan_script=device_scan_script(), **device_quantiser()), 48, 0)
    assert api.convert_in_memory(srcs[1], params(jpeg_quality=80, width=48), 0) == want
    got_png = api.convert_in_memory(srcs[1], params(png_optimize=True, width=48), 1)
    assert Image.open(io.BytesIO(got_png)).size[0] == 48
    with pytest.raises(package().CaesiumError) as e:
        api.convert_in_memory(srcs[1], params(), 3)
    assert e.value.code == 10407   # same format
    # mixed batch through the one entry point: order kept, every type served
    from gen_synth import synth_jpeg, synth_png
    from _util import oracle_lossy
    j = synth_jpeg(9, 80, 64)
    outs = api.cs_batch_compress([srcs[1], j, srcs[0]], params(webp_quality=60, jpeg_quality=80))
    assert outs[1] == oracle_lossy(j) and outs[0] == O.webp_encode_rgb(np.ascontiguousarray(libwebp_rgb(srcs[1])), 60) and outs[2][:4] == b"RIFF"


def test_emul_compress_to_size_on_webp(api):
    """--max-size on a WebP: libcaesium's walk over webp.quality (80, then bisection, 2 % tolerance, ten tries), every try a decode + encode"""
    from oracle import oracle as O
    from test_pipeline_emul import reference_size_walk
    src = webp_of(12, 160, 120, 90, texture=30.0)
    rgb = np.ascontiguousarray(libwebp_rgb(src))
    full = len(O.webp_encode_rgb(rgb, 80))
    target = full * 6 // 10
    qs, want = reference_size_walk(src, target, encode=lambda s, q: O.webp_encode_rgb(rgb, q))
    got = api.compress_to_size_in_memory(src, params(), target)
    assert got == want and len(qs) > 1


def test_emul_webp_metadata_carried_over(api, reference_samples):
    """-e on WebP -> WebP: the source's ICC profile and EXIF ride in an extended-format file (VP8X flags, ICCP in front of the frame, EXIF
    behind it), the frame itself is the one written without -e; no metadata in the source, or no -e: the simple format"""
    b = io.BytesIO()
    icc = bytes(range(256)) * 2 + b"x"     # odd length: chunk padding
    exif = b"Exif\x00\x00MM\x00*\x00\x00\x00\x08\x00\x00"
    Image.fromarray(synth_rgb(12, 75, 49, texture=10.0), "RGB").save(b, format="WEBP", quality=70, icc_profile=icc, exif=exif)
    src = b.getvalue()
    plain = api.compress_in_memory(src, params(webp_quality=60))
    kept = api.compress_in_memory(src, params(webp_quality=60, keep_metadata=True))
    assert plain[12:16] == b"VP8 " and kept[12:16] == b"VP8X" and kept[20] == 0x28
    assert int.from_bytes(kept[4:8], "little") == len(kept) - 8
    assert int.from_bytes(kept[24:27], "little") == 74 and int.from_bytes(kept[27:30], "little") == 48
    im = Image.open(io.BytesIO(kept)); im.load()
    assert im.info["icc_profile"] == icc and im.info["exif"] == Image.open(io.BytesIO(src)).info["exif"] and len(im.info["exif"]) >= 10
    assert np.array_equal(np.asarray(im.convert("RGB")), libwebp_rgb(plain))
    assert plain[12:] in kept   # the VP8 chunk is carried whole
    w1 = open(os.path.join(reference_samples, "level_1_1/w1.webp"), "rb").read()   # VP8X + EXIF + XMP
    kept = api.compress_in_memory(w1, params(webp_quality=60, keep_metadata=True))
    im = Image.open(io.BytesIO(kept)); im.load()
    assert kept[20] == 0x08 and im.info["exif"] == Image.open(io.BytesIO(w1)).info["exif"] and b"XMP " not in kept
    w0 = open(os.path.join(reference_samples, "w0.webp"), "rb").read()
    assert api.compress_in_memory(w0, params(webp_quality=60, keep_metadata=True)) == api.compress_in_memory(w0, params(webp_quality=60))
    # --lossless -e: the VP8L coder's file carries them as well (libcaesium's webp::compress sets ICC / EXIF whatever the coder); the canvas
    # comes from the VP8L header, the VP8X alpha flag from its alpha_is_used bit
    plain_l = api.compress_in_memory(src, params(webp_lossless=True))
    kept_l = api.compress_in_memory(src, params(webp_lossless=True, keep_metadata=True))
    assert plain_l[12:16] == b"VP8L" and kept_l[12:16] == b"VP8X" and kept_l[20] == 0x28 and plain_l[12:] in kept_l
    assert int.from_bytes(kept_l[24:27], "little") == 74 and int.from_bytes(kept_l[27:30], "little") == 48 and int.from_bytes(kept_l[4:8], "little") == len(kept_l) - 8
    im = Image.open(io.BytesIO(kept_l)); im.load()
    assert im.info["icc_profile"] == icc and im.info["exif"] == Image.open(io.BytesIO(src)).info["exif"]
    assert np.array_equal(np.asarray(im.convert("RGB")), np.asarray(Image.open(io.BytesIO(plain_l)).convert("RGB")))
    rgba = synth_rgb(13, 40, 30, texture=5.0)
    a4 = np.dstack([rgba, np.where(np.arange(40)[None, :] < 20, 255, 90).astype(np.uint8).repeat(30, 0).reshape(30, 40)])
    bb = io.BytesIO(); Image.fromarray(a4, "RGBA").save(bb, format="WEBP", lossless=True, exif=exif)
    kept_a = api.compress_in_memory(bb.getvalue(), params(webp_lossless=True, keep_metadata=True))
    assert kept_a[12:16] == b"VP8X" and kept_a[20] == 0x18          # EXIF + alpha
    im = Image.open(io.BytesIO(kept_a)); im.load()
    assert im.mode == "RGBA" and np.array_equal(np.asarray(im), a4)


def transparent_files():
    """WebP files with transparency as libwebp writes them: lossy frames with an ALPH chunk (lossless alpha with libwebp's choice of filter; alpha_quality
    below 100 adds its level reduction, which needs no undoing), lossless pictures with an alpha channel; smooth, noisy, binary and constant alpha"""
    out = []
    rng = np.random.default_rng(5)
    for k, (w, h) in enumerate([(64, 48), (97, 61), (33, 17), (1, 1), (16, 16), (130, 5)]):
        rgb = synth_rgb(300 + k, w, h, texture=10.0)
        yy, xx = np.mgrid[0:h, 0:w]
        alphas = {"ramp": ((xx * 255) // max(1, w - 1)).astype(np.uint8), "noise": rng.integers(0, 256, (h, w), dtype=np.uint8),
                  "mask": np.where((xx // 5 + yy // 3) % 2 == 0, 0, 255).astype(np.uint8), "half": np.full((h, w), 128, np.uint8)}
        for name, a in alphas.items():
            im = Image.fromarray(np.dstack([rgb, a]), "RGBA")
            for label, kw in (("lossy", dict(quality=80)), ("lossy_aq50", dict(quality=60, alpha_quality=50)), ("lossy_m6", dict(quality=90, method=6)), ("lossless", dict(lossless=True))):
                if k > 1 and label in ("lossy_aq50", "lossy_m6"):
                    continue
                b = io.BytesIO()
                im.save(b, format="WEBP", **kw)
                out.append(("%s_%s_%dx%d" % (label, name, w, h), b.getvalue()))
    return out


def test_emul_transparent_files_decode_like_libwebp(api):
    files = transparent_files()
    outs = api.webp_decode([f[1] for f in files])
    filters = set()
    for (name, blob), got in zip(files, outs):
        assert not isinstance(got, Exception), (name, got)
        ref = Image.open(io.BytesIO(blob))
        want = np.asarray(ref.convert("RGBA"))
        if (want[:, :, 3] == 255).all():   # libwebp found the plane opaque after its level reduction: an RGB picture for both
            assert got.shape[2] == 3 and np.array_equal(got, want[:, :, :3]), name
            continue
        assert got.shape == want.shape and np.array_equal(got, want), name
        at = blob.find(b"ALPH")
        if at > 0:
            filters.add((blob[at + 8] >> 2) & 3)
    assert len(filters) >= 2, filters   # the files between them use more than one of the alpha filters


def test_emul_transparent_sources_keep_their_alpha(api):
    """-q / --lossless / --format png / --format jpeg on WebP files with transparency: the alpha plane survives exactly (it is coded losslessly either way), the
    colour goes through the same encoders as an opaque picture's"""
    from _util import package
    pkg = package()
    rgb = synth_rgb(41, 80, 56, texture=12.0)
    yy, xx = np.mgrid[0:56, 0:80]
    a = np.where((xx - 40) ** 2 + (yy - 28) ** 2 < 500, 255, ((xx * 3) % 256)).astype(np.uint8)
    files = {}
    for label, kw in (("lossy", dict(quality=85)), ("lossless", dict(lossless=True))):
        b = io.BytesIO()
        Image.fromarray(np.dstack([rgb, a]), "RGBA").save(b, format="WEBP", **kw)
        files[label] = b.getvalue()
    opaque = webp_of(3, 64, 48, 80)
    for label, src in files.items():
        src_rgba = np.asarray(Image.open(io.BytesIO(src)).convert("RGBA"))
        # lossy again: VP8X + ALPH + VP8, alpha exact, colour = what the opaque path makes of the same RGB
        outs = api.cs_batch_compress([opaque, src, opaque], pkg.default_parameters(webp_quality=70))
        assert outs[0] == outs[2] and outs[0][12:16] == b"VP8 "
        out = outs[1]
        assert out[12:16] == b"VP8X" and out[30:34] == b"ALPH", label
        got = np.asarray(Image.open(io.BytesIO(out)).convert("RGBA"))
        assert np.array_equal(got[:, :, 3], src_rgba[:, :, 3]), label
        plain = io.BytesIO()
        Image.fromarray(src_rgba[:, :, :3], "RGB").save(plain, format="WEBP", lossless=True)
        want = api.compress_in_memory(plain.getvalue(), pkg.default_parameters(webp_quality=70))
        assert np.array_equal(got[:, :, :3], np.asarray(Image.open(io.BytesIO(want)).convert("RGB"))), label
        # losslessly: exactly the RGBA
        out = api.compress_in_memory(src, pkg.default_parameters(webp_lossless=True))
        assert out[8:16] == b"WEBPVP8L" and np.array_equal(np.asarray(Image.open(io.BytesIO(out)).convert("RGBA")), src_rgba), label
        # to PNG (lossless trials): exactly the RGBA; to JPEG: the plane is dropped
        out = api.convert_in_memory(src, pkg.default_parameters(png_optimize=True), 1)
        im = Image.open(io.BytesIO(out))
        assert im.format == "PNG" and np.array_equal(np.asarray(im.convert("RGBA")), src_rgba), label
        out = api.convert_in_memory(src, pkg.default_parameters(jpeg_quality=90), 0)
        assert Image.open(io.BytesIO(out)).mode == "RGB"
        # with a resize: colour and plane through the same Lanczos branch (the plane as a grey picture)
        outs = api.cs_batch_compress([src, opaque], pkg.default_parameters(webp_quality=70, width=40))
        small = Image.open(io.BytesIO(outs[0]))
        assert small.mode == "RGBA" and small.size == (40, 28) and isinstance(outs[1], bytes), label
        plane = io.BytesIO()
        Image.fromarray(np.repeat(src_rgba[:, :, 3:4], 3, axis=2), "RGB").save(plane, format="WEBP", lossless=True)   # the plane as a grey picture, resized the same way
        want = np.asarray(Image.open(io.BytesIO(api.compress_in_memory(plane.getvalue(), pkg.default_parameters(webp_lossless=True, width=40)))).convert("RGB"))[:, :, 0]
        assert np.array_equal(np.asarray(small)[:, :, 3], want), label
        # the RGBA consumers (lossless WebP, PNG): the two resized halves joined again -- the colour of the opaque picture with the same RGB, the plane as above
        colour = io.BytesIO()
        Image.fromarray(src_rgba[:, :, :3], "RGB").save(colour, format="WEBP", lossless=True)
        want_rgb = np.asarray(Image.open(io.BytesIO(api.compress_in_memory(colour.getvalue(), pkg.default_parameters(webp_lossless=True, width=40)))).convert("RGB"))
        want_rgba = np.dstack([want_rgb, want])
        outs = api.cs_batch_compress([src, opaque, src], pkg.default_parameters(webp_lossless=True, width=40))
        assert outs[0] == outs[2] and outs[0][8:16] == b"WEBPVP8L" and isinstance(outs[1], bytes), label
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(outs[0])).convert("RGBA")), want_rgba), label
        out = api.convert_in_memory(src, pkg.default_parameters(png_optimize=True, width=40), 1)
        im = Image.open(io.BytesIO(out))
        assert im.format == "PNG" and im.size == (40, 28) and np.array_equal(np.asarray(im.convert("RGBA")), want_rgba), label
    # metadata travels with the alpha: ICCP in front of ALPH, EXIF behind the frame
    meta = io.BytesIO()
    Image.fromarray(np.dstack([rgb, a]), "RGBA").save(meta, format="WEBP", quality=85, exif=b"Exif\0\0MM\0*\0\0\0\x08\0\0", icc_profile=b"fake profile bytes")
    out = api.compress_in_memory(meta.getvalue(), pkg.default_parameters(webp_quality=70, keep_metadata=True))
    im = Image.open(io.BytesIO(out))
    assert im.mode == "RGBA" and im.info.get("icc_profile") == b"fake profile bytes" and out[20] & 0x38 == 0x38


def test_boolean_decoder_equals_the_byte_wise_form(tmp_path):
    """vp8_dec.h's BoolDec (32 bits of look-ahead, one shift per symbol) decides like RFC 6386's two-byte form, bit for bit, and reports the end of the data at the same symbol"""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "booldec_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(here, "..", "include"), os.path.join(here, "booldec_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, check=True)
    assert r.stdout.decode().startswith("ok "), r.stdout


def test_emul_damaged_transparent_files_fail_alone(api):
    """bit flips inside the ALPH chunk and in non-opaque VP8L streams, truncated chunks, a plane shorter than the picture: refused per file or decoded to a
    picture of the right shape, never past a buffer (tools/asan_emul.sh runs this under the sanitizers)"""
    rng = np.random.default_rng(11)
    files = [f for f in transparent_files() if "64x48" in f[0] or "33x17" in f[0]]
    good = webp_of(3, 64, 48, 80)
    bad = []
    for name, blob in files:
        at = blob.find(b"ALPH")
        lo, hi = (at + 8, at + 8 + int.from_bytes(blob[at + 4:at + 8], "little")) if at > 0 else (20, len(blob))
        for _ in range(3):
            b = bytearray(blob)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
            bad.append(bytes(b))
        if at > 0:   # the chunk cut short (its size field and the RIFF size adjusted), and a raw plane that is too small
            n = int.from_bytes(blob[at + 4:at + 8], "little")
            keep = max(1, n // 2)
            cut = blob[:at + 4] + keep.to_bytes(4, "little") + blob[at + 8:at + 8 + keep] + (b"\0" if keep & 1 else b"") + blob[at + 8 + n + (n & 1):]
            bad.append(cut[:4] + (len(cut) - 8).to_bytes(4, "little") + cut[8:])
            raw = blob[:at + 4] + (5).to_bytes(4, "little") + b"\0abcd\0" + blob[at + 8 + n + (n & 1):]
            bad.append(raw[:4] + (len(raw) - 8).to_bytes(4, "little") + raw[8:])
    outs = api.webp_decode([good] + bad + [good])
    assert np.array_equal(outs[0], libwebp_rgb(good)) and np.array_equal(outs[-1], libwebp_rgb(good))
    refused = 0
    for o in outs[1:-1]:
        assert isinstance(o, Exception) or (o.ndim == 3 and o.shape[2] in (3, 4))
        refused += isinstance(o, Exception)
    assert refused >= 3
