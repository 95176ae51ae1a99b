"""Pins the CPU oracle of the lossy WebP row to libwebp itself, which is executable in this container:
  W1  oracle/webp_oracle.c cso_webp_rgb_to_yuv   == WebPPictureImportRGB, bit for bit;
  W2 / W3  oracle/vp8enc_oracle.c                == WebPEncode with a default WebPConfig at the quality (the reference's call), BYTE FOR BYTE, for every
           libwebp here (1.2.0, 1.2.2, 1.6.0 agree with each other, so does what libwebp-sys 0.9.5 vendors in between), live when a libwebp is present
           and against the committed digests of libwebp's files otherwise (tests/golden/libwebp_vp8enc.json, made by tests/golden/make_libwebp_goldens.py)."""
import ctypes as C
import ctypes.util
import hashlib
import json
import os

import numpy as np
import pytest

from gen_synth import synth_rgb
from libwebp_pin import compare, libwebp_encode, libwebps
from oracle import oracle as O

PIL = pytest.importorskip("PIL.Image")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "libwebp_vp8enc.json")
SAMPLES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_samples")


def crop(seed, w, h, texture=4.0):
    return np.ascontiguousarray(synth_rgb(seed, w + 400, h + 300, texture=texture)[150:150 + h, 200:200 + w])


def pictures(big=False):
    """(name, rgb, quality): every quality class (error diffusion on / off above 98, the two branches of the quality curve), frame-edge shapes (one macroblock wide /
    high, 1 x 1, partial macroblocks), flat and noisy content (the i16 / i4 balance, flat-source and flatness penalties, skip-everything macroblocks), pictures
    of more than one statistics chunk, the reference's own sample pictures; big: configs[3]-sized ones and a noisy large one whose token books overflow 16 bits"""
    out = []
    for q in (0, 1, 10, 30, 50, 74, 75, 76, 80, 90, 95, 98, 99, 100):
        out.append(("q%d_200x120" % q, crop(0, 200, 120), q))
    for k, (w, h) in enumerate(((1, 1), (3, 5), (16, 16), (17, 17), (15, 33), (97, 61), (250, 16), (16, 200), (129, 130), (333, 251))):
        out.append(("s%dx%d_q85" % (w, h), crop(20 + k, w, h, texture=6.0), 85))
        out.append(("s%dx%d_q40_noisy" % (w, h), crop(40 + k, w, h, texture=30.0), 40))
    out.append(("flat_grey_96x64", np.full((64, 96, 3), 119, np.uint8), 85))
    g = np.zeros((80, 112, 3), np.uint8); g[:, :, 0] = np.arange(112)[None, :] * 2; g[:, :, 1] = np.arange(80)[:, None] * 3; g[:, :, 2] = 60
    out.append(("gradient_112x80", g, 75))
    out.append(("chunks_640x481_q85", crop(7, 640, 481, texture=10.0), 85))
    for name in ("w0.webp", "level_1_1/w1.webp", "p0.png"):
        path = os.path.join(SAMPLES, name)
        if os.path.exists(path):
            out.append(("sample_" + os.path.basename(name), np.ascontiguousarray(np.asarray(PIL.open(path).convert("RGB"))), 80))
    if big:
        for seed in range(3):
            out.append(("synth%d_1500x844_q85" % seed, np.ascontiguousarray(synth_rgb(seed, 1500, 844)), 85))
        rng = np.random.default_rng(5)
        noisy = (synth_rgb(3, 1920, 1080).astype(int) + rng.integers(-40, 40, (1080, 1920, 3))).clip(0, 255).astype(np.uint8)
        out.append(("noisy_1920x1080_q85", np.ascontiguousarray(noisy), 85))
    return out


def test_bytes_equal_libwebp_live():
    """every libwebp in the container, default configuration: the oracle's file IS libwebp's file"""
    libs = libwebps()
    if not libs:
        pytest.skip("no libwebp with the encoder API")
    for name, rgb, q in pictures():
        mine = O.vp8enc_encode_rgb(rgb, q)
        for ver, W in libs:
            ref = libwebp_encode(W, rgb, q)
            assert mine == ref, (name, ver, compare(ref, mine, verbose=False))


def test_bytes_equal_committed_libwebp_digests():
    gold = json.load(open(GOLDEN))["cases"]
    seen = 0
    for name, rgb, q in pictures():
        if name not in gold:
            continue
        mine = O.vp8enc_encode_rgb(rgb, q)
        assert len(mine) == gold[name]["bytes"] and hashlib.sha256(mine).hexdigest() == gold[name]["sha256"], name
        seen += 1
    assert seen >= 35


def test_big_pictures_equal_committed_libwebp_digests():
    """configs[3]'s size, and a noisy 1080p picture whose per-slot token counts pass 65534 (libwebp halves its 16-bit books on the way: order-dependent)"""
    gold = json.load(open(GOLDEN))["cases"]
    for name, rgb, q in pictures(big=True):
        if not (name.startswith("synth") or name.startswith("noisy")):
            continue
        mine = O.vp8enc_encode_rgb(rgb, q)
        assert hashlib.sha256(mine).hexdigest() == gold[name]["sha256"], name


def test_the_webp_crate_s_picture_set_up_imports_alike():
    """crate webp 0.3.1 sets use_argb = 1 before WebPPictureImportRGB; WebPEncode then converts ARGB -> YUV itself: the same file"""
    libs = libwebps()
    if not libs:
        pytest.skip("no libwebp with the encoder API")
    rgb = crop(5, 150, 90, texture=8.0)
    for _, W in libs:
        assert libwebp_encode(W, rgb, 85, use_argb=True) == libwebp_encode(W, rgb, 85)


def test_parser_reads_back_what_the_encoder_decided():
    """the stream parser (the tool the pin was built with) against the encoder's own trace: header fields, segment map, modes, every level"""
    for name, rgb, q in pictures()[12:30:3]:
        data, frame, mbs = O.vp8enc_encode_rgb(rgb, q, trace=True)
        f2, m2 = O.vp8_parse(data)
        assert f2.header() == frame.header() | {"quant": f2.header()["quant"], "filt": f2.header()["filt"], "nseg": f2.header()["nseg"]}, name
        if frame.num_segments > 1:
            assert list(f2.seg_quant) == list(frame.seg_quant) and list(f2.seg_filter) == list(frame.seg_filter)
        for k in ("is_i4", "uvmode", "bmodes", "levels"):
            assert np.array_equal(mbs[k], m2[k]), (name, k)
        if frame.update_map:
            assert np.array_equal(mbs["segment"], m2["segment"]), name
        assert bytes(f2.probas) == bytes(frame.probas)


def test_every_stream_decodes():
    for name, rgb, q in pictures()[::5]:
        im = PIL.open(__import__("io").BytesIO(O.vp8enc_encode_rgb(rgb, q)))
        im.load()
        assert im.size == (rgb.shape[1], rgb.shape[0]), name


# ------------------------------------------------------------------------------------------------ W1: the import
def libwebp_import_rgb(rgb):
    """libwebp's own RGB -> YUV 4:2:0 (WebPPictureImportRGB on a picture with use_argb = 0), read out of its WebPPicture: use_argb, colorspace, width, height
    (4 x int32), y / u / v pointers, y_stride, uv_stride -- the head of the struct in webp/encode.h since libwebp 0.5"""
    name = ctypes.util.find_library("webp")
    if not name:
        pytest.skip("no system libwebp")
    W = C.CDLL(name)
    if not hasattr(W, "WebPPictureImportRGB"):
        pytest.skip("libwebp without the encoder API")
    h, w, _ = rgb.shape
    buf = (C.c_uint8 * 1024)()
    assert W.WebPPictureInitInternal(buf, 0x020f)
    ints = C.cast(buf, C.POINTER(C.c_int32))
    assert ints[0] == 0
    ints[2], ints[3] = w, h
    rgb = np.ascontiguousarray(rgb)
    W.WebPPictureImportRGB.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert W.WebPPictureImportRGB(buf, rgb.ctypes.data, w * 3)
    ptrs = C.cast(buf, C.POINTER(C.c_void_p))
    cw, ch = (w + 1) // 2, (h + 1) // 2
    planes = []
    for k, (rows, cols, stride) in enumerate([(h, w, ints[10]), (ch, cw, ints[11]), (ch, cw, ints[11])]):
        planes.append(np.ctypeslib.as_array(C.cast(ptrs[2 + k], C.POINTER(C.c_uint8)), shape=(rows, stride))[:, :cols].copy())
    W.WebPPictureFree(buf)
    return planes


@pytest.mark.parametrize("w,h", [(64, 48), (33, 17), (1, 1), (2, 5), (255, 3), (16, 16), (161, 97)])
def test_rgb_to_yuv_is_libwebp_s_import(w, h):
    """W1 pinned to the library: luma and the gamma-weighted chroma of WebPPictureImportRGB, bit for bit, odd sizes included; the macroblock padding
    repeats each plane's last sample"""
    rng = np.random.default_rng(w * 1000 + h)
    for rgb in (rng.integers(0, 256, (h, w, 3), dtype=np.uint8), crop(w + h, w, h, texture=5.0)):
        y, u, v = O.webp_rgb_to_yuv(rgb)
        Y, U, V = libwebp_import_rgb(rgb)
        cw, ch = (w + 1) // 2, (h + 1) // 2
        assert np.array_equal(y[:h, :w], Y) and np.array_equal(u[:ch, :cw], U) and np.array_equal(v[:ch, :cw], V)
        assert (y[:, w:] == y[:, w - 1:w]).all() and (y[h:] == y[h - 1]).all() and (u[:, cw:] == u[:, cw - 1:cw]).all() and (v[ch:] == v[ch - 1]).all()


def test_gamma_tables_cover_every_sample_value():
    v = np.arange(256, dtype=np.uint8)
    rgb = np.repeat(np.repeat(np.stack([v, v[::-1], np.roll(v, 77)], -1)[None], 2, 0), 2, 1).reshape(2, 512, 3)
    y, u, vv = O.webp_rgb_to_yuv(rgb)
    Y, U, V = libwebp_import_rgb(rgb)
    assert np.array_equal(y[:2, :512], Y) and np.array_equal(u[:1, :256], U) and np.array_equal(vv[:1, :256], V)
