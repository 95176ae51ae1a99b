"""Pins the CPU oracle (oracle/jpeg_oracle.c) before anything is compared against it.

 * committed golden vectors made by libjpeg-turbo (tests/golden/make_golden.py),
 * live libjpeg-turbo through Pillow (same ISLOW pipeline as mozjpeg),
 * the reference's own fixtures samples/j0.JPG and samples/level_1_0/j1.jpg (only here, where
   /root/reference exists): entropy layer must round-trip byte-identically.
"""
import glob
import hashlib
import io
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from gen_synth import synth_jpeg, synth_rgb

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PIL = pytest.importorskip("PIL.Image")


def pil_ycc(data):
    im = PIL.open(io.BytesIO(data))
    im.draft("YCbCr", im.size)
    im.load()
    return np.asarray(im)


def test_quality_table_q80_matches_survey():
    # mozjpeg base table #3 at q=80 (s=40)
    t = O.quality_tables(80, 3)
    assert list(t[0][:8]) == [6, 6, 6, 7, 10, 15, 22, 34]
    assert t[0][63] == 167 and np.array_equal(t[0], t[1])
    t85 = O.quality_tables(85, 3)
    assert list(t85[0][:8]) == [5, 5, 5, 5, 8, 11, 17, 26] and t85[0][63] == 125


def test_golden_vectors_libjpeg_turbo():
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    assert man["cases"]
    for case in man["cases"]:
        src = open(os.path.join(GOLD, case["name"] + ".src.jpg"), "rb").read()
        for q in case["qualities"]:
            want = open(os.path.join(GOLD, f"{case['name']}.q{q}.jpg"), "rb").read()
            got = O.jpeg_compress(src, O.params(quality=q, progressive=1, subsampling=420, qtable_profile=3, marker_style=0))
            assert got == want, (case["name"], q)


@pytest.mark.parametrize("w,h,ss", [(101, 67, 2), (104, 72, 1), (99, 73, 0), (50, 34, 2), (17, 9, 2), (8, 8, 0), (1, 1, 2), (640, 360, 2)])
@pytest.mark.parametrize("prog", [False, True])
def test_decode_pixels_equal_libjpeg_turbo(w, h, ss, prog):
    src = synth_jpeg(11, w, h, subsampling=ss, progressive=prog, texture=30)
    assert np.array_equal(O.decode(src).pixels(), pil_ycc(src))


@pytest.mark.parametrize("ssn,ss", [(444, 0), (422, 1), (420, 2)])
@pytest.mark.parametrize("prog", [1, 0])
def test_transcode_bytes_equal_libjpeg_turbo(ssn, ss, prog):
    for (w, h, q, src_ss) in [(101, 67, 80, 0), (320, 256, 30, 0), (33, 31, 97, 0), (16, 16, 51, 0), (99, 73, 80, 1), (99, 73, 60, 2), (17, 9, 90, 2), (18, 11, 75, 1)]:
        src = synth_jpeg(5, w, h, subsampling=src_ss, texture=25)
        ref = pil_ycc(src)
        out = O.jpeg_compress(src, O.params(quality=q, progressive=prog, subsampling=ssn, marker_style=0, qtable_profile=0))
        b = io.BytesIO()
        PIL.fromarray(ref, "YCbCr").save(b, format="JPEG", quality=q, subsampling=ss, progressive=bool(prog), optimize=True)
        assert out == b.getvalue(), (w, h, q)


def test_grayscale_and_restart_sources():
    g = PIL.fromarray(synth_rgb(7, 203, 155)).convert("L")
    b = io.BytesIO(); g.save(b, format="JPEG", quality=90); src = b.getvalue()
    for prog in (1, 0):
        out = O.jpeg_compress(src, O.params(quality=75, progressive=prog, marker_style=0, qtable_profile=0))
        b = io.BytesIO(); PIL.fromarray(pil_ycc(src), "L").save(b, format="JPEG", quality=75, progressive=bool(prog), optimize=True)
        assert out == b.getvalue()
    src = synth_jpeg(9, 320, 256, restart_rows=1)
    ci = O.decode(src)
    assert ci.im.restart_interval == 20
    assert np.array_equal(ci.pixels(), pil_ycc(src))


def test_1080p_full_size_byte_parity():
    src = synth_jpeg(0)
    qt = O.quality_tables(80, 3)
    out = O.jpeg_compress(src, O.params(quality=80, marker_style=0))
    b = io.BytesIO()
    PIL.fromarray(pil_ycc(src), "YCbCr").save(b, format="JPEG", qtables=[list(map(int, qt[0]))] * 2, subsampling=2, progressive=True, optimize=True)
    assert out == b.getvalue()


def test_lossless_transcode_invariants():
    src = synth_jpeg(21, 333, 222, texture=20)
    a = O.decode(src)
    out = O.jpeg_compress(src, O.params(progressive=1), lossless=True)
    b = O.decode(out)
    assert len(out) < len(src)
    for c in range(3):
        assert np.array_equal(a.coefs(c), b.coefs(c))
    assert np.array_equal(a.pixels(), b.pixels())


def test_truncated_sequential_streams_equal_libjpeg_turbo():
    """libjpeg's insufficient-data rule (jdhuff.c): the MCU in which the data runs out is finished on zero bits, every later
    MCU up to the next restart marker stays zero.  (Progressive files cut short are NOT pinned: libjpeg then also applies
    inter-block smoothing, which this oracle does not restate.)"""
    from PIL import ImageFile
    old = ImageFile.LOAD_TRUNCATED_IMAGES
    ImageFile.LOAD_TRUNCATED_IMAGES = True
    try:
        for kw in ({}, {"restart_rows": 1}, {"subsampling": 0}, {"subsampling": 1}, {"optimize": True}):
            src = synth_jpeg(3, 200, 150, texture=30, **kw)
            for frac in (0.3, 0.5, 0.66, 0.9, 0.99):
                cut = src[:int(len(src) * frac)] + b"\xff\xd9"
                assert np.array_equal(O.decode(cut).pixels(), pil_ycc(cut)), (kw, frac)
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = old


def test_bad_inputs_fail_cleanly():
    with pytest.raises(O.OracleError):
        O.decode(b"")
    with pytest.raises(O.OracleError):
        O.decode(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    src = synth_jpeg(1, 64, 48)
    with pytest.raises(O.OracleError):
        O.decode(src[:200])


@pytest.mark.parametrize("rel,style", [("j0.JPG", 1), ("level_1_0/j1.jpg", 0)])
def test_reference_fixture_entropy_roundtrip(reference_samples, rel, style):
    """samples/j0.JPG (mozjpeg-made, 8 optimised scans, merged DQT/DHT) and j1.jpg (stock 10 scans):
    decode to coefficients, re-encode with the file's own script -> DQT..EOI byte-identical."""
    man = json.load(open(os.path.join(GOLD, "manifest.json")))["reference"][rel]
    d = open(os.path.join(reference_samples, rel), "rb").read()
    assert hashlib.sha256(d).hexdigest() == man["sha256_file"]
    ci = O.decode(d)
    out = ci.encode(O.params(progressive=1, marker_style=style), script=ci.scans())
    tail = out[out.index(b"\xff\xdb"):]
    assert len(tail) == man["tail_len"]
    assert hashlib.sha256(tail).hexdigest() == man["sha256_dqt_to_eoi"]
    assert tail == d[d.index(b"\xff\xdb"):]
    if rel == "j0.JPG":
        # j0's DQT == mozjpeg table #3 at libjpeg scale 98 (quality 51)
        assert np.array_equal(ci.qtable(0), O.quality_tables(51, 3)[0])
        assert ci.scans() == O.stock_script(3, 1)
    else:
        assert ci.scans() == O.stock_script(3, 0)


def test_scan_search_is_pinned_by_the_reference_fixture(reference_samples):
    """mozjpeg's optimize_scans search, restated from recall (oracle/jpeg_oracle.c cso_search_progression), run on the coefficients of the
    reference's own mozjpeg-made sample: it must pick j0's script WITHOUT being given it, and the lossless transcode that libcaesium does
    with --lossless (BASELINE configs[0]) then reproduces j0's own DQT..EOI byte for byte (SURVEY 8c-4 iii).  On j1 (made by plain
    libjpeg: stock script) the search picks another script and the file gets smaller."""
    d = open(os.path.join(reference_samples, "j0.JPG"), "rb").read()
    out = O.jpeg_compress(d, O.params(progressive=1, marker_style=1, scan_script=2), lossless=True)
    assert O.decode(out).scans() == O.stock_script(3, 1) == O.decode(d).scans()
    assert out[out.index(b"\xff\xdb"):] == d[d.index(b"\xff\xdb"):]
    d1 = open(os.path.join(reference_samples, "level_1_0", "j1.jpg"), "rb").read()
    out1 = O.jpeg_compress(d1, O.params(progressive=1, marker_style=1, scan_script=2), lossless=True)
    stock1 = O.jpeg_compress(d1, O.params(progressive=1, marker_style=1, scan_script=0), lossless=True)
    assert len(out1) < len(stock1)
    assert O.decode(out1).scans() != O.stock_script(3, 0)
    a, b = O.decode(out1), O.decode(d1)
    assert np.array_equal(a.pixels(), b.pixels())


def test_scan_search_grey_and_small_images():
    """the search's grey list (23 candidates) and its early exits on tiny images: the result decodes to the same coefficients"""
    for i, (w, h, ss) in enumerate([(64, 48, 0), (101, 67, 2), (320, 240, 2)]):
        src = synth_jpeg(10 + i, w, h, subsampling=ss, texture=10.0 * i)
        a = O.jpeg_compress(src, O.params(quality=80, scan_script=2))
        b = O.jpeg_compress(src, O.params(quality=80, scan_script=0))
        assert np.array_equal(O.decode(a).pixels(), O.decode(b).pixels())
    from PIL import Image
    im = Image.open(io.BytesIO(synth_jpeg(3, 160, 120))).convert("L")
    bb = io.BytesIO(); im.save(bb, format="JPEG", quality=90)
    g = O.jpeg_compress(bb.getvalue(), O.params(quality=80, scan_script=2))
    assert O.decode(g).im.ncomp == 1 and np.array_equal(O.decode(g).pixels(), O.decode(O.jpeg_compress(bb.getvalue(), O.params(quality=80))).pixels())
