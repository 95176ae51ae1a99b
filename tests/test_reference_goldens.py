"""Parity tier P1: the oracle -- and on the MI355X the HIP path -- against outputs of the REAL caesiumclt 1.4.0.

The reference holds no golden bytes and its engine cannot be built in the authoring container, so the vectors have to be made
elsewhere: tests/golden/make_reference_goldens.sh is the exact recipe (cargo install caesiumclt --version 1.4.0 + the command lines).
Until its output tree tests/golden/libcaesium/ is committed every case here SKIPS and says so; once it is there each recipe is one
test, named after the SURVEY 8a rows it pins, and a mismatch is a failure that shows where the bytes part."""
import os

import pytest

from _util import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")
TREE = os.path.join(GOLD, "libcaesium")

# recipe directory -> (SURVEY 8a rows it pins, engine profile, how the oracle states the same job)
def _jpeg(quality=80, progressive=1, subsampling=420, keep_metadata=0):
    def f(src):
        from oracle import oracle as O
        return O.jpeg_compress(src, O.params(quality=quality, progressive=progressive, subsampling=subsampling, qtable_profile=3, marker_style=1, scan_script=2,
                                             keep_metadata=keep_metadata, trellis=1, deringing=1))
    return f


def _jpeg_lossless(src):
    from oracle import oracle as O
    return O.jpeg_compress(src, O.params(progressive=1, marker_style=1, scan_script=2), lossless=True)


def _jpeg_resized(width=0, height=0, long_edge=0):
    def f(src):
        from oracle import oracle as O
        w, h = width, height
        if long_edge:
            im = O.decode(src).im
            w, h = (long_edge, 0) if im.width >= im.height else (0, long_edge)
        return O.jpeg_compress_resized(src, O.params(quality=80, progressive=1, subsampling=420, qtable_profile=3, marker_style=1, scan_script=2, trellis=1, deringing=1), w, h)
    return f


def _png(level):
    def f(src):
        from _util import oracle_png
        return oracle_png(src, level)
    return f


def _png_lossy(src):
    from _util import oracle_png_lossy
    return oracle_png_lossy(src, quality=80)


def _jpeg_to_webp(quality, long_edge=0):
    def f(src):
        from _util import oracle_jpeg_to_webp
        from oracle import oracle as O
        w = h = 0
        if long_edge:
            im = O.decode(src).im
            w, h = (long_edge, 0) if im.width >= im.height else (0, long_edge)
        return oracle_jpeg_to_webp(src, quality, w, h)
    return f


RECIPES = {
    "jpeg_q80": ("J1-J9 (mozjpeg profile: table #3, trellis, deringing, scan search)", _jpeg(80)),
    "jpeg_q51": ("J7 trellis at a coarse table", _jpeg(51)),
    "jpeg_q95": ("J7 trellis at a fine table", _jpeg(95)),
    "jpeg_q80_baseline": ("J7/J8 sequential output: the trellis passes' sequential statistics", _jpeg(80, progressive=0)),
    "jpeg_q80_444": ("J5/J7 4:4:4", _jpeg(80, subsampling=444)),
    "jpeg_q80_422": ("J5/J7 4:2:2", _jpeg(80, subsampling=422)),
    "jpeg_lossless": ("J10 coefficient transcode + scan search", _jpeg_lossless),
    "jpeg_q80_exif": ("metadata carry-over (8f-3)", _jpeg(80, keep_metadata=1)),
    "jpeg_q80_width100": ("R1 + S1 resize chain", _jpeg_resized(width=100)),
    "jpeg_q80_long1500": ("R1 + T5 long edge", _jpeg_resized(long_edge=1500)),
    "png_lossless_o0": ("P1-P4 oxipng -o0", _png(0)), "png_lossless_o2": ("P1-P4 oxipng -o2", _png(2)),
    "png_lossless_o3": ("P1-P4 oxipng -o3 (configs[2])", _png(3)), "png_lossless_o6": ("P1-P4 oxipng -o6", _png(6)),
    "png_q80": ("8f-4 lossy PNG (imagequant)", _png_lossy),
    "jpeg_to_webp_q85_long1500": ("S3 + R1 + W1-W3 (configs[3])", _jpeg_to_webp(85, 1500)),
    "jpeg_to_webp_q85": ("S3 + W1-W3", _jpeg_to_webp(85)),
}


def _cases():
    out = []
    for recipe in sorted(RECIPES):
        d = os.path.join(TREE, recipe)
        if not os.path.isdir(d):
            out.append(pytest.param(recipe, None, id=recipe, marks=pytest.mark.skip(reason="tests/golden/libcaesium/ not made yet: run tests/golden/make_reference_goldens.sh where cargo exists (parity tier P1 UNPINNED until then)")))
            continue
        for name in sorted(os.listdir(d)):
            out.append(pytest.param(recipe, name, id=f"{recipe}/{name}"))
    return out


def _source(name):
    for root, _, files in os.walk(GOLD):
        if "libcaesium" in root.split(os.sep):
            continue
        for f in files:
            if f == name or os.path.splitext(f)[0] == os.path.splitext(name)[0]:
                return open(os.path.join(root, f), "rb").read()
    raise FileNotFoundError(name)


def _report(got, want, rows):
    if got == want:
        return
    first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
    pytest.fail(f"P1 parity FAILS for rows {rows}: {len(got)} bytes here, {len(want)} from caesiumclt 1.4.0, first difference at offset {first}")


@pytest.mark.parametrize("recipe,name", _cases())
def test_oracle_reproduces_caesiumclt(recipe, name):
    rows, oracle = RECIPES[recipe]
    want = open(os.path.join(TREE, recipe, name), "rb").read()
    _report(oracle(_source(name)), want, rows)


def test_recipe_is_committed_and_lists_every_row():
    """the recipe itself is part of the repository whether or not it has been run: a shell script with the pinned tool version and one
    command line per recipe this file knows"""
    sh = open(os.path.join(GOLD, "make_reference_goldens.sh")).read()
    assert "cargo install caesiumclt --version 1.4.0" in sh
    for recipe in RECIPES:
        assert f"run {recipe} " in sh or recipe.startswith("png_lossless_o"), recipe
