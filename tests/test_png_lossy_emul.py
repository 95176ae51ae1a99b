"""`-q` on a PNG (png.optimize not set): the reductions, the median-cut quantiser and the same filter trials / coder, emulated
kernels against the oracle; the device run is tests/test_zz_png_lossy_gpu.py.  What the quantiser is and is not: oracle/png_oracle.c."""
import io

import numpy as np
import pytest

from _util import emul_api, oracle_png_lossy, package, png_cases
from gen_synth import synth_png

PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def api():
    return emul_api()


def lossy_cases(big=False):
    names = ("RGB_97x61", "RGBA_97x61", "L_97x61", "P_97x61", "I;16_97x61", "RGB_200x150_3chunks", "reduce_rgba_opaque", "reduce_rgb_grey", "palette_rgb_few", "palette_rgba_translucent",
             "palette_257_colours", "adam7_RGB_33x21", "RGB_1x1", "RGBA_300x2")
    cases = [c for c in png_cases() if c[0] in names]
    cases.append(("RGBA_soft_alpha", synth_png(71, 120, 90, "RGBA", texture=5.0)))
    cases.append(("RGB_tall_24x600", synth_png(73, 24, 600, "RGB", texture=4.0)))   # three bands of k_png_dither's 256 rows, the last one short: the error rows handed down through HBM
    cases.append(("RGB_513_rows", synth_png(74, 9, 513, "RGB", texture=6.0)))         # a band of one row
    if big:
        cases.append(("RGB_640x480", synth_png(72, 640, 480, "RGB", texture=2.0)))
    return cases


def check_lossy(api, cases, level=2, quality=80):
    pkg = package()
    outs = api.cs_batch_compress([c[1] for c in cases], pkg.default_parameters(png_optimize=False, png_optimization_level=level, png_quality=quality))
    floor = 30 if quality >= 80 else 12   # dB: what is asked of the result depends on what was asked of the quantiser
    for (name, src), out in zip(cases, outs):
        assert not isinstance(out, Exception), (name, out)
        assert out == oracle_png_lossy(src, level, quality=quality), name
        a, b = PIL.open(io.BytesIO(src)), PIL.open(io.BytesIO(out))
        assert a.size == b.size
        if a.mode != "I;16":
            x, y = np.asarray(a.convert("RGBA")).astype(np.float64), np.asarray(b.convert("RGBA")).astype(np.float64)
            mse = ((x - y) ** 2).mean()
            assert mse == 0 or 10 * np.log10(255.0 ** 2 / mse) > floor, name   # error diffusion trades a few dB for the absence of bands: still close


def test_lossy_equals_oracle(api):
    check_lossy(api, lossy_cases())


def lattice_png(seed, w, h, alpha=False):
    """pixels on a coarse lattice of colours with a little noise in the low bits: every colour bin of a lattice point has the same mean on every channel,
    so each split of the median cut falls inside a run of bins with EQUAL keys (the order by bin id decides) -- and more than 256 distinct pixels"""
    rng = np.random.default_rng(seed)
    ch = 4 if alpha else 3
    base = rng.choice(np.array([0, 64, 128, 192, 248], np.uint8), size=(h, w, ch))
    if alpha:
        base[:, :, 3] = rng.choice(np.array([0, 128, 255], np.uint8), size=(h, w))
    px = base + rng.integers(0, 4, size=(h, w, ch), dtype=np.uint8) * (base < 250)
    b = io.BytesIO()
    PIL.fromarray(px.astype(np.uint8), "RGBA" if alpha else "RGB").save(b, format="PNG")
    return b.getvalue()


def test_median_falls_among_equal_keys(api):
    """the weighted median inside runs of bins that tie on the cut channel (three counting passes on the device, a sort in the oracle), down to boxes of two
    bins, at a quality that keeps splitting and at one that stops early"""
    pkg = package()
    cases = [("lattice_rgb", lattice_png(5, 96, 64)), ("lattice_rgba", lattice_png(6, 80, 50, alpha=True)), ("lattice_small", lattice_png(7, 24, 16))]
    noise = io.BytesIO()   # white noise, 400 pixels wide: eight rows hold more distinct colour bins than the histogram kernel's LDS table has slots (its direct path)
    PIL.fromarray(np.random.default_rng(8).integers(0, 256, size=(24, 400, 3), dtype=np.uint8), "RGB").save(noise, format="PNG")
    cases.append(("noise", noise.getvalue()))
    for quality in (100, 40):
        outs = api.cs_batch_compress([c[1] for c in cases], pkg.default_parameters(png_optimize=False, png_optimization_level=2, png_quality=quality))
        for (name, src), out in zip(cases, outs):
            assert not isinstance(out, Exception), (name, out)
            assert out == oracle_png_lossy(src, 2, quality=quality), (name, quality)


def test_quality_sets_the_palette_size(api):
    """-q is imagequant's quality: the fewest colours whose error is within the bound of that quality; lower -q, fewer colours, smaller file"""
    cases = [c for c in lossy_cases() if c[0] in ("RGB_200x150_3chunks", "RGBA_soft_alpha", "RGB_97x61")]
    for q in (0, 10, 50, 100):
        check_lossy(api, cases, level=1, quality=q)
    src = dict(cases)["RGB_200x150_3chunks"]
    sizes, colours = [], []
    for q in (0, 10, 40, 70, 90, 100):
        out = api.cs_batch_compress([src], package().default_parameters(png_optimize=False, png_optimization_level=1, png_quality=q))[0]
        sizes.append(len(out)); colours.append(len(PIL.open(io.BytesIO(out)).getcolors(1 << 20)))
    assert colours == sorted(colours) and colours[0] == 2 and colours[-1] > 200 and colours[1] < colours[4]
    assert sizes[0] < sizes[2] < sizes[-1]


def test_max_size_walks_the_quality(api):
    """--max-size on PNG files: libcaesium's bisection over png.quality, every try a run of the lossy pipeline; mixed with JPEGs in one call"""
    from gen_synth import synth_jpeg
    from test_pipeline_emul import reference_size_walk
    pkg = package()
    cases = dict(lossy_cases())
    a, b = cases["RGB_200x150_3chunks"], cases["RGBA_soft_alpha"]

    def enc(src, q):
        return oracle_png_lossy(src, 1, quality=q)
    sizes = {q: len(enc(a, q)) for q in (1, 30, 80)}
    assert sizes[1] < sizes[30] < sizes[80]
    jpg = synth_jpeg(12, 160, 120, texture=20)
    for target in (sizes[30] + 40, sizes[80] * 2, sizes[1] + 10):
        p = pkg.default_parameters(png_optimization_level=1)
        outs = api.batch_compress_to_size([a, jpg, b, b"junk"], p, target)
        assert outs[0] == reference_size_walk(a, target, encode=enc)[1]
        assert outs[2] == reference_size_walk(b, target, encode=enc)[1]
        assert outs[1] == reference_size_walk(jpg, target)[1] and outs[3].code == 10200
    seq, _ = reference_size_walk(a, sizes[30] + 40, encode=enc)
    assert seq[:2] == [80, 40] and len(seq) >= 3
    # unreachable target: the q = 1 file comes back, or the error when the caller does not want it
    assert api.compress_to_size_in_memory(a, pkg.default_parameters(png_optimization_level=1), 50, True) == enc(a, 1)
    with pytest.raises(pkg.CaesiumError) as e:
        api.compress_to_size_in_memory(a, pkg.default_parameters(png_optimization_level=1), 50, False)
    assert e.value.code == 10500
    # png.optimize: one lossless try, whatever the target
    from _util import oracle_png
    assert api.compress_to_size_in_memory(a, pkg.default_parameters(png_optimize=True, png_optimization_level=1), 50, True) == oracle_png(a, 1)


def test_quantised_files_are_indexed(api):
    pkg = package()
    src = dict(lossy_cases())["RGB_200x150_3chunks"]
    out = api.cs_batch_compress([src], pkg.default_parameters(png_optimize=False))[0]
    im = PIL.open(io.BytesIO(out))
    assert im.mode == "P" and len(out) < len(src)
