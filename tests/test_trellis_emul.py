"""mozjpeg's quantiser half -- trellis quantisation + overshoot deringing, CSH_PROFILE=mozjpeg -- on the CPU emulation build of the
kernels, byte for byte against the oracle's restatement (oracle/jpeg_oracle.c quantize_trellis_row / cso_dering_block).  The same
bodies run on the MI355X in tests/test_trellis_gpu.py.  Parity with the real crate is UNPINNED (tests/golden/make_reference_goldens.sh)."""
import io

import numpy as np
import pytest

from _util import emul_api, oracle_lossy, oracle_resized, package
from gen_synth import synth_jpeg, synth_rgb


@pytest.fixture(scope="module")
def api():
    return emul_api()


def params(**kw):
    return package().default_parameters(**kw)


def saturated_jpeg(w=160, h=120, seed=5, quality=95, subsampling=0):
    """a picture with clipped highlights: white bars and a white disc on a gradient -- blocks whose samples sit at 255 in runs that
    start the block, end it, or lie inside it (what overshoot deringing rewrites)"""
    from PIL import Image
    img = synth_rgb(seed, w, h, texture=4).astype(np.int32)
    yy, xx = np.mgrid[0:h, 0:w]
    img[(xx // 11) % 3 == 0] = 255
    img[(yy - h // 2) ** 2 + (xx - w // 2) ** 2 < (min(w, h) // 4) ** 2] = 255
    img[:, : w // 5] = np.clip(img[:, : w // 5] + 90, 0, 255)
    b = io.BytesIO()
    Image.fromarray(img.astype(np.uint8), "RGB").save(b, format="JPEG", quality=quality, subsampling=subsampling)
    return b.getvalue()


CASES = [(128, 96, 2, 45), (101, 67, 2, 0), (97, 61, 2, 80), (64, 48, 0, 30), (33, 31, 2, 60), (8, 8, 2, 20), (1, 1, 2, 0), (17, 9, 0, 50),
         (250, 130, 2, 10), (104, 72, 1, 30), (33, 17, 1, 60)]


def check_profiles_equal_oracle(api, monkeypatch, cases=CASES):
    for prof in ("mozjpeg", "mozjpeg-trellis", "mozjpeg-dering"):
        monkeypatch.setenv("CSH_PROFILE", prof)
        for (w, h, ss, tex) in cases:
            src = synth_jpeg(7, w, h, subsampling=ss, texture=tex)
            assert api.compress_in_memory(src, params()) == oracle_lossy(src), (prof, w, h, ss, tex)


def test_emul_profiles_equal_oracle(api, monkeypatch):
    check_profiles_equal_oracle(api, monkeypatch)


def check_quality_sweep(api, monkeypatch):
    """q 1: 16-bit tables, one DC candidate class; q 100: every coefficient is a list entry (the lists spill out of LDS), 9 DC levels"""
    monkeypatch.setenv("CSH_PROFILE", "mozjpeg")
    src = synth_jpeg(3, 120, 88, texture=35)
    for q in (1, 25, 51, 80, 95, 100):
        assert api.compress_in_memory(src, params(jpeg_quality=q)) == oracle_lossy(src, q), q


def test_emul_quality_sweep(api, monkeypatch):
    check_quality_sweep(api, monkeypatch)


def check_deringing_on_clipped_highlights(api, monkeypatch):
    from oracle import oracle as O
    monkeypatch.setenv("CSH_PROFILE", "mozjpeg-dering")
    for ss in (0, 2, 1):
        src = saturated_jpeg(subsampling=ss)
        out = api.compress_in_memory(src, params(jpeg_chroma_subsampling={0: 444, 1: 422, 2: 420}[ss]))
        assert out == oracle_lossy(src, subsampling={0: 444, 1: 422, 2: 420}[ss])
        assert out != O.jpeg_compress(src, O.params(quality=80, subsampling={0: 444, 1: 422, 2: 420}[ss], scan_script=2))   # it does rewrite these blocks
    monkeypatch.setenv("CSH_PROFILE", "mozjpeg")
    src = saturated_jpeg(203, 155, seed=8)
    assert api.compress_in_memory(src, params()) == oracle_lossy(src)


def test_emul_deringing_on_clipped_highlights(api, monkeypatch):
    check_deringing_on_clipped_highlights(api, monkeypatch)


def check_grey_sequential_resize_and_batches(api, monkeypatch):
    from PIL import Image
    monkeypatch.setenv("CSH_PROFILE", "mozjpeg")
    g = Image.fromarray(synth_rgb(7, 203, 155, 20)).convert("L")
    b = io.BytesIO(); g.save(b, format="JPEG", quality=90)
    grey = b.getvalue()
    assert api.compress_in_memory(grey, params()) == oracle_lossy(grey)
    # --jpeg-baseline: the statistics pass is a sequential one-component scan, the DC path is priced with ITS optimal DC table
    for src in (synth_jpeg(4, 150, 90, texture=40), grey, synth_jpeg(6, 64, 48, subsampling=0, texture=10)):
        assert api.compress_in_memory(src, params(jpeg_progressive=False)) == oracle_lossy(src, progressive=0)
    # a batch of different sizes and layouts: the trellis stage's work items, chunks and side arrays are per (image, component)
    srcs = [synth_jpeg(i, 90 + 37 * i, 70 + 11 * i, subsampling=(2, 0, 1)[i % 3], texture=8 * i) for i in range(6)] + [grey, saturated_jpeg()]
    for src, out in zip(srcs, api.batch_compress(srcs, params())):
        assert out == oracle_lossy(src)
    src = synth_jpeg(11, 200, 150, texture=25)
    assert api.compress_in_memory(src, params(width=120)) == oracle_resized(src, 120, 0)
    # a coefficient transcode has no quantiser: --lossless is what it is without the profile
    from _util import oracle_lossless
    assert api.compress_in_memory(src, params(jpeg_optimize=True)) == oracle_lossless(src)


def test_emul_grey_sequential_resize_and_batches(api, monkeypatch):
    check_grey_sequential_resize_and_batches(api, monkeypatch)


def check_size_targeting(api, monkeypatch):
    """--max-size under the profile: the retained DCT is what the trellis quantiser works from, so a try is re-quantise + statistics +
    trellis + coding; every file of a batch ends where libcaesium's walk over full runs ends"""
    from test_pipeline_emul import reference_size_walk
    monkeypatch.setenv("CSH_PROFILE", "mozjpeg")
    srcs = [synth_jpeg(i, 160 + 16 * i, 120, subsampling=(0, 2, 1)[i % 3], texture=10 + 9 * i) for i in range(4)]
    for src, out in zip(srcs, api.batch_compress_to_size(srcs, params(), 4000)):
        assert out == reference_size_walk(src, 4000)[1]
    b = api.batch(srcs, params())
    b.retain_dct()
    b.run()
    b.set_quality([33, 0, 97, 5])
    b.rerun_encode()
    for src, out, q in zip(srcs, b.fetch(), [33, 80, 97, 5]):
        assert out == oracle_lossy(src, q), q


def test_emul_size_targeting(api, monkeypatch):
    check_size_targeting(api, monkeypatch)


def test_trellis_saves_bytes_at_equal_quality():
    """sanity of the restatement itself (it is unpinned): at the same table the trellis output is smaller than the scalar quantiser's and
    its PSNR drop is small -- mozjpeg's published behaviour (several per cent at equal quality); profiles/r03_trellis_gain.txt has the
    equal-PSNR table of the bench set"""
    from PIL import Image
    from oracle import oracle as O
    src = synth_jpeg(2, 320, 240, texture=6)
    ref = np.asarray(Image.open(io.BytesIO(src)).convert("RGB")).astype(np.float64)

    def psnr(blob):
        return 10 * np.log10(255 ** 2 / np.mean((np.asarray(Image.open(io.BytesIO(blob)).convert("RGB")).astype(np.float64) - ref) ** 2))
    plain = O.jpeg_compress(src, O.params(quality=80, scan_script=2))
    moz = O.jpeg_compress(src, O.params(quality=80, scan_script=2, trellis=1, deringing=1))
    assert len(moz) < 0.95 * len(plain)
    assert psnr(plain) - psnr(moz) < 0.6


def test_emul_pools_that_overflow_are_grown_and_the_run_repeated(api, monkeypatch):
    """CSH_TEST_POOL_SHIFT: every token / list pool estimate divided by 16 -- the statistics lists and the coding stages' lists overflow, the run is
    repeated with pools four times the size until they fit (pipeline.cpp csh_batch_run).  The lists that take the trellis's levels (round 5) must
    come out of the retries as they come out of a first run: the same bytes as the oracle's."""
    monkeypatch.setenv("CSH_TEST_POOL_SHIFT", "4")
    srcs = [synth_jpeg(41, 320, 240, texture=40), synth_jpeg(42, 200, 136, subsampling=0, texture=70), synth_jpeg(43, 97, 61, texture=10)]
    for prof in (None, "scalar"):
        if prof: monkeypatch.setenv("CSH_PROFILE", prof)
        else: monkeypatch.delenv("CSH_PROFILE", raising=False)
        outs = api.batch_compress(srcs, params(jpeg_quality=80))
        for s_, o in zip(srcs, outs):
            assert o == oracle_lossy(s_), prof



def test_emul_trellis_levels_through_the_lists_or_through_the_tiles(api, monkeypatch):
    """the coding stages' lists are filtered from the list k_trellis_ac wrote its levels into (the default), or built from the coefficient tiles again
    (CSH_NZ_ONCE=0): the same bytes either way, and the emulation build stops if a block's list room and its programme's entry count ever part"""
    monkeypatch.delenv("CSH_PROFILE", raising=False)
    srcs = [synth_jpeg(51, 200, 136, texture=60), synth_jpeg(52, 97, 61, subsampling=0, texture=25), saturated_jpeg()]
    for q in (80, 30, 97):
        monkeypatch.delenv("CSH_NZ_ONCE", raising=False)
        once = api.batch_compress(srcs, params(jpeg_quality=q))
        monkeypatch.setenv("CSH_NZ_ONCE", "0")
        twice = api.batch_compress(srcs, params(jpeg_quality=q))
        assert once == twice and all(isinstance(o, bytes) for o in once), q
        assert once[0] == oracle_lossy(srcs[0], q)
