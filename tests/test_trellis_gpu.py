"""mozjpeg's quantiser half (CSH_PROFILE=mozjpeg: trellis quantisation + overshoot deringing) on the MI355X, through the C ABI, byte for
byte against the oracle; bodies shared with tests/test_trellis_emul.py."""
import pytest

import test_trellis_emul as E
from _util import oracle_lossy, product_api
from gen_synth import synth_jpeg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: the product has no CPU path"
    return a


def test_profiles_equal_oracle(api, monkeypatch):
    E.check_profiles_equal_oracle(api, monkeypatch, E.CASES + [(640, 360, 2, 25)])


def test_quality_sweep(api, monkeypatch):
    E.check_quality_sweep(api, monkeypatch)


def test_deringing_on_clipped_highlights(api, monkeypatch):
    E.check_deringing_on_clipped_highlights(api, monkeypatch)


def test_grey_sequential_resize_and_batches(api, monkeypatch):
    E.check_grey_sequential_resize_and_batches(api, monkeypatch)


def test_size_targeting(api, monkeypatch):
    E.check_size_targeting(api, monkeypatch)


def test_pools_that_overflow_are_grown_and_the_run_repeated(api, monkeypatch):
    E.test_emul_pools_that_overflow_are_grown_and_the_run_repeated(api, monkeypatch)


def test_1080p_full_size_and_a_wide_batch(api, monkeypatch):
    """BASELINE configs[1]'s size, and enough files that the AC kernel's workgroups loop over several chunks each"""
    monkeypatch.setenv("CSH_PROFILE", "mozjpeg")
    srcs = [synth_jpeg(i, 1920, 1080) for i in range(2)] + [synth_jpeg(20 + i, 640, 480, texture=3 * i) for i in range(24)]
    outs = api.batch_compress(srcs, E.params())
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(len(srcs)) as ex:   # every output of the batch, byte for byte (the oracle releases the GIL)
        want = list(ex.map(oracle_lossy, srcs))
    for i, (o, w) in enumerate(zip(outs, want)):
        assert o == w, i


# ---- parity tier P1 on the device: the HIP path against the REAL caesiumclt 1.4.0 (activates when tests/golden/libcaesium/ exists;
# tests/golden/make_reference_goldens.sh makes it; tests/test_reference_goldens.py is the oracle's twin of this test)
import os

import test_reference_goldens as G

DEVICE_RECIPES = {"jpeg_q80": dict(jpeg_quality=80), "jpeg_q51": dict(jpeg_quality=51), "jpeg_q95": dict(jpeg_quality=95),
                  "jpeg_q80_baseline": dict(jpeg_quality=80, jpeg_progressive=False), "jpeg_q80_444": dict(jpeg_quality=80, jpeg_chroma_subsampling=444),
                  "jpeg_q80_422": dict(jpeg_quality=80, jpeg_chroma_subsampling=422), "jpeg_lossless": dict(jpeg_optimize=True),
                  "jpeg_q80_exif": dict(jpeg_quality=80, keep_metadata=True), "jpeg_q80_width100": dict(jpeg_quality=80, width=100)}


@pytest.mark.parametrize("recipe,name", [c for c in G._cases() if c.values[0] in DEVICE_RECIPES])
def test_device_reproduces_caesiumclt(api, monkeypatch, recipe, name):
    monkeypatch.setenv("CSH_PROFILE", "mozjpeg")
    want = open(os.path.join(G.TREE, recipe, name), "rb").read()
    G._report(api.compress_in_memory(G._source(name), E.params(**DEVICE_RECIPES[recipe])), want, G.RECIPES[recipe][0])
