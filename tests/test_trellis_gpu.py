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


def test_1080p_full_size_and_a_wide_batch(api, monkeypatch):
    """BASELINE configs[1]'s size, and enough files that the AC kernel's workgroups loop over several chunks each"""
    monkeypatch.setenv("CSH_PROFILE", "mozjpeg")
    srcs = [synth_jpeg(i, 1920, 1080) for i in range(2)] + [synth_jpeg(20 + i, 640, 480, texture=3 * i) for i in range(24)]
    outs = api.batch_compress(srcs, E.params())
    for i in (0, 1, 2, 13, 25):
        assert outs[i] == oracle_lossy(srcs[i]), i
    assert all(isinstance(o, bytes) and o[:2] == b"\xff\xd8" for o in outs)
