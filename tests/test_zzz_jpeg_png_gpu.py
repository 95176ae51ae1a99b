"""JPEG in, PNG out on the device, through the C ABI and the CLI, against the oracle (file bytes).  After every other device test: this
path went in after the last device run of its round."""
import pytest

from _util import oracle_jpeg_to_png, package, product_api
from test_jpeg_png_emul import check
from test_webp_emul import webp_cases

# a wedged kernel must end the run, not hold the box (these files are last, so ending the process loses nothing after them)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]   # green on the MI355X since round 1 (GPUTEST_r01)


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: libcaesium_hip has no CPU path"
    return a


def test_targets_equal_oracle(api):
    check(api, webp_cases(), True)
    check(api, webp_cases(), False)
    check(api, webp_cases()[:3], True, level=5)
    check(api, webp_cases()[:2], True, width=60)
    check(api, webp_cases()[1:3], False, height=40)


def test_failures_and_mixed_batches(api):
    import test_jpeg_png_emul as T
    T.test_mixed_batch_and_failures(api)
    T.test_damaged_jpegs_convert_like_the_oracle_or_fail(api)
    T.test_pixels_survive_a_lossless_target(api)


def test_1080p_photograph_with_long_edge(api):
    src = webp_cases(big=True)[-1][1]
    outs = api.batch_convert([src] * 2, package().default_parameters(png_optimize=True, png_optimization_level=1, width=1500), 1)
    want = oracle_jpeg_to_png(src, True, 1, 1500, 0)
    assert all(o == want for o in outs)


def test_cli_jpeg_to_png_on_device(tmp_path):
    import os

    from test_cli import PRODUCT_CLI, jpeg_to_png_step
    assert os.path.exists(PRODUCT_CLI)
    jpeg_to_png_step(PRODUCT_CLI, tmp_path)
