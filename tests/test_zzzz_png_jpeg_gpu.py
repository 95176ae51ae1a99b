"""PNG in, JPEG out on the device, through the C ABI and the CLI, against the oracle (file bytes).  After every other device test: this
path went in after the last device run of its round."""
import pytest

from _util import oracle_png_to_jpeg, package, product_api

# a wedged kernel must end the run, not hold the box (this file is last, so ending the process loses nothing after it)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]   # green on the MI355X since round 1 (GPUTEST_r01)


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: libcaesium_hip has no CPU path"
    return a


def test_png_sources_equal_oracle(api):
    import test_png_jpeg_emul as T
    T.test_every_png_format_converts_like_the_oracle(api)
    T.test_encoder_parameters_apply(api)
    T.test_resize_in_front(api)
    T.test_pixels_are_the_png(api)
    T.test_mixed_batch_and_failures(api)


def test_1080p_photograph(api):
    from gen_synth import synth_png
    src = synth_png(95, 1920, 1080, "RGB", texture=4.0, compress_level=1)
    outs = api.batch_convert([src] * 3, package().default_parameters(jpeg_quality=80), 0)
    want = oracle_png_to_jpeg(src, 80)
    assert all(o == want for o in outs)
    outs = api.batch_convert([src], package().default_parameters(jpeg_quality=80, width=1280), 0)
    assert outs[0] == oracle_png_to_jpeg(src, 80, 1280, 0)


def test_cli_png_to_jpeg_on_device(tmp_path):
    import os

    from test_cli import PRODUCT_CLI, png_to_jpeg_step
    assert os.path.exists(PRODUCT_CLI)
    png_to_jpeg_step(PRODUCT_CLI, tmp_path)
