"""PNG sources with --width / --height on the device, through the C ABI and the CLI, against the oracle (file bytes).  After every other
device test: this path went in after the last device run of its round."""
import pytest

from _util import oracle_png_resized, package, product_api

# a wedged kernel must end the run, not hold the box (these files are last, so ending the process loses nothing after them)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]   # green on the MI355X since round 1 (GPUTEST_r01)


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: libcaesium_hip has no CPU path"
    return a


def test_resized_pngs_equal_oracle(api):
    import test_png_resize_emul as T
    T.test_every_case_resizes_like_the_oracle_or_is_refused(api)
    T.test_sizes_and_shapes(api)
    T.test_sixteen_bit_images_stay_sixteen_bit(api)
    T.test_sixteen_bit_images_with_a_colour_key(api)
    T.test_result_is_close_to_pillows_lanczos(api)
    T.test_mixed_batch_with_jpegs_and_damage(api)


def test_photograph_to_long_edge(api):
    from gen_synth import synth_png
    src = synth_png(90, 1920, 1080, "RGB", texture=4.0, compress_level=1)
    outs = api.cs_batch_compress([src] * 2, package().default_parameters(png_optimize=True, png_optimization_level=1, width=800))
    want = oracle_png_resized(src, True, 1, 800, 0)
    assert all(o == want for o in outs)


def test_cli_png_resize_on_device(tmp_path):
    import os

    from test_cli import PRODUCT_CLI, png_resize_step
    assert os.path.exists(PRODUCT_CLI)
    png_resize_step(PRODUCT_CLI, tmp_path)
