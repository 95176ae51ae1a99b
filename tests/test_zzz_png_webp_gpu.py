"""PNG in, WebP out on the device, through the C ABI and the CLI, against the oracle (file bytes).  After every other device test: this
path went in after the last device run of its round."""
import pytest

from _util import package, product_api
from test_png_webp_emul import check, extra_cases

# a wedged kernel must end the run, not hold the box (these files are last, so ending the process loses nothing after them)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]   # green on the MI355X since round 1 (GPUTEST_r01)


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: libcaesium_hip has no CPU path"
    return a


def test_png_sources_equal_oracle(api):
    from _util import png_cases
    cases = png_cases(small=False)
    assert check(api, cases, 85) >= len(cases) // 2
    assert check(api, extra_cases(), 60) == 4
    assert check(api, cases[:8], 20) >= 4


def test_refusals_mixed_and_damaged(api):
    import test_png_webp_emul as T
    T.test_transparency_becomes_an_alph_chunk(api)
    T.test_resize_in_front(api)
    T.test_mixed_sources_keep_their_order(api)
    T.test_damaged_pngs_convert_like_the_oracle_or_fail(api)


def test_1080p_photograph(api):
    from gen_synth import synth_png
    from oracle import oracle as O
    src = synth_png(70, 1920, 1080, "RGB", texture=4.0, compress_level=1)
    outs = api.batch_convert([src] * 3, package().default_parameters(webp_quality=85), 3)
    want = O.png_to_webp(src, 85)
    assert all(o == want for o in outs)


def test_cli_png_to_webp_on_device(tmp_path):
    import os

    from test_cli import PRODUCT_CLI, png_to_webp_step
    assert os.path.exists(PRODUCT_CLI)
    png_to_webp_step(PRODUCT_CLI, tmp_path)
