"""`-q` on a PNG on the device, through the C ABI and the CLI, against the oracle (file bytes)."""
import pytest

from _util import product_api
from test_png_lossy_emul import check_lossy, lossy_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: libcaesium_hip has no CPU path"
    return a


def test_lossy_equals_oracle(api):
    check_lossy(api, lossy_cases(big=True))
    check_lossy(api, lossy_cases()[:4], level=0)
    check_lossy(api, lossy_cases()[:6], level=1, quality=30)   # fewer colours: the cut stops at the bound of that quality (k_png_mediancut)


def test_median_falls_among_equal_keys(api):
    from test_png_lossy_emul import test_median_falls_among_equal_keys as body
    body(api)


def test_max_size_on_png_files(api):
    from test_png_lossy_emul import test_max_size_walks_the_quality
    test_max_size_walks_the_quality(api)


def test_cli_lossy_png_on_device(tmp_path):
    import os

    from test_cli import PRODUCT_CLI, lossy_png_step
    assert os.path.exists(PRODUCT_CLI)
    lossy_png_step(PRODUCT_CLI, tmp_path)
