"""JPEG in, WebP out on the device, through the C ABI, against the oracle (oracle/webp_oracle.c + the JPEG / resize oracle in
front of it): file bytes; configs[3] shape (1080p JPEG, --long-edge 1500, -q 85) included."""
import io

import numpy as np
import pytest

from _util import oracle_jpeg_to_webp, package, product_api
from test_webp_emul import check, webp_cases

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: libcaesium_hip has no CPU path"
    return a


def test_convert_equals_oracle(api):
    check(api, webp_cases(), 85)
    check(api, webp_cases(), 30)
    check(api, webp_cases()[:4], 100)


def test_token_partitions_as_decision_streams_and_as_chains(api, monkeypatch):
    import test_webp_emul as E
    E.test_token_partitions_as_decision_streams_and_as_chains(api, monkeypatch)


def test_convert_with_resize(api):
    check(api, webp_cases()[:3], 85, width=60)
    check(api, webp_cases()[1:4], 75, height=40)


def test_config4_shape(api):
    """1920x1080 JPEG -> --long-edge 1500 (1500x844) -> WebP q85"""
    cases = webp_cases(big=True)[-1:]
    pkg = package()
    out = api.batch_convert([cases[0][1]] * 3, pkg.default_parameters(webp_quality=85, width=1500), 3)
    want = oracle_jpeg_to_webp(cases[0][1], 85, 1500, 0)
    assert all(o == want for o in out)
    im = PIL.open(io.BytesIO(out[0]))
    assert im.size == (1500, 844)


def test_refusals(api):
    from test_webp_emul import test_entry_point_and_refusals
    test_entry_point_and_refusals(api)
