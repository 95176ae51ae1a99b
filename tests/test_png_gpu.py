"""Lossless PNG row on the device, through the C ABI, against the oracle (oracle/png_oracle.c): decoded rows, every
trial's filtered stream and zlib size, the winner, and the output file, byte for byte; at full size through properties
(libpng decodes the output to the input's pixels, zlib accepts the stream, the file is not larger than the input)."""
import io
import zlib

import numpy as np
import pytest

from _util import oracle_png, package, png_cases, product_api
from gen_synth import synth_png
from oracle import oracle as O
from test_png_emul import check_batch

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: libcaesium_hip has no CPU path"
    return a


def test_level3_stage_by_stage(api):
    check_batch(api, png_cases(small=False), 3)


@pytest.mark.parametrize("level", [0, 1, 2, 4, 5, 6])
def test_other_levels(api, level):
    cases = [c for c in png_cases(small=False) if c[0] in ("RGB_97x61", "LA_97x61", "P_97x61", "RGB_flat_64x48", "I;16_97x61", "RGB_200x150_3chunks", "RGBA_511x300")]
    check_batch(api, cases, level)


def test_min_cost_path_parse_zopfli_and_wide_rows(api):
    """round 6's paths on the device: chunks that take the min-cost-path parse (four-wave workgroups, 256-position tiles, the work queue), png.force_zopfli
    (fifteen passes), a row of more than 65536 bytes (k_png_scores<true>), a larger smooth picture (several chunks per trial, live and dead trials)"""
    from test_png_emul import deep_parse_cases
    check_batch(api, deep_parse_cases() + [("smooth_rgb_640x360", synth_png(41, 640, 360, "RGB", texture=0.5))], 3)
    check_batch(api, [("wide_rgb_22000x2", synth_png(61, 22000, 2, "RGB", texture=1.0)), ("narrow_next_to_it", synth_png(62, 40, 3, "RGB"))], 3)
    pkg = package()
    for name, png in deep_parse_cases()[:3]:
        z = api.compress_in_memory(png, pkg.default_parameters(png_optimize=True, png_force_zopfli=True))
        assert z == O.png_optimize(png, 3, zopfli=True)[0], name
    # the same batch again: the queue counters and the scratch areas are the previous run's
    check_batch(api, deep_parse_cases(), 3, stages=False)


def test_keep_metadata(api):
    cases = [c for c in png_cases() if c[0] == "RGB_with_text_and_phys"]
    check_batch(api, cases, 3, keep_metadata=True, stages=False)
    check_batch(api, cases, 3, keep_metadata=False, stages=False)


def test_entry_points_and_refusals(api):
    pkg = package()
    good = dict(png_cases())["RGB_97x61"]
    p = pkg.default_parameters(png_optimize=True)
    assert api.compress_in_memory(good, p) == oracle_png(good)
    adam7 = good[:33] + (8).to_bytes(4, "big") + b"acTL" + bytes(8) + zlib.crc32(b"acTL" + bytes(8)).to_bytes(4, "big") + good[33:]   # animated: refused
    i0 = good.index(b"IDAT") + 4
    outs = api.cs_batch_compress([good, bytes(adam7), good[:len(good) // 2], b"\x89PNG\r\n\x1a\n" + b"\0" * 40, good], p)
    assert outs[0] == oracle_png(good) and outs[4] == outs[0]
    assert [getattr(o, "code", 0) for o in outs] == [0, 10201, 30100, 30100, 0]
    # damage inside the zlib stream, many places: the device and the oracle agree on refusal or on the file
    blobs = []
    rng = np.random.default_rng(5)
    for k in range(48):
        b = bytearray(good)
        at = i0 + 2 + int(rng.integers(0, len(good) - i0 - 20))
        b[at] ^= 1 << int(rng.integers(0, 8))
        blobs.append(bytes(b))
    outs = api.cs_batch_compress(blobs, p)
    for b, o in zip(blobs, outs):
        try:
            ref = oracle_png(b)
        except O.PngError:
            ref = None
        if ref is None:
            assert isinstance(o, Exception) and o.code == 30100
        else:
            assert o == ref


def test_jpeg_and_png_in_one_call(api):
    from _util import oracle_lossless
    from gen_synth import synth_jpeg
    pkg = package()
    png = dict(png_cases())["L_97x61"]
    jpg = synth_jpeg(1, 64, 48)
    outs = api.cs_batch_compress([jpg, png, jpg], pkg.default_parameters(png_optimize=True, jpeg_optimize=True))
    assert outs[1] == oracle_png(png)
    assert outs[0] == oracle_lossless(jpg) and outs[2] == outs[0]
    # (a PNG without png.optimize takes the lossy form of the pipeline: tests/test_zz_png_lossy_gpu.py)


def test_full_size_batch_by_properties(api):
    """configs[2] shape (3840x2160 RGB8, Pillow level 6), -o3: what any correct result must satisfy, the 1080p file of the same batch against
    the oracle byte for byte, and -- the oracle needs a minute per 4K file -- the three 4K files against the oracle's COMMITTED answer
    (tests/golden/oracle_digests_4k.json, made by tests/golden/make_oracle_digests_4k.py): byte parity at configs[2]'s real size."""
    import hashlib
    import json
    import os
    pkg = package()
    blobs = [synth_png(40 + k, 3840, 2160, "RGB", texture=float(k)) for k in range(3)] + [synth_png(50, 1920, 1080, "RGB", texture=2.0)]
    outs = api.cs_batch_compress(blobs, pkg.default_parameters(png_optimize=True, png_optimization_level=3))
    for src, out in zip(blobs, outs):
        assert not isinstance(out, Exception), out
        assert len(out) <= len(src)
        a, b = PIL.open(io.BytesIO(src)), PIL.open(io.BytesIO(out))
        assert a.mode == b.mode and np.array_equal(np.asarray(a), np.asarray(b))
    assert outs[3] == oracle_png(blobs[3])
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_digests_4k.json")))["digests"]
    for k in range(3):
        want = gold[f"png_optimize/synth_png({40 + k},3840,2160,RGB,texture={float(k)})/o3"]
        if hashlib.sha256(blobs[k]).hexdigest() != want["in_sha256"]:
            pytest.skip("this Pillow / zlib writes the synthetic 4K input differently from the one the digests were made with")
        assert len(outs[k]) == want["out_bytes"] and hashlib.sha256(outs[k]).hexdigest() == want["out_sha256"], k


def test_damaged_files_are_refused_or_decoded_like_the_oracle(api):
    from test_png_emul import agree_with_oracle, damaged_pngs
    assert agree_with_oracle(api, damaged_pngs(2, 640)) == 0


def test_reference_sample_pngs(api, reference_samples):
    import test_png_emul as E
    E.test_emul_reference_sample_pngs(api, reference_samples)


def test_indexed_images_lose_unused_depth(api):
    from test_png_emul import test_indexed_images_lose_unused_depth as body
    body(api)


def test_indexed_palettes_with_duplicates_and_translucent_entries(api):
    import test_png_emul as E
    E.test_indexed_palettes_with_duplicates_and_translucent_entries(api)
