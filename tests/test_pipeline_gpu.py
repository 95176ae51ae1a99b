"""Parity tests proper: the HIP path (through the C ABI of libcaesium_hip.so) against the oracle."""
import io
import json
import os

import numpy as np
import pytest

from _util import oracle_lossless, oracle_lossy, product_api
from gen_synth import synth_jpeg, synth_rgb

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def api():
    a = product_api()
    assert a.device_count() >= 1, "no HIP device: the product has no CPU path"
    return a


def params(**kw):
    from _util import package
    return package().default_parameters(**kw)


def test_golden_vectors_through_c_abi(api, monkeypatch):
    """committed libjpeg-turbo vectors: same DQT..EOI as the golden file (marker style differs: merged DQT/DHT).  The vectors pin the SCALAR
    quantiser (libjpeg-turbo has no trellis): CSH_PROFILE=scalar; the library's default profile re-quantises and is checked against the oracle below."""
    from oracle import oracle as O
    monkeypatch.setenv("CSH_PROFILE", "scalar")
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    for case in man["cases"]:
        src = open(os.path.join(GOLD, case["name"] + ".src.jpg"), "rb").read()
        for q in case["qualities"]:
            want = open(os.path.join(GOLD, f"{case['name']}.q{q}.jpg"), "rb").read()
            got = api.compress_in_memory(src, params(jpeg_quality=q))
            assert got == oracle_lossy(src, q), (case["name"], q)
            # the golden file and our output decode to the same coefficients and share every scan's bytes
            a, b = O.decode(want), O.decode(got)
            for c in range(3):
                assert np.array_equal(a.coefs(c), b.coefs(c))
            assert a.encode(O.params(marker_style=0)) == want and b.encode(O.params(marker_style=0)) == want


@pytest.mark.parametrize("w,h,ss,tex", [(128, 96, 2, 45), (101, 67, 2, 0), (97, 61, 2, 80), (64, 48, 0, 30), (33, 31, 2, 60),
                                         (8, 8, 2, 20), (1, 1, 2, 0), (17, 9, 0, 50), (640, 360, 2, 25), (250, 130, 2, 10), (104, 72, 1, 30), (33, 17, 1, 60)])
def test_bytes_equal_oracle(api, w, h, ss, tex):
    src = synth_jpeg(7, w, h, subsampling=ss, texture=tex)
    assert api.compress_in_memory(src, params()) == oracle_lossy(src)


@pytest.mark.parametrize("q", [1, 25, 51, 80, 95, 100])
def test_quality_sweep(api, q):
    src = synth_jpeg(3, 200, 120, texture=35)
    assert api.compress_in_memory(src, params(jpeg_quality=q)) == oracle_lossy(src, q)


def test_stage_taps_equal_oracle(api):
    from oracle import oracle as O
    blobs = [synth_jpeg(5, 128, 96, texture=45), synth_jpeg(6, 97, 61, texture=80), synth_jpeg(3, 64, 48, subsampling=0)]
    b = api.batch(blobs, params())
    b.run()
    outs = b.fetch()
    for i, src in enumerate(blobs):
        ref = oracle_lossy(src)
        oi, oo = O.decode(src), O.decode(ref)
        for c in range(3):
            assert np.array_equal(b.coefs(i, c, 0)[0], oi.coefs_zigzag(c)), ("decode", i, c)
            assert np.array_equal(b.coefs(i, c, 1)[0], oo.coefs_zigzag(c)), ("requant", i, c)
        assert outs[i] == ref


def test_progressive_and_restart_and_gray_inputs(api):
    from PIL import Image
    srcs = [synth_jpeg(2, 104, 72, subsampling=2, progressive=True, texture=20), synth_jpeg(9, 320, 256, restart_rows=1, texture=15),
            synth_jpeg(4, 150, 90, optimize=True, texture=40)]
    g = Image.fromarray(synth_rgb(7, 203, 155, 20)).convert("L")
    b = io.BytesIO(); g.save(b, format="JPEG", quality=90); srcs.append(b.getvalue())
    outs = api.batch_compress(srcs, params())
    for src, out in zip(srcs, outs):
        assert out == oracle_lossy(src)


def test_lossless_transcode(api):
    from oracle import oracle as O
    srcs = [synth_jpeg(21, 333, 222, texture=20), synth_jpeg(2, 104, 72, progressive=True, texture=30), synth_jpeg(8, 64, 64, subsampling=0)]
    outs = api.batch_compress(srcs, params(jpeg_optimize=True))
    for src, out in zip(srcs, outs):
        assert out == oracle_lossless(src)
        a, b = O.decode(src), O.decode(out)
        for c in range(3):
            assert np.array_equal(a.coefs(c), b.coefs(c))


def test_scan_search_conditional_stages(api):
    E.test_emul_scan_search_conditional_stages(api)


def test_rerun_after_a_run_that_took_the_conditional_stages(api):
    E.test_emul_rerun_after_a_run_that_took_the_conditional_stages(api)


def test_refinement_scans_parse_and_apply(api, monkeypatch):
    E.test_emul_refinement_scans_parse_and_apply(api, monkeypatch)


def test_irregular_progressions_decode_in_file_order(api):
    for _ in range(3):   # a race between two first scans over one band shows as a flaky difference
        E.test_emul_irregular_progressions_decode_in_file_order(api)


def test_two_dc_refinement_passes(api):
    for _ in range(3):   # the lost-bit race between two refinement scans of one launch was a flaky difference
        E.test_emul_two_dc_refinement_passes(api)


def test_non_interleaved_sequential_scans(api):
    E.test_emul_non_interleaved_sequential_scans(api)


def test_six_huffman_tables(api):
    E.test_emul_six_huffman_tables(api)


def test_blocks_longer_than_a_subsequence(api):
    E.test_emul_blocks_longer_than_a_subsequence(api)


def test_streams_cut_short(api):
    E.test_emul_streams_cut_short(api)


def test_fuzzed_streams_agree_with_the_oracle(api):
    E.test_emul_fuzzed_streams_agree_with_the_oracle(api)


def test_restart_intervals_decode_in_parallel(api):
    E.test_emul_restart_intervals_decode_in_parallel(api)


def test_correction_bit_overflow_flush(api):
    from test_pipeline_emul import crafted_corrbit_stream
    blob = crafted_corrbit_stream()
    assert api.compress_in_memory(blob, params(jpeg_optimize=True)) == oracle_lossless(blob)
    big = crafted_corrbit_stream(384, 640)   # 1920 luma blocks in one run: the wave-per-run kernel (k_ac_runs_long) cuts it
    assert api.compress_in_memory(big, params(jpeg_optimize=True)) == oracle_lossless(big)


def test_long_eob_runs_and_flat_images(api):
    from PIL import Image
    for im in (np.full((64, 64, 3), 128, np.uint8), np.full((4160, 4096, 3), 77, np.uint8)):
        b = io.BytesIO(); Image.fromarray(im).save(b, format="JPEG", quality=92, subsampling=2)
        src = b.getvalue()
        assert api.compress_in_memory(src, params()) == oracle_lossy(src)


def test_batch_order_and_per_item_errors(api):
    good = [synth_jpeg(i, 80 + 8 * i, 64, texture=10 * i) for i in range(5)]
    blobs = [good[0], b"not an image", good[1], good[2][:150], good[3], b"\x89PNG\r\n\x1a\n" + b"\0" * 32, good[4]]
    outs = api.batch_compress(blobs, params())
    assert [isinstance(o, Exception) for o in outs] == [False, True, False, True, False, True, False]
    assert outs[1].code == 10200 and outs[5].code == 10201
    for src, out in zip([blobs[0], blobs[2], blobs[4], blobs[6]], [outs[0], outs[2], outs[4], outs[6]]):
        assert out == oracle_lossy(src)


def test_1080p_full_size(api):
    srcs = [synth_jpeg(i) for i in range(2)]
    outs = api.batch_compress(srcs, params())
    for src, out in zip(srcs, outs):
        assert out == oracle_lossy(src)


@pytest.mark.parametrize("rel", ["j0.JPG", "level_1_0/j1.jpg"])
def test_reference_fixture_lossless(api, reference_samples, rel):
    """BASELINE configs[0] (samples/j0.JPG --lossless) and the reference's other JPEG sample through the HIP path: both are progressive
    inputs (k_decode_prog), 2000x3000 and 1600x1200-class frames; device == oracle byte for byte (the fixtures travel in tests/golden)."""
    src = open(os.path.join(reference_samples, rel), "rb").read()
    assert api.compress_in_memory(src, params(jpeg_optimize=True)) == oracle_lossless(src)


# ---- the same behavioural cases the emulation suite runs, on the real library ----------------------------------------
import test_pipeline_emul as E


def test_scan_search_reproduces_reference_fixture(api, reference_samples):
    E.test_emul_scan_search_reproduces_reference_fixture(api, reference_samples)


def test_plain_profile_keeps_the_stock_script(api, monkeypatch):
    E.test_emul_plain_profile_keeps_the_stock_script(api, monkeypatch)




def test_sequential_output(api):
    E.test_emul_sequential_output(api)


@pytest.mark.parametrize("ss_in", [0, 1, 2])
@pytest.mark.parametrize("ss_out", [444, 422, 420])
def test_every_chroma_layout_combination(api, ss_in, ss_out):
    E.test_emul_every_chroma_layout_combination(api, ss_in, ss_out)


def test_fused_420_edge_rules(api, monkeypatch):
    E.test_emul_fused_420_edge_rules(api, monkeypatch)


def test_metadata_and_icc_policy(api):
    E.test_emul_metadata_and_icc_policy(api)


def test_compress_to_size_and_convert(api):
    E.test_emul_compress_to_size_and_convert(api)


def test_parallel_decoder_is_the_path_taken(api):
    E.test_emul_parallel_decoder_is_the_path_taken(api)


def test_truncated_stream_stays_on_the_parallel_decoder(api):
    E.test_emul_truncated_stream_stays_on_the_parallel_decoder(api)


def test_batch_size_targeting_and_requant_equivalence(api):
    E.test_emul_batch_size_targeting_and_requant_equivalence(api)


@pytest.mark.parametrize("ss", [0, 1, 2])
def test_resize_lanczos3(api, ss):
    E.test_emul_resize_lanczos3(api, ss)


def test_resize_1080p_to_long_edge_1500(api):
    """BASELINE config 4's geometry: 1920x1080 --long-edge 1500 -> 1500x844"""
    from _util import oracle_resized
    src = synth_jpeg(0)
    out = api.compress_in_memory(src, params(width=1500, jpeg_quality=85))
    assert out == oracle_resized(src, 1500, 0, quality=85)
    from PIL import Image
    assert Image.open(io.BytesIO(out)).size == (1500, 844)
