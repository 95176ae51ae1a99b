"""Lossless PNG row: the kernel sources compiled for the CPU (emulation build, tests only) against the oracle, stage by
stage and file bytes.  The same cases run on the device in test_png_gpu.py."""
import io

import numpy as np
import pytest

from _util import emul_api, oracle_png, package, png_cases
from oracle import oracle as O

PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def api():
    return emul_api()


def check_batch(api, cases, level, keep_metadata=False, stages=True):
    pkg = package()
    p = pkg.default_parameters(png_optimize=True, png_optimization_level=level, keep_metadata=keep_metadata)
    blobs = [c[1] for c in cases]
    b = api.png_batch(blobs, p)
    try:
        b.run()
        outs = b.fetch()
        for i, (name, blob) in enumerate(cases):
            assert not isinstance(outs[i], Exception), (name, outs[i])
            ref, chosen = O.png_optimize(blob, level, keep_metadata)
            if stages:
                P = O.png_decode(blob, keep_metadata)
                P.reduce()   # P2: the device works on the reduced image from here on
                assert np.array_equal(b.rows(i), P.rows()), name
                got, have = b.scores(i)
                want = P.scores()
                for k in range(5):
                    if have >> k & 1:
                        bad = np.argwhere(got[:, :, k] != want[:, :, k])
                        assert not len(bad), (name, "score", k, "row/filter", bad[0].tolist(), int(got[tuple(bad[0])][k]), int(want[tuple(bad[0])][k]))
                trials, winner = b.trials(i)
                greedy = {s: len(O.deflate_zlib(P.filtered(s)[0], 0)) for s, _ in trials}   # the min-cost-path parse only for trials within 9/8 of the smallest greedy stream
                for s, zbytes in trials:
                    f, _ = P.filtered(s)
                    assert np.array_equal(b.stream(i, s), f), (name, s)
                    assert zbytes == (len(O.deflate_zlib(f)) if greedy[s] * 8 <= min(greedy.values()) * 9 else greedy[s]), (name, s)
                if chosen >= 0:
                    assert trials[winner][0] == chosen, name
            assert outs[i] == ref, name
    finally:
        b.close()


def test_indexed_palettes_with_duplicates_and_translucent_entries(api):
    """the index_depth reduction with its round-4 rules (duplicates merged, translucent entries first): the host decides, the device renumbers; file == oracle"""
    from test_oracle_png import palette_with_duplicates
    check_batch(api, palette_with_duplicates(), 2)
    check_batch(api, palette_with_duplicates(), 1, keep_metadata=True)


def test_level3_stage_by_stage(api):
    check_batch(api, png_cases(), 3)


@pytest.mark.parametrize("level", [1, 2, 6])
def test_other_levels(api, level):
    cases = [c for c in png_cases() if c[0] in ("RGB_97x61", "LA_97x61", "RGB_flat_64x48", "I;16_97x61")]
    check_batch(api, cases, level)


def test_keep_metadata(api):
    cases = [c for c in png_cases() if c[0] == "RGB_with_text_and_phys"]
    check_batch(api, cases, 3, keep_metadata=True, stages=False)
    check_batch(api, cases, 3, keep_metadata=False, stages=False)


def test_refusals_and_mixed_batch(api):
    pkg = package()
    good = dict(png_cases())["RGB_97x61"]
    import zlib
    adam7 = good[:33] + (8).to_bytes(4, "big") + b"acTL" + bytes(8) + zlib.crc32(b"acTL" + bytes(8)).to_bytes(4, "big") + good[33:]   # animated: refused
    cut = good[:len(good) // 2]
    # damage inside the zlib stream: flip bits in the middle of the IDAT payload
    i0 = good.index(b"IDAT") + 4
    bad = bytearray(good); bad[i0 + 40] ^= 0xFF; bad[i0 + 41] ^= 0xFF
    short = bytearray(good)   # a valid stream that ends early: re-deflate half of the rows
    blobs = [good, bytes(adam7), cut, bytes(bad), b"\x89PNG\r\n\x1a\n" + b"\0" * 40, good]
    p = pkg.default_parameters(png_optimize=True)
    outs = api.cs_batch_compress(blobs, p)
    assert outs[0] == oracle_png(good) and outs[5] == outs[0]
    assert isinstance(outs[1], Exception) and outs[1].code == 10201
    assert isinstance(outs[2], Exception) and outs[2].code == 30100
    assert isinstance(outs[4], Exception) and outs[4].code == 30100
    # the damaged stream either fails in the oracle too, or both produce the same file
    try:
        ref = oracle_png(bytes(bad))
    except O.PngError:
        ref = None
    if ref is None:
        assert isinstance(outs[3], Exception) and outs[3].code == 30100
    else:
        assert outs[3] == ref


def test_jpeg_and_png_in_one_call(api):
    from _util import oracle_lossless
    from gen_synth import synth_jpeg
    pkg = package()
    png = dict(png_cases())["L_97x61"]
    jpg = synth_jpeg(1, 64, 48)
    p = pkg.default_parameters(png_optimize=True, jpeg_optimize=True)
    outs = api.cs_batch_compress([jpg, png, jpg], p)
    assert outs[1] == oracle_png(png)
    assert outs[0] == oracle_lossless(jpg) and outs[2] == outs[0]
    # without png.optimize a PNG takes the lossy form of the pipeline (tests/test_png_lossy_emul.py)
    from _util import oracle_png_lossy
    outs = api.cs_batch_compress([png], pkg.default_parameters())
    assert outs[0] == oracle_png_lossy(png)


def damaged_pngs(seed, count):
    """valid PNGs with one random injury each: a flipped bit anywhere, a zeroed run, a cut, a chunk length off by a little"""
    rng = np.random.default_rng(seed)
    base = [c[1] for c in png_cases() if c[0] in ("L_level1_input", "RGB_stored_input", "P_97x61", "RGBA_300x2")]
    out = []
    for k in range(count):
        b = bytearray(base[k % len(base)])
        kind = int(rng.integers(0, 4))
        at = int(rng.integers(8, len(b)))
        if kind == 0:
            b[at] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            n = int(rng.integers(1, 9))
            b[at:at + n] = bytes(len(b[at:at + n]))
        elif kind == 2:
            del b[at:]
        else:
            i = bytes(b).find(b"IDAT") - 4
            b[i + 3] = (b[i + 3] + int(rng.integers(1, 5))) & 255
        out.append(bytes(b))
    return out


def agree_with_oracle(api, blobs, level=1):
    pkg = package()
    outs = api.cs_batch_compress(blobs, pkg.default_parameters(png_optimize=True, png_optimization_level=level))
    bad = 0
    for b, o in zip(blobs, outs):
        try:
            ref = O.png_optimize(b, level)[0]
        except O.PngError as e:
            ref = e
        if isinstance(ref, Exception):
            if not (isinstance(o, Exception) and o.code == ref.code):
                bad += 1
        elif o != ref:
            bad += 1
    return bad


def test_damaged_files_are_refused_or_decoded_like_the_oracle(api):
    assert agree_with_oracle(api, damaged_pngs(1, 160)) == 0


def test_emul_reference_sample_pngs(api, reference_samples):
    """samples/p0.png and level_2_0/p2.png through the device pipeline: the oracle's bytes, libpng's pixels"""
    import io, os
    import numpy as np
    from PIL import Image
    from _util import oracle_png, package
    for rel in ("p0.png", "level_1_0/level_2_0/p2.png"):
        data = open(os.path.join(reference_samples, rel), "rb").read()
        out = api.compress_in_memory(data, package().default_parameters(png_optimize=True, png_optimization_level=3))
        assert out == oracle_png(data, 3)
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(out)).convert("RGB")), np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))


def deep_parse_cases():
    """pictures whose chunks take the min-cost-path parse (matches in at least one token of 512): smooth, flat, indexed, one whose last chunk is short"""
    from gen_synth import synth_png
    return [("smooth_rgb_200x150", synth_png(51, 200, 150, "RGB", texture=0.4)), ("smooth_l_300x220", synth_png(52, 300, 220, "L", texture=0.3)),
            ("indexed_160x120", synth_png(53, 160, 120, "P", texture=1.0)), ("smooth_rgba_130x70", synth_png(54, 130, 70, "RGBA", texture=0.5))]


def test_emul_min_cost_path_parse(api):
    """chunks with enough matches: candidates from both hash tables, costs from the previous pass, the per-segment path -- file == oracle"""
    check_batch(api, deep_parse_cases(), 3)
    check_batch(api, deep_parse_cases()[:2], 1, stages=False)


def test_emul_rows_longer_than_65536_bytes(api):
    """a row of more than 65536 bytes: the pair counters of the Bigrams / BigEnt scores no longer fit 16 bits (k_png_scores<true>: the key space in two halves)"""
    from gen_synth import synth_png
    check_batch(api, [("wide_rgb_22000x2", synth_png(61, 22000, 2, "RGB", texture=1.0)), ("narrow_next_to_it", synth_png(62, 40, 3, "RGB"))], 3)


def test_emul_zopfli_means_more_passes(api):
    """png.force_zopfli (--zopfli): the same coder with CSP_DEEP_ITERS_ZOPFLI passes of the cost model; == the oracle's statement of it, never larger than
    a plain run by more than a block header's noise, and the pixels are the input's"""
    import io
    from PIL import Image
    from _util import package
    for name, png in deep_parse_cases()[:3]:
        z = api.compress_in_memory(png, package().default_parameters(png_optimize=True, png_force_zopfli=True))
        plain = api.compress_in_memory(png, package().default_parameters(png_optimize=True))
        assert z == O.png_optimize(png, 3, zopfli=True)[0], name
        assert len(z) <= len(plain) + 16, (name, len(z), len(plain))
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(z)).convert("RGBA")), np.asarray(Image.open(io.BytesIO(png)).convert("RGBA"))), name


def index_depth_cases():
    """8-bit indexed files whose pixels use only the first 16 / 4 / 2 / 1 entries of a long palette, with and without tRNS, one whose pixels reach entry 16,
    one with a chunk that counts on the palette (bKGD, carried under keep_metadata), and an interlaced one"""
    from test_png_webp_emul import make_png
    rng = np.random.default_rng(21)
    plte = bytes(rng.integers(0, 256, 3 * 200, dtype=np.uint8))
    out = []
    for name, top, w, h, extra in (("idx16", 16, 61, 37, []), ("idx4", 4, 50, 20, []), ("idx2", 2, 33, 9, []), ("idx1", 1, 8, 5, []), ("idx17", 17, 40, 30, []),
                                   ("idx4_trns", 4, 45, 31, [(b"tRNS", bytes([0, 128, 255, 7, 9, 200]))]), ("idx16_trns_short", 16, 30, 30, [(b"tRNS", bytes([10, 20]))]),
                                   ("idx4_bkgd", 4, 24, 24, [(b"bKGD", bytes([1]))])):
        px = rng.integers(0, top, size=(h, w), dtype=np.uint8)
        px[h // 2, w // 2] = top - 1
        out.append((name, make_png(w, h, 8, 3, px.tobytes(), extra=[(b"PLTE", plte)] + extra)))
    # entries anywhere in the palette: the used ones are renumbered; a tRNS that only made unused entries transparent goes; a pixel past the palette blocks it
    for name, values, w, h, extra in (("scattered3", [3, 77, 150], 40, 21, []), ("scattered20", list(range(5, 200, 10)), 64, 33, []),
                                      ("scattered_trns", [1, 9, 130], 31, 17, [(b"tRNS", bytes([255, 0] + [255] * 7 + [40]))]),
                                      ("trns_unused_only", [2, 3], 16, 16, [(b"tRNS", bytes([0, 0]))]), ("past_palette", [0, 1, 230], 20, 10, [])):
        px = rng.choice(np.array(values, np.uint8), size=(h, w))
        px.flat[:len(values)] = values
        out.append((name, make_png(w, h, 8, 3, px.tobytes(), extra=[(b"PLTE", plte)] + extra)))
    b = io.BytesIO()
    im = PIL.fromarray(rng.integers(0, 3, size=(19, 27), dtype=np.uint8), "P")
    im.putpalette(list(plte[:3 * 64]))
    im.save(b, format="PNG")
    out.append(("idx3_pillow", b.getvalue()))
    return out


def test_indexed_images_lose_unused_depth(api):
    """an indexed image that does not use its whole palette loses the unused entries and is packed at the depth the rest needs, PLTE / tRNS written again (oracle: index_depth) --
    rows, trials, winner and file equal the oracle's; the result decodes to the source's pixels; a chunk tied to the palette blocks it"""
    cases = index_depth_cases()
    check_batch(api, cases, 2)
    check_batch(api, cases, 1, keep_metadata=True)
    pkg = package()
    outs = api.cs_batch_compress([c[1] for c in cases], pkg.default_parameters(png_optimize=True, png_optimization_level=2))
    depth = {}
    for (name, src), out in zip(cases, outs):
        assert not isinstance(out, Exception), (name, out)
        a, b = PIL.open(io.BytesIO(src)), PIL.open(io.BytesIO(out))
        assert np.array_equal(np.asarray(a.convert("RGBA")), np.asarray(b.convert("RGBA"))), name
        depth[name] = out[24]
    assert depth["idx16"] == 4 and depth["idx4"] == 2 and depth["idx2"] == 1 and depth["idx1"] == 1 and depth["idx17"] == 8 and depth["idx4_trns"] == 2 and depth["idx3_pillow"] in (2, 8), depth
    assert depth["scattered3"] == 2 and depth["scattered20"] == 8 and depth["scattered_trns"] == 2 and depth["trns_unused_only"] == 1 and depth["past_palette"] == 8, depth
    outs = dict(zip([c[0] for c in cases], outs))
    assert b"tRNS" not in outs["trns_unused_only"] and b"tRNS" in outs["scattered_trns"]
    plte_len = lambda f: int.from_bytes(f[f.index(b"PLTE") - 4:f.index(b"PLTE")], "big") // 3
    assert plte_len(outs["scattered20"]) == 20 and plte_len(outs["scattered3"]) == 3 and plte_len(outs["idx17"]) == 17 and plte_len(outs["past_palette"]) == 200
    kept = api.cs_batch_compress([dict(cases)["idx4_bkgd"]], pkg.default_parameters(png_optimize=True, png_optimization_level=2, keep_metadata=True))[0]
    assert kept[24] == 8   # bKGD is an index into the palette as it stands
    lossy = api.cs_batch_compress([dict(cases)["idx16"]], pkg.default_parameters(png_optimize=False, png_optimization_level=2, png_quality=80))[0]
    from _util import oracle_png_lossy
    assert lossy == oracle_png_lossy(dict(cases)["idx16"], 2, quality=80) and lossy[24] == 4

