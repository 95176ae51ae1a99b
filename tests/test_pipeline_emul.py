"""Kernel LOGIC tests on the CPU emulation build (tests/emul/README.md) + C-ABI surface checks.
The parity tests proper are tests/test_pipeline_gpu.py (-m gpu)."""
import ctypes
import io
import os
import re

import numpy as np
import pytest

from _util import ROOT, emul_api, oracle_lossless, oracle_lossy, oracle_resized, package
from gen_synth import synth_jpeg, synth_rgb


@pytest.fixture(scope="module")
def api():
    return emul_api()


def params(**kw):
    return package().default_parameters(**kw)


@pytest.mark.parametrize("w,h,ss,tex", [(128, 96, 2, 45), (101, 67, 2, 0), (97, 61, 2, 80), (64, 48, 0, 30), (33, 31, 2, 60),
                                         (8, 8, 2, 20), (1, 1, 2, 0), (17, 9, 0, 50), (250, 130, 2, 10), (16, 16, 0, 90), (104, 72, 1, 30), (33, 17, 1, 60), (3, 5, 1, 10)])
def test_emul_bytes_equal_oracle(api, w, h, ss, tex):
    src = synth_jpeg(7, w, h, subsampling=ss, texture=tex)
    assert api.compress_in_memory(src, params()) == oracle_lossy(src)


def test_emul_scan_search_conditional_stages(api):
    """mozjpeg's search looks at luma Al 3 only when Al 2 beat Al 1, and at the splits at 12 / 18 only while the search runs on: those
    candidates are coded in stages of their own, for the images that ask for them only.  A batch that mixes such images with ones whose
    search stops early takes all three extra stages and every file still equals the oracle's; a batch of calm pictures takes none."""
    from oracle import oracle as O
    rich = [(0, 90, 80), (2, 40, 95), (0, 10, 80), (4, 90, 80), (1, 90, 80)]    # (seed, texture, quality): luma Al 2, splits at 12 and at 18 (luma and chroma)
    calm = [(0, 0, 30), (3, 0, 30)]
    for q in (80, 95, 30):
        srcs = [synth_jpeg(sd, 160, 120, texture=tx) for sd, tx, qq in rich + calm if qq == q]
        if not srcs:
            continue
        b = api.batch(srcs, params(jpeg_quality=q))
        t = b.run()
        outs = b.fetch()
        for src, out in zip(srcs, outs):
            assert out == oracle_lossy(src, q)
        if q == 80:
            assert t.n_search_extra == 3
            scripts = [O.decode(o).scans() for o in outs]
            assert any(s[1][4] == 2 for s in scripts)                                      # a luma band scan at Al 2: Al 3 was tried
            assert any((0,) == s[1][0] and s[1][2] in (12, 18) for s in scripts)           # a late split won
        if q == 30:
            assert t.n_search_extra == 0


@pytest.mark.parametrize("q", [1, 25, 51, 95, 100])
def test_emul_quality_sweep(api, q):
    src = synth_jpeg(3, 120, 88, texture=35)
    assert api.compress_in_memory(src, params(jpeg_quality=q)) == oracle_lossy(src, q)


def test_emul_inputs_progressive_restart_gray_optimised(api):
    from PIL import Image
    srcs = [synth_jpeg(2, 104, 72, subsampling=2, progressive=True, texture=20), synth_jpeg(9, 160, 128, restart_rows=1, texture=15),
            synth_jpeg(4, 150, 90, optimize=True, texture=40)]
    g = Image.fromarray(synth_rgb(7, 203, 155, 20)).convert("L")
    b = io.BytesIO(); g.save(b, format="JPEG", quality=90); srcs.append(b.getvalue())
    for src, out in zip(srcs, api.batch_compress(srcs, params())):
        assert out == oracle_lossy(src)


def test_emul_parallel_decoder_is_the_path_taken(api):
    """baseline (sequential, no DRI) inputs go through the self-synchronising decoder without falling back"""
    blobs = [synth_jpeg(i, 400, 300, texture=15 * i) for i in range(4)] + [synth_jpeg(9, 320, 240, optimize=True, texture=60)]
    blobs += [synth_jpeg(2, 104, 72, progressive=True), synth_jpeg(9, 160, 128, restart_rows=1)]
    b = api.batch(blobs, params())
    t = b.run()
    assert t.n_images == 7 and t.n_seq_decoded == 0 and t.n_prog_decoded == 1 and t.n_par_fallback == 0   # restart intervals -> parallel decoder too, progressive -> wave-per-chain kernel
    for src, out in zip(blobs, b.fetch()):
        assert out == oracle_lossy(src)


def deep_refinement_file(seed=5, w=203, h=155, ss=2, texture=40):
    """a progressive file whose script refines deeper and in more pieces than any encoder's stock script: three refinement passes over luma, chroma
    bands refined apart from each other and in another order than they were first coded"""
    from oracle import oracle as O
    ci = O.decode(synth_jpeg(seed, w, h, subsampling=ss, texture=texture))
    script = [((0, 1, 2), 0, 0, 0, 1),
              ((0,), 1, 63, 0, 3), ((1,), 1, 5, 0, 1), ((1,), 6, 63, 0, 1), ((2,), 1, 63, 0, 2),
              ((0,), 1, 63, 3, 2), ((2,), 21, 63, 2, 1), ((0,), 1, 63, 2, 1), ((2,), 1, 20, 2, 1), ((0, 1, 2), 0, 0, 1, 0),
              ((1,), 6, 63, 1, 0), ((0,), 1, 63, 1, 0), ((1,), 1, 5, 1, 0), ((2,), 1, 63, 1, 0)]
    return ci.encode(O.params(progressive=1, marker_style=0), script=script)


def test_emul_refinement_scans_parse_and_apply(api, monkeypatch):
    """AC refinement scans: the serial parse finds every block's bit position, the apply is one lane per block (k_decode_refine.hip).  Deep scripts, EOB
    runs across many blocks (a flat picture), long correction-bit stretches, streams cut inside refinement scans; the same bytes from the one-wave-per-
    chain kernel (CSH_PROG_PAR=1) and from everything on chains (CSH_PROG_PAR=0)"""
    from PIL import Image
    srcs = [deep_refinement_file(), deep_refinement_file(7, 64, 48, 0, 10), deep_refinement_file(8, 129, 67, 1, 80), crafted_corrbit_stream(),
            synth_jpeg(2, 520, 390, progressive=True, texture=3)]
    b = io.BytesIO(); Image.fromarray(np.full((256, 320, 3), 99, np.uint8)).save(b, format="JPEG", quality=90, progressive=True); srcs.append(b.getvalue())
    deep = srcs[0]
    first_refine = [i for i in range(len(deep) - 1) if deep[i] == 0xFF and deep[i + 1] == 0xDA][5]
    for frac in (0.05, 0.4, 0.8, 0.99):
        n = first_refine + int((len(deep) - first_refine) * frac)
        srcs += [deep[:n] + b"\xff\xd9", deep[:n + 1]]
    want = [oracle_lossless(s) for s in srcs]
    for mode in (None, "1", "0"):
        if mode is None: monkeypatch.delenv("CSH_PROG_PAR", raising=False)
        else: monkeypatch.setenv("CSH_PROG_PAR", mode)
        bt = api.batch(srcs, params(jpeg_optimize=True))
        t = bt.run()
        assert t.n_seq_decoded == 0 and t.n_par_fallback == 0
        assert (t.n_refine_chains >= 18) if mode is None else (t.n_refine_chains == 0), mode   # files cut short lose chains
        for i, (w_, out) in enumerate(zip(want, bt.fetch())):
            assert out == w_, (mode, i)
    monkeypatch.delenv("CSH_PROG_PAR", raising=False)
    assert api.batch_compress(srcs[:4], params()) == [oracle_lossy(s) for s in srcs[:4]]


def two_dc_refinements_file(seed=6, w=211, h=157, ss=2, texture=30):
    """a regular progression whose DC starts at Al = 2 and is refined twice (to Al = 1, then to Al = 0): both DC refinement scans of the image sit in
    one launch of k_dc_refine and touch the same int16 (the OR into it is atomic)"""
    from oracle import oracle as O
    ci = O.decode(synth_jpeg(seed, w, h, subsampling=ss, texture=texture))
    script = [((0, 1, 2), 0, 0, 0, 2), ((0,), 1, 63, 0, 1), ((1,), 1, 63, 0, 1), ((2,), 1, 63, 0, 1),
              ((0, 1, 2), 0, 0, 2, 1), ((0,), 1, 63, 1, 0), ((0, 1, 2), 0, 0, 1, 0), ((1,), 1, 63, 1, 0), ((2,), 1, 63, 1, 0)]
    return ci.encode(O.params(progressive=1, marker_style=0), script=script)


def test_emul_two_dc_refinement_passes(api):
    srcs = [two_dc_refinements_file(), two_dc_refinements_file(11, 640, 480, 2, 60), two_dc_refinements_file(12, 97, 75, 0, 15)]
    want = [oracle_lossless(s) for s in srcs]
    bt = api.batch(srcs, params(jpeg_optimize=True))
    t = bt.run()
    assert t.n_seq_decoded == 0 and t.n_par_fallback == 0
    assert bt.fetch() == want
    assert api.batch_compress(srcs, params()) == [oracle_lossy(s) for s in srcs]


def test_emul_irregular_progressions_decode_in_file_order(api):
    """a damaged scan header can make two first scans cover one band, or a refinement scan come before the band's first scan: libjpeg warns and decodes in file
    order (the later scan wins).  The parallel kinds of the progressive decoder run side by side, so such a file must stay on the ordered chains"""
    src = bytearray(deep_refinement_file())
    sos = [i for i in range(len(src) - 1) if src[i] == 0xFF and src[i + 1] == 0xDA]
    def patched(scan, Ss=None, Se=None, AhAl=None):
        b = bytearray(src)
        at = sos[scan] + 4 + 1 + 2 * b[sos[scan] + 4]      # behind the component list: Ss, Se, Ah/Al
        if Ss is not None: b[at] = Ss
        if Se is not None: b[at + 1] = Se
        if AhAl is not None: b[at + 2] = AhAl
        return bytes(b)
    files = [patched(3, Ss=1),            # chroma 6-63 first scan now covers 1-63: two first scans over 1-5
             patched(5, AhAl=0x00),       # luma's first refinement claims to be a first scan
             patched(1, AhAl=0x32),       # luma's first scan claims to be a refinement (nothing to refine yet)
             patched(7, AhAl=0x10)]       # a refinement that skips a level
    bt = api.batch(files, params(jpeg_optimize=True))
    t = bt.run()
    outs = bt.fetch()
    assert t.n_refine_chains == 0
    # a regular progression whose DATA leaves the band: luma's first refinement scan cut to 1-40 (its runs were coded for 1-63: one of them ends behind 40
    # and libjpeg stores the coefficient there, in the band no scan of this file refines any more), and a first scan whose band was narrowed
    from oracle import oracle as O
    files += [patched(11, Se=40)]   # the last luma refinement: nothing behind it makes the progression irregular
    plain = bytearray(O.decode(synth_jpeg(5, 120, 88, texture=40)).encode(O.params(progressive=1, marker_style=0), script=[((0, 1, 2), 0, 0, 0, 0), ((0,), 1, 63, 0, 0), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0)]))
    at = [i for i in range(len(plain) - 1) if plain[i] == 0xFF and plain[i + 1] == 0xDA][1]
    plain[at + 4 + 1 + 2 * plain[at + 4] + 1] = 20   # luma's only AC scan narrowed to 1-20: its runs land behind 20
    files += [bytes(plain)]
    bt2 = api.batch(files[4:], params(jpeg_optimize=True))
    t2 = bt2.run()
    outs += bt2.fetch()
    assert t2.n_seq_decoded == 2   # both handed to the kernel that decodes in file order
    for i, (f, out) in enumerate(zip(files, outs)):
        try:
            want = oracle_lossless(f)
        except Exception:
            want = None
        if want is None:
            assert isinstance(out, Exception), i
        else:
            assert out == want, i


def test_emul_non_interleaved_sequential_scans(api):
    """a sequential-mode file whose components come in three separate scans (legal, rare): each scan is its own segment of the
    parallel decoder, the block grid of a non-interleaved scan is the component's real one (no MCU padding blocks)"""
    from oracle import oracle as O
    for (w, h, ss) in [(203, 155, 2), (64, 48, 0), (99, 73, 1)]:
        ci = O.decode(synth_jpeg(5, w, h, subsampling=ss, texture=30))
        src = ci.encode(O.params(progressive=0, marker_style=0), script=[((0,), 0, 63, 0, 0), ((1,), 0, 63, 0, 0), ((2,), 0, 63, 0, 0)])
        b = api.batch([src], params())
        t = b.run()
        assert t.n_seq_decoded == 0 and t.n_par_fallback == 0
        assert b.fetch()[0] == oracle_lossy(src)
        assert api.compress_in_memory(src, params(jpeg_optimize=True)) == oracle_lossless(src)


def six_tables(src):
    """give Cr its own (identical) Huffman tables, ids 2/2: the file then defines six tables, more than the compact table-set
    form of the parallel decoder holds, so the whole batch runs the 8-slot variant of its kernels"""
    out = bytearray(src[:2]); i = 2; dht = {}
    while i < len(src):
        m, L = src[i + 1], int.from_bytes(src[i + 2:i + 4], "big")
        seg = src[i:i + 2 + L]
        if m == 0xC4:
            p = 4
            while p < len(seg):
                n = sum(seg[p + 1:p + 17]); dht[seg[p]] = bytes(seg[p + 1:p + 17 + n]); p += 17 + n
        if m == 0xDA:
            for cls in (0, 1):
                body = bytes([cls << 4 | 2]) + dht[cls << 4 | 1]
                out += b"\xff\xc4" + (2 + len(body)).to_bytes(2, "big") + body
            sos = bytearray(seg)
            sos[5 + 2 * (sos[4] - 1) + 1] = 0x22
            return bytes(out + sos + src[i + 2 + L:])
        out += seg; i += 2 + L
    raise AssertionError("no SOS")


def test_emul_six_huffman_tables(api):
    srcs = [six_tables(synth_jpeg(3, 203, 155, texture=30)), synth_jpeg(5, 160, 120, texture=20), six_tables(synth_jpeg(8, 333, 222, restart_rows=1, texture=25))]
    assert oracle_lossy(srcs[0]) == oracle_lossy(synth_jpeg(3, 203, 155, texture=30))
    b = api.batch(srcs, params())
    t = b.run()
    assert t.n_seq_decoded == 0 and t.n_par_fallback == 0
    for src, out in zip(srcs, b.fetch()):
        assert out == oracle_lossy(src)


def long_block_jpeg(w=48, h=32, seed=3):
    """grayscale baseline JPEG, standard Huffman tables, quantiser 1, every AC coefficient +-512..1023: each block codes to
    ~205 bytes, i.e. spans two or three 128-byte sub-sequences of the parallel decoder"""
    from PIL import Image
    b = io.BytesIO(); Image.new("L", (w, h), 128).save(b, format="JPEG", quality=100, optimize=False); ref = b.getvalue()
    segs = {}; i = 2
    while ref[i + 1] != 0xDA:
        L = int.from_bytes(ref[i + 2:i + 4], "big"); segs.setdefault(ref[i + 1], []).append(ref[i:i + 2 + L]); i += 2 + L
    sos = ref[i:i + 2 + int.from_bytes(ref[i + 2:i + 4], "big")]
    tabs = {}
    for seg in segs[0xC4]:
        p = 4
        while p < len(seg):
            bits = seg[p + 1:p + 17]; n = sum(bits); vals = seg[p + 17:p + 17 + n]
            code, k, enc = 0, 0, {}
            for l in range(1, 17):
                for _ in range(bits[l - 1]): enc[vals[k]] = (code, l); code += 1; k += 1
                code <<= 1
            tabs[seg[p]] = enc; p += 17 + n
    rng = np.random.default_rng(seed)
    out = []; acc = 0; nb = 0
    def put(v, n):
        nonlocal acc, nb
        acc = (acc << n) | (v & ((1 << n) - 1)); nb += n
        while nb >= 8:
            byte = (acc >> (nb - 8)) & 255; out.append(byte)
            if byte == 255: out.append(0)
            nb -= 8
    def coef(table, run, v):
        a = abs(v); s = a.bit_length()
        c, l = table[(run << 4) | s]; put(c, l)
        if s: put(v if v > 0 else v - 1, s)
    pred = 0
    for _ in range(((w + 7) // 8) * ((h + 7) // 8)):
        dc = int(rng.integers(-200, 200)); coef(tabs[0x00], 0, dc - pred); pred = dc
        for k in range(1, 64): coef(tabs[0x10], 0, int(rng.integers(512, 1024)) * int(rng.choice([-1, 1])))
    if nb: put((1 << (8 - nb)) - 1, 8 - nb)
    dqt = b"\xff\xdb\x00\x43\x00" + bytes([1] * 64)
    head = ref[:2] + b"".join(s for m in (0xE0,) for s in segs.get(m, [])) + dqt + segs[0xC0][0] + b"".join(segs[0xC4])
    return head + sos + bytes(out) + b"\xff\xd9"


def test_emul_blocks_longer_than_a_subsequence(api):
    """a block of ~205 coded bytes is decoded by two or three lanes: the first stores the octets it completes, the others
    what follows, and the octet a cut falls into is written coefficient by coefficient from both sides"""
    srcs = [long_block_jpeg(16, 8, 4), long_block_jpeg(32, 16, 6), long_block_jpeg(48, 32, 3), long_block_jpeg(200, 120, 5)]
    for lossless in (True, False):
        b = api.batch(srcs, params(jpeg_optimize=lossless))
        t = b.run()
        # such a stream hardly self-synchronises (no block boundary inside most sub-sequences), so on a GPU the states settle
        # about one cut per round: the two small files are through well inside the 40 rounds, the large one may be handed to
        # the sequential kernel -- either way the bytes are the oracle's
        assert t.n_seq_decoded <= 2
        for src, out in zip(srcs, b.fetch()):
            assert out == (oracle_lossless(src) if lossless else oracle_lossy(src))


def restart_cases():
    """restart-interval sources: intervals of rows and of odd block counts, every layout, grayscale, more than 8 intervals
    (the RSTm index wraps), and two broken ones (a marker out of order, a marker missing)"""
    from PIL import Image
    out = []
    for i, (w, h, ss, kw) in enumerate([(160, 128, 2, {"restart_marker_rows": 1}), (200, 168, 2, {"restart_marker_blocks": 7}), (99, 73, 0, {"restart_marker_blocks": 3}),
                                         (104, 72, 1, {"restart_marker_rows": 2}), (333, 222, 2, {"restart_marker_blocks": 1}), (64, 64, 2, {"restart_marker_blocks": 1000})]):
        b = io.BytesIO(); Image.fromarray(synth_rgb(20 + i, w, h, 25)).save(b, format="JPEG", quality=88, subsampling=ss, **kw)
        out.append(b.getvalue())
    g = Image.fromarray(synth_rgb(7, 203, 155, 20)).convert("L")
    b = io.BytesIO(); g.save(b, format="JPEG", quality=90, restart_marker_blocks=5); out.append(b.getvalue())
    good = out[1]
    i0 = good.index(b"\xff\xd1")
    out.append(good[:i0] + b"\xff\xd3" + good[i0 + 2:])            # RST1 replaced by RST3
    i1 = good.index(b"\xff\xd2")
    out.append(good[:i1] + good[i1 + 2:])                            # RST2 missing
    return out


def test_emul_restart_intervals_decode_in_parallel(api):
    blobs = restart_cases()
    b = api.batch(blobs, params())
    t = b.run()
    assert t.n_seq_decoded == 2 and t.n_par_fallback == 0      # only the two broken files leave the parallel decoder
    for i, (src, out) in enumerate(zip(blobs, b.fetch())):
        assert out == oracle_lossy(src), i
    for src, out in zip(blobs[:4], api.batch_compress(blobs[:4], params(jpeg_optimize=True))):
        assert out == oracle_lossless(src)


def test_emul_relaxation_is_order_independent(api):
    """the emulation normally runs lanes in ascending order, which lets the in-place relaxation converge in its first
    sweep; running every launch in DESCENDING order forces the work-list rounds a real GPU needs (and every atomics-based
    kernel to cope with another arrival order).  Bytes must not change."""
    import ctypes
    blobs = [synth_jpeg(1, 640, 360, texture=25), synth_jpeg(4, 333, 222, subsampling=0, texture=50), synth_jpeg(2, 104, 72, progressive=True), deep_refinement_file()]
    api.L.csh_emul_set_reverse.argtypes = [ctypes.c_int]
    api.L.csh_emul_set_reverse(1)
    try:
        b = api.batch(blobs, params())
        t = b.run()
        outs = b.fetch()
    finally:
        api.L.csh_emul_set_reverse(0)
    assert t.n_par_fallback == 0 and t.n_seq_decoded == 0 and t.n_prog_decoded == 2 and t.n_refine_chains == 6   # the refinement scans' waves take tickets: a scan's predecessor has started
    for src, out in zip(blobs, outs):
        assert out == oracle_lossy(src)


def test_emul_worst_case_concurrency_and_shared_tables(api):
    """GPU lanes of one list round may all read the states from BEFORE the round (pure Jacobi).  With similar (optimised)
    or identical luma/chroma Huffman tables a wrong block-in-MCU label then creeps one sub-sequence per round; the
    prefix-sum re-labelling must settle it.  The emulation's jacobi mode reproduces exactly that schedule."""
    import ctypes
    from PIL import Image
    from oracle import oracle as O
    # one Huffman table set shared by every component: craft with the oracle (sequential, standard luma tables for all)
    blobs = [synth_jpeg(9, 320, 240, optimize=True, texture=60), synth_jpeg(3, 640, 480, optimize=True, texture=25)]
    api.L.csh_emul_set_jacobi.argtypes = [ctypes.c_int]
    api.L.csh_emul_set_jacobi(1)
    try:
        b = api.batch(blobs, params())
        t = b.run()
        outs = b.fetch()
    finally:
        api.L.csh_emul_set_jacobi(0)
    assert t.n_par_fallback == 0, "parallel decoder fell back"
    for src, out in zip(blobs, outs):
        assert out == oracle_lossy(src)


def test_emul_streams_cut_short(api):
    """files cut at many points, sequential / restart-interval / progressive: whatever path takes them (parallel decoder with
    hand-over, wave-per-chain, sequential kernel) the bytes are the oracle's, which follows libjpeg's insufficient-data rule"""
    blobs = []
    for kw in ({}, {"restart_rows": 1}, {"progressive": True}, {"subsampling": 0, "optimize": True}):
        src = synth_jpeg(6, 160, 120, texture=30, **kw)
        start = src.index(b"\xff\xda") + 14
        for frac in (0.02, 0.31, 0.5, 0.77, 0.97):
            n = start + int((len(src) - start) * frac)
            blobs += [src[:n] + b"\xff\xd9", src[:n + 1]]
    for lossless in (False, True):
        outs = api.batch_compress(blobs, params(jpeg_optimize=lossless))
        for i, (src, out) in enumerate(zip(blobs, outs)):
            assert not isinstance(out, Exception), (i, out)
            assert out == (oracle_lossless(src) if lossless else oracle_lossy(src)), i


def test_emul_truncated_stream_stays_on_the_parallel_decoder(api):
    """a sequential-mode file cut short is decoded in parallel all the same: the lane in which the data runs out finishes that
    MCU on zero bits, later MCUs stay zero (DC included) -- libjpeg's insufficient-data rule, no hand-over to the slow kernel"""
    src = synth_jpeg(3, 200, 150, texture=30)
    cut = src[:len(src) * 2 // 3] + b"\xff\xd9"
    b = api.batch([cut], params())
    t = b.run()
    assert t.n_par_fallback == 0 and t.n_seq_decoded == 0
    assert b.fetch()[0] == oracle_lossy(cut)


@pytest.mark.parametrize("ss_in", [0, 1, 2])
@pytest.mark.parametrize("ss_out", [444, 422, 420])
def test_emul_every_chroma_layout_combination(api, ss_in, ss_out):
    """--jpeg-chroma-subsampling 4:4:4 / 4:2:2 / 4:2:0 from 4:4:4 / 4:2:2 / 4:2:0 sources (odd sizes: every edge rule)"""
    for (w, h) in [(99, 73), (64, 48), (17, 9), (2, 3)]:
        src = synth_jpeg(6, w, h, subsampling=ss_in, texture=35)
        assert api.compress_in_memory(src, params(jpeg_chroma_subsampling=ss_out)) == oracle_lossy(src, subsampling=ss_out), (w, h)


def test_emul_fused_420_edge_rules(api, monkeypatch):
    """k_resample_fdct_420 (4:2:0 kept, no resize): every combination of odd / even width and height around block and MCU
    boundaries, checked against the oracle and against the three-kernel chain it replaces"""
    sizes = [(w, h) for w in (5, 6, 15, 16, 17, 18, 31, 32, 33, 34, 47, 49) for h in (5, 6, 15, 16, 17, 18, 33, 34)]
    srcs = [synth_jpeg(11 + i, w, h, subsampling=2, texture=40 + (i % 5) * 10) for i, (w, h) in enumerate(sizes)]
    want = [oracle_lossy(s) for s in srcs]
    b = api.batch(srcs, params())
    b.run()
    got = b.fetch()
    bad = [sizes[i] for i in range(len(sizes)) if got[i] != want[i]]
    assert not bad, bad
    monkeypatch.setenv("CSH_NO_FUSED_420", "1")
    b = api.batch(srcs, params())
    b.run()
    assert b.fetch() == want


def test_emul_metadata_and_icc_policy(api):
    """-e / --strip-icc: EXIF+COM copied only with keep_metadata, ICC (APP2 ICC_PROFILE) follows jpeg_preserve_icc"""
    from PIL import Image
    im = Image.fromarray(synth_rgb(3, 96, 64, 20))
    exif = Image.Exif(); exif[0x010F] = "caesium-hip test"; exif[0x0112] = 6
    b = io.BytesIO(); im.save(b, format="JPEG", quality=90, exif=exif.tobytes(), icc_profile=b"fake-icc-profile-bytes" * 40, comment=b"hello")
    src = b.getvalue()
    assert b"ICC_PROFILE" in src and b"Exif" in src
    for keep in (0, 1):
        for icc in (0, 1):
            out = api.compress_in_memory(src, params(keep_metadata=bool(keep), jpeg_preserve_icc=bool(icc)))
            assert out == oracle_lossy(src, keep_metadata=keep, preserve_icc=icc)
            assert (b"ICC_PROFILE" in out) == bool(icc) and (b"Exif" in out) == bool(keep) and (b"hello" in out) == bool(keep)
            out = api.compress_in_memory(src, params(keep_metadata=bool(keep), jpeg_preserve_icc=bool(icc), jpeg_optimize=True))
            assert out == oracle_lossless(src, keep_metadata=keep, preserve_icc=icc)


@pytest.mark.parametrize("ss", [0, 1, 2])
def test_emul_resize_lanczos3(api, ss):
    """--width/--height/--long-edge: decode -> RGB (jdcolor) -> image-rs Lanczos3 (f32, vertical then horizontal) -> YCbCr
    (jccolor) -> encode.  Bit-exact against the oracle's restatement (0 ULP, so the 1-ULP bar of the north star holds)."""
    src = synth_jpeg(4, 200, 140, subsampling=ss, texture=30)
    for (w, h) in [(150, 0), (0, 35), (97, 201), (200, 140), (333, 0), (1, 1), (0, 1000)]:
        out = api.compress_in_memory(src, params(width=w, height=h))
        assert out == oracle_resized(src, w, h), (w, h)
    from PIL import Image
    g = Image.fromarray(synth_rgb(7, 83, 55, 20)).convert("L")
    b = io.BytesIO(); g.save(b, format="JPEG", quality=90)
    assert api.compress_in_memory(b.getvalue(), params(width=40, jpeg_chroma_subsampling=444)) == oracle_resized(b.getvalue(), 40, 0, subsampling=444)
    # a batch mixing resized geometry classes keeps order; lossless + resize is refused per item
    outs = api.batch_compress([src, src[:100], src], params(height=70))
    assert outs[0] == outs[2] == oracle_resized(src, 0, 70) and isinstance(outs[1], Exception)
    with pytest.raises(package().CaesiumError) as e:
        api.compress_in_memory(src, params(width=50, jpeg_optimize=True))
    assert e.value.code == 10201
    # the two Lanczos passes as separate kernels with the f32 image between them (what rows too wide for LDS still take): the same bytes as the fused kernel
    os.environ["CSH_RESIZE_TWO_PASS"] = "1"
    try:
        for (w, h) in [(150, 0), (97, 201), (333, 0)]:
            assert api.compress_in_memory(src, params(width=w, height=h)) == oracle_resized(src, w, h), (w, h)
    finally:
        del os.environ["CSH_RESIZE_TWO_PASS"]


def test_emul_sequential_output(api):
    """--jpeg-baseline: one interleaved sequential scan with optimal tables (DHT order DC0 AC0 DC1 AC1)"""
    from PIL import Image
    srcs = [synth_jpeg(3, 120, 88, texture=35), synth_jpeg(5, 97, 61, subsampling=0, texture=60), synth_jpeg(2, 104, 72, progressive=True, texture=20)]
    g = Image.fromarray(synth_rgb(7, 83, 55, 20)).convert("L")
    b = io.BytesIO(); g.save(b, format="JPEG", quality=90); srcs.append(b.getvalue())
    for src, out in zip(srcs, api.batch_compress(srcs, params(jpeg_progressive=False))):
        assert out == oracle_lossy(src, progressive=0)
        assert b"\xff\xc0" in out[:700] and b"\xff\xc2" not in out[:700]
    for src, out in zip(srcs, api.batch_compress(srcs, params(jpeg_progressive=False, jpeg_optimize=True))):
        assert out == oracle_lossless(src, progressive=0)
    # 16-bit quantisation entries (q=1) force SOF1
    out = api.compress_in_memory(srcs[0], params(jpeg_progressive=False, jpeg_quality=1))
    assert out == oracle_lossy(srcs[0], 1, progressive=0) and b"\xff\xc1" in out[:900]


def test_emul_lossless(api):
    srcs = [synth_jpeg(21, 133, 122, texture=20), synth_jpeg(2, 104, 72, progressive=True, texture=30), synth_jpeg(8, 64, 64, subsampling=0)]
    for src, out in zip(srcs, api.batch_compress(srcs, params(jpeg_optimize=True))):
        assert out == oracle_lossless(src)


def test_emul_scan_search_reproduces_reference_fixture(api, reference_samples):
    """BASELINE configs[0], through the kernels: `--lossless` on the reference's own mozjpeg-made sample runs the scan search on j0's
    coefficients (64 candidate scans coded in two stages, the decisions replayed on the host in between) and comes back with j0's own
    DQT..EOI -- the 8-scan script is not given, it is found.  j1 (stock script in) leaves with a searched script and fewer bytes."""
    d = open(os.path.join(reference_samples, "j0.JPG"), "rb").read()
    out = api.compress_in_memory(d, params(jpeg_optimize=True))
    assert out[out.index(b"\xff\xdb"):] == d[d.index(b"\xff\xdb"):]
    d1 = open(os.path.join(reference_samples, "level_1_0", "j1.jpg"), "rb").read()
    out1 = api.compress_in_memory(d1, params(jpeg_optimize=True))
    assert out1 == oracle_lossless(d1) and len(out1) < len(d1)


def test_emul_plain_profile_keeps_the_stock_script(api, monkeypatch):
    """CSH_PROFILE=plain: jpeg_simple_progression's ten scans, the profile that is pinned to libjpeg-turbo's bytes"""
    from oracle import oracle as O
    monkeypatch.setenv("CSH_PROFILE", "plain")
    src = synth_jpeg(5, 128, 96, texture=45)
    out = api.compress_in_memory(src, params())
    assert out == O.jpeg_compress(src, O.params(quality=80, scan_script=0)) == oracle_lossy(src)
    assert O.decode(out).scans() == O.stock_script(3, 0)
    monkeypatch.setenv("CSH_PROFILE", "scalar")   # the scan search over the scalar quantiser: the pieces pinned by j0.JPG and libjpeg-turbo
    assert api.compress_in_memory(src, params()) == O.jpeg_compress(src, O.params(quality=80, scan_script=2)) == oracle_lossy(src)
    monkeypatch.delenv("CSH_PROFILE")             # the default: what libcaesium's -q runs -- scan search + trellis quantisation + deringing
    assert api.compress_in_memory(src, params()) == O.jpeg_compress(src, O.params(quality=80, scan_script=2, trellis=1, deringing=1)) == oracle_lossy(src)


def test_emul_long_eob_runs_and_flat_images(api):
    """flat / near-flat images: EOB runs spanning thousands of blocks (incl. the 0x7FFF split at 520x512 luma blocks)"""
    from PIL import Image
    rng = np.random.default_rng(5)
    imgs = [np.full((64, 64, 3), 128, np.uint8), np.full((4160, 4096, 3), 77, np.uint8)]
    a = np.full((256, 256, 3), 90, np.uint8); a[100:108, 40:48] = rng.integers(0, 255, (8, 8, 3)); imgs.append(a)
    for im in imgs:
        b = io.BytesIO(); Image.fromarray(im).save(b, format="JPEG", quality=92, subsampling=2)
        src = b.getvalue()
        assert api.compress_in_memory(src, params()) == oracle_lossy(src)


def crafted_corrbit_stream(h=96, w=256):
    """every luma AC coefficient is +-2 or +-4 in a long stretch of blocks: at the last refinement scan each such block
    carries 63 correction bits and no newly significant coefficient, so the pending-bits limit (937) forces EOBRUN flushes"""
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    base = O.forward(rng.integers(0, 255, (h, w, 3), dtype=np.uint8), O.params(quality=90, subsampling=420))
    y = base.coefs_view(0)
    y[:, :, 1:] = rng.choice(np.array([-4, -2, 2, 4], dtype=np.int16), size=y[:, :, 1:].shape)
    y[2, 5, 7] = 1      # one newly significant coefficient in the middle, and a few blocks with none at all
    y[3, 0:4, 1:] = 0
    return base.encode(O.params(progressive=1, marker_style=0))


def test_emul_correction_bit_overflow_flush(api):
    """refinement scans with > 937 pending correction bits force an early EOBRUN flush (jcphuff MAX_CORR_BITS)"""
    blob = crafted_corrbit_stream()
    assert api.compress_in_memory(blob, params(jpeg_optimize=True)) == oracle_lossless(blob)
    big = crafted_corrbit_stream(384, 640)   # 1920 luma blocks in one run: the wave-per-run kernel (k_ac_runs_long) cuts it
    assert api.compress_in_memory(big, params(jpeg_optimize=True)) == oracle_lossless(big)


def fuzzed_blobs(seed, count, whole_file, scale=1):
    """bit flips, byte overwrites and deletions in the entropy-coded data (or anywhere in the file) of four kinds of source"""
    rng = np.random.default_rng(seed)
    k = scale
    srcs = [synth_jpeg(3, 120 * k, 88 * k, texture=30), synth_jpeg(4, 96 * k, 64 * k, progressive=True, texture=20), synth_jpeg(5, 104 * k, 72 * k, restart_rows=1, texture=25),
            synth_jpeg(6, 64 * k, 48 * k, subsampling=0, optimize=True, texture=10 * (k - 1))]
    blobs = []
    for k in range(count):
        s = bytearray(srcs[k % 4])
        start = 2 if whole_file else s.index(b"\xff\xda") + 14
        for _ in range(int(rng.integers(1, 4))):
            i = int(rng.integers(start, len(s) - 2))
            mode = int(rng.integers(0, 3))
            if mode == 0: s[i] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1: s[i] = int(rng.integers(0, 256))
            else: del s[i:i + int(rng.integers(1, 6))]
        blobs.append(bytes(s))
    return blobs


def test_emul_fuzzed_streams_agree_with_the_oracle(api):
    """damaged files: the device path and the oracle either both refuse a file or produce the same bytes (the decoders
    share libjpeg's rules for bad codes, overlong runs, data that ends early, markers in the data)"""
    for lossless in (True, False):
        for whole_file in (False, True):
            blobs = fuzzed_blobs(7 + 2 * lossless + whole_file, 48, whole_file)
            outs = api.batch_compress(blobs, params(jpeg_optimize=lossless))
            for i, (src, out) in enumerate(zip(blobs, outs)):
                try:
                    want = oracle_lossless(src) if lossless else oracle_lossy(src)
                except Exception as e:   # noqa: BLE001 - the oracle refuses the file
                    want = e
                assert isinstance(want, Exception) == isinstance(out, Exception), (lossless, whole_file, i, out, want)
                if not isinstance(out, Exception):
                    assert out == want, (lossless, whole_file, i)


def test_emul_batch_order_and_errors(api):
    good = [synth_jpeg(i, 80 + 8 * i, 64, texture=10 * i) for i in range(4)]
    blobs = [good[0], b"not an image", good[1], good[2][:150], good[3], b"\x89PNG\r\n\x1a\n" + b"\0" * 32]
    outs = api.batch_compress(blobs, params())
    assert [isinstance(o, Exception) for o in outs] == [False, True, False, True, False, True]
    assert outs[1].code == 10200 and outs[5].code == 10201
    assert outs[0] == oracle_lossy(good[0]) and outs[4] == oracle_lossy(good[3])


def _with_declared_size(src, w, h):
    """the same JPEG with another size in its SOF0"""
    i = src.index(b"\xff\xc0")
    return src[:i + 5] + bytes([h >> 8, h & 255, w >> 8, w & 255]) + src[i + 9:]


def test_emul_oversized_header_fails_alone(api):
    """a ~1 KB file that declares 65535 x 65535 must fail by itself (it used to size -- and overflow -- the whole batch's
    pools), and a two-byte SOS at the very end of a file must not be read past"""
    good = [synth_jpeg(i, 96, 64, texture=15) for i in range(3)]
    huge = _with_declared_size(good[0], 65535, 65535)
    i = good[1].index(b"\xff\xda")
    short_sos = good[1][:i] + b"\xff\xda\x00\x02"
    outs = api.batch_compress([good[0], huge, good[1], short_sos, good[2]], params())
    assert [isinstance(o, Exception) for o in outs] == [False, True, False, True, False]
    assert "too large" in str(outs[1]) and "SOS" in str(outs[3])
    assert outs[0] == oracle_lossy(good[0]) and outs[4] == oracle_lossy(good[2])


def test_emul_batch_extent_splits_by_bytes_and_declared_pixels(api):
    """scan offsets are 32-bit, so a device batch is cut by input bytes (2 GiB) and by the pools its headers announce (96 GiB),
    not by file count alone"""
    head = synth_jpeg(0, 64, 48)[:700]
    assert api.batch_extent([(len(head), head)] * 5000) == 2048                      # count cap
    assert api.batch_extent([(5 << 20, head)] * 1024) == 409                         # 2 GiB / 5 MiB, before the sum passes the cap
    big = _with_declared_size(head, 16000, 16000)                                    # 256 MP -> ~6.4 GB of pools each
    assert api.batch_extent([(len(big), big)] * 100) == 16
    assert api.batch_extent([(3 << 30, head)] * 3) == 1                              # a single file always goes through, alone
    # a WebP header's canvas counts too (VP8L: 14 + 14 bits behind the signature; VP8X: 24-bit sizes) -- a tiny file may declare 16383 x 16383
    bits = (16383 - 1) | ((16383 - 1) << 14)
    vp8l = b"RIFF" + (26).to_bytes(4, "little") + b"WEBPVP8L" + (14).to_bytes(4, "little") + b"\x2f" + bits.to_bytes(4, "little") + bytes(9)
    assert api.batch_extent([(len(vp8l), vp8l)] * 100) == 15
    vp8x = b"RIFF" + (30).to_bytes(4, "little") + b"WEBPVP8X" + (10).to_bytes(4, "little") + bytes(4) + (15999).to_bytes(3, "little") + (15999).to_bytes(3, "little") + bytes(8)
    assert api.batch_extent([(len(vp8x), vp8x)] * 100) == 16


def reference_size_walk(src, max_size, return_smallest=True, encode=None):
    """libcaesium's bisection restated: -> (quality sequence, bytes or None).  encode(src, q): the one-try engine (JPEG lossy by default)"""
    oracle_lossy = encode or globals()["oracle_lossy"]
    tol = max_size * 2 // 100
    q, less, high, seq = 80, 1, 101, []
    for _ in range(10):
        seq.append(q)
        out = oracle_lossy(src, q)
        if len(out) <= max_size and max_size - len(out) < tol:
            return seq, out
        if len(out) <= max_size:
            less = q
        else:
            high = q
        nq = min(max((high + less) // 2, 1), 100)
        if nq == q:
            if q == 1 and high == 1 and not return_smallest:
                return seq, None
            return seq, out
        q = nq
    return seq, None


def test_emul_compress_to_size_and_convert(api):
    src = synth_jpeg(12, 320, 200, texture=30)
    for target in (len(oracle_lossy(src, 60)), len(oracle_lossy(src, 93)) + 3, len(src) * 4, 2000):
        p = params()
        seq, want = reference_size_walk(src, target)
        assert want is not None
        assert api.compress_to_size_in_memory(src, p, target) == want
        assert p.jpeg_quality == seq[-1]          # &mut CSParameters: the quality fields are left at the last try
    # the j0.JPG walk of SURVEY 2b: 80,40,60,50,55,52,51
    seq, _ = reference_size_walk(src, len(oracle_lossy(src, 51)) + 1)
    assert seq[:2] == [80, 40]
    tiny = api.compress_to_size_in_memory(src, params(), 10, True)       # unreachable: the q=1 file comes back
    assert tiny == oracle_lossy(src, 1)
    with pytest.raises(package().CaesiumError) as e:
        api.compress_to_size_in_memory(src, params(), 10, False)
    assert e.value.code == 10500
    with pytest.raises(package().CaesiumError) as e:
        api.convert_in_memory(src, params(), 0)
    assert e.value.code == 10407


def test_emul_batch_size_targeting_and_requant_equivalence(api):
    """--max-size over a batch: one decode + DCT, then only re-quantise/re-code per round; every file's result equals the
    restated libcaesium walk, and a re-quantised run at quality q equals a full run at q"""
    srcs = [synth_jpeg(i, 160 + 16 * i, 120, subsampling=(0, 2, 1)[i % 3], texture=10 + 9 * i) for i in range(5)] + [b"junk"]
    target = 5000
    outs = api.batch_compress_to_size(srcs, params(), target)
    assert isinstance(outs[5], Exception) and outs[5].code == 10200
    for src, out in zip(srcs[:5], outs[:5]):
        assert out == reference_size_walk(src, target)[1]
    b = api.batch(srcs[:5], params())
    b.retain_dct()
    b.run()
    for qs in ([33, 0, 97, 5, 61], [80, 80, 1, 100, 0]):
        b.set_quality(qs)
        b.rerun_encode()
        now = [q or prev for q, prev in zip(qs, getattr(test_emul_batch_size_targeting_and_requant_equivalence, "_last", [80] * 5))]
        test_emul_batch_size_targeting_and_requant_equivalence._last = now
        for src, out, q in zip(srcs[:5], b.fetch(), now):
            assert out == oracle_lossy(src, q), q
    del test_emul_batch_size_targeting_and_requant_equivalence._last


def test_emul_rerun_after_a_run_that_took_the_conditional_stages(api):
    """A re-quantised run of the same batch starts from "no work item is in a file": the conditional stages of the scan search (luma Al 3,
    the splits at 12 / 18) that ran -- and placed their winners -- in one run and do not run in the next must leave nothing behind
    (k_reset_works; ADVICE r03: stale ScanWork records of a skipped stage).  Runs at q 80 (all three extra stages, late splits win), then
    q 30 (none), then q 80 again: every file equals a fresh encode at that quality."""
    rich = [(0, 90), (0, 10), (4, 90), (1, 90)]
    srcs = [synth_jpeg(sd, 160, 120, texture=tx) for sd, tx in rich] + [synth_jpeg(0, 160, 120, texture=0)]
    b = api.batch(srcs, params(jpeg_quality=80))
    b.retain_dct()
    t = b.run()
    assert t.n_search_extra == 3
    for q, extra in ((30, None), (80, 3), (95, None), (30, None)):
        b.set_quality([q] * len(srcs))
        t = b.rerun_encode()
        if extra is not None:
            assert t.n_search_extra == extra
        for src, out in zip(srcs, b.fetch()):
            assert out == oracle_lossy(src, q), q


# ---- the product library: loads and exports everything include/caesium_hip.h declares (no compute without a GPU)
def test_product_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "caesium_hip.h")).read()
    declared = set(re.findall(r"\b(cs(?:h|p|wd|l)?_[a-z_0-9]+)\s*\(", hdr))
    assert {"cs_compress_in_memory", "cs_compress_to_size_in_memory", "cs_convert_in_memory", "cs_batch_compress"} <= declared
    path = package().library_path()
    assert os.path.exists(path), "libcaesium_hip.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(path)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(package().binding.EXPORTS)


def test_product_library_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    api = package().load()
    assert api.device_count() == 0
    with pytest.raises(package().CaesiumError) as e:
        api.compress_in_memory(synth_jpeg(0, 64, 48), params())
    assert e.value.code == 10001


def test_emul_gif_and_tiff_files_are_passed_through(api):
    """SURVEY 2 rows 9-10 (out of the JPEG / PNG / WebP scope: "passthrough"): a GIF or TIFF among the inputs comes back as it is, with
    Success, in its place -- the batch keeps its order and the other files are compressed as usual; a resize of such a file is refused"""
    gif = b"GIF89a" + (16).to_bytes(2, "little") + (8).to_bytes(2, "little") + b"\x80\x00\x00" + bytes(6) + b"\x2c" + bytes(9) + b"\x02\x02\x44\x01\x00\x3b"
    tif = b"II*\x00\x08\x00\x00\x00" + b"\x00\x00" + bytes(4)
    jpg = synth_jpeg(3, 64, 48, texture=10)
    outs = api.cs_batch_compress([gif, jpg, tif, jpg], params())
    assert outs[0] == gif and outs[2] == tif and outs[1] == oracle_lossy(jpg) and outs[3] == outs[1]
    assert api.compress_in_memory(gif, params()) == gif
    sized = api.batch_compress_to_size([tif, jpg], params(), 4000)
    assert sized[0] == tif and sized[1] == reference_size_walk(jpg, 4000)[1]
    with pytest.raises(package().CaesiumError) as e:
        api.compress_in_memory(gif, params(width=8))
    assert e.value.code == 10407 or e.value.code > 0
