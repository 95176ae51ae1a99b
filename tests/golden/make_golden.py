"""Regenerates tests/golden/*: known-answer vectors for the JPEG hot path.

Two sources (both pin the ORACLE; the GPU path is then compared with the oracle):
 1. libjpeg-turbo (through Pillow) -- the same ISLOW integer pipeline mozjpeg uses: source JPEGs of
    synthetic images and the byte-exact output libjpeg-turbo gives for "decode to YCbCr, re-encode
    q80 with mozjpeg base table #3, 4:2:0, progressive, optimised Huffman".
 2. the reference's own fixtures /root/reference/samples/{j0.JPG,level_1_0/j1.jpg} -- only digests are
    committed (the files themselves stay in /root/reference): sha256 of the DQT..EOI tail, which the
    oracle must reproduce when it re-encodes the decoded coefficients with the file's own scan script.
Run here (needs Pillow and /root/reference):  python tests/golden/make_golden.py
"""
import hashlib
import io
import json
import os
import sys

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
from gen_synth import synth_jpeg  # noqa: E402

MOZ3 = [16, 16, 16, 18, 25, 37, 56, 85, 16, 17, 20, 27, 34, 40, 53, 75, 16, 20, 24, 31, 43, 62, 91, 135, 18, 27, 31, 40, 53, 74, 106, 156,
        25, 34, 43, 53, 69, 94, 131, 189, 37, 40, 62, 74, 94, 124, 169, 238, 56, 53, 91, 106, 131, 169, 226, 311, 85, 75, 135, 156, 189, 238, 311, 418]


def table(q):
    s = 5000 // q if q < 50 else 200 - 2 * q
    return [min(max((b * s + 50) // 100, 1), 32767) for b in MOZ3]


def pil_transcode(src, q, subsampling=2, progressive=True):
    im = Image.open(io.BytesIO(src)); im.draft("YCbCr", im.size); im.load()
    t = table(q)
    b = io.BytesIO()
    kw = dict(qtables=[t, t], subsampling=subsampling) if im.mode == "YCbCr" else dict(qtables=[t])
    im.save(b, format="JPEG", progressive=progressive, optimize=True, **kw)
    return b.getvalue()


def main():
    manifest = {"libjpeg_turbo": features.version("libjpeg_turbo"), "cases": [], "reference": {}}
    cases = [(0, 101, 67, 2, False, 0), (1, 160, 120, 2, False, 0), (2, 104, 72, 1, True, 0), (3, 64, 48, 0, False, 0), (4, 50, 34, 2, True, 0),
             (5, 128, 96, 2, False, 45), (6, 97, 61, 2, False, 80)]
    for seed, w, h, ss, prog, tex in cases:
        src = synth_jpeg(seed, w, h, subsampling=ss, progressive=prog, texture=tex)
        name = f"synth{seed}_{w}x{h}_ss{ss}{'_prog' if prog else ''}{'_tex' if tex else ''}"
        open(os.path.join(HERE, name + ".src.jpg"), "wb").write(src)
        for q in (80, 51):
            out = pil_transcode(src, q)
            open(os.path.join(HERE, f"{name}.q{q}.jpg"), "wb").write(out)
        manifest["cases"].append({"name": name, "qualities": [80, 51]})
    ref = "/root/reference/samples"
    for rel in ("j0.JPG", "level_1_0/j1.jpg"):
        d = open(os.path.join(ref, rel), "rb").read()
        tail = d[d.index(b"\xff\xdb"):]
        manifest["reference"][rel] = {"size": len(d), "sha256_file": hashlib.sha256(d).hexdigest(),
                                      "tail_len": len(tail), "sha256_dqt_to_eoi": hashlib.sha256(tail).hexdigest()}
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
