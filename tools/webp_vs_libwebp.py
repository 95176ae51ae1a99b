"""The lossy WebP encoder (oracle/webp_oracle.c = the device's bytes) against libwebp's own encoder at the SAME quantiser index: libwebp driven through ctypes
(tests/test_oracle_webp.py::libwebp_encode) with what this encoder does not have switched off (one segment, no SNS, no loop filter), with its defaults, and at
method 0.  `python tools/webp_vs_libwebp.py > profiles/r02_webp_vs_libwebp.txt`"""
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
from PIL import Image                          # noqa: E402
from gen_synth import synth_rgb                # noqa: E402
from oracle import oracle as O                 # noqa: E402
from test_oracle_webp import libwebp_encode    # noqa: E402


def psnr(data, rgb):
    a = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(np.float64)
    return 10 * np.log10(255.0 ** 2 / ((a - rgb) ** 2).mean())


pics = [("synth0 1500x844", np.ascontiguousarray(synth_rgb(0, 1500, 844)))]
j0 = os.path.join(ROOT, "tests", "golden", "reference_samples", "j0.JPG")
if os.path.exists(j0):
    pics.append(("j0.JPG at 1500x2250", np.ascontiguousarray(np.asarray(Image.open(j0).convert("RGB").resize((1500, 2250), Image.LANCZOS)))))
for name, rgb in pics:
    for q in (50, 75, 85, 92):
        rows = (("this encoder", O.webp_encode_rgb(rgb, q)), ("libwebp m4, 1 segment, no SNS, no filter", libwebp_encode(rgb, q)),
                ("libwebp m4 defaults (4 segments, SNS 50, filter 60)", libwebp_encode(rgb, q, 4, 50, 60)), ("libwebp m0, 1 segment, no SNS, no filter", libwebp_encode(rgb, q, method=0)))
        for label, d in rows:
            print("%-20s q%-3d %-52s %8d B  %.2f dB" % (name, q, label, len(d), psnr(d, rgb)))
