#!/usr/bin/env python3
"""Device time of the WebP decoder alone (cswd_batch_run): `tools/webp_decode_bench.py [files] [width] [height]` -- lossy (q90, method 2) and lossless
pictures of the synthetic recipe, batches of `files` with 8 distinct ones."""
import io
import os
import sys
import time
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
from PIL import Image                      # noqa: E402
from _util import package, product_api    # noqa: E402
from gen_synth import synth_rgb            # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
api = product_api()
CByteArray = package().binding.CByteArray
L = api.L
for kind, kw in (("lossy q90", dict(quality=90, method=2)), ("lossless", dict(lossless=True, method=2))):
    src = []
    for k in range(8):
        b = io.BytesIO()
        Image.fromarray(synth_rgb(300 + k, w, h), "RGB").save(b, format="WEBP", **kw)
        src.append(b.getvalue())
    blobs = [src[k % 8] for k in range(n)]
    keep = [C.create_string_buffer(x, len(x)) for x in blobs]
    ins = (CByteArray * n)()
    for i, buf in enumerate(keep):
        ins[i].data = C.cast(buf, C.POINTER(C.c_uint8)); ins[i].length = len(blobs[i])
    for rep in range(2):
        hnd = C.c_void_p()
        t0 = time.time()
        rc = L.cswd_batch_create(ins, n, 0, C.byref(hnd))
        t1 = time.time()
        rc = rc or L.cswd_batch_run(hnd)
        t2 = time.time()
        L.cswd_batch_destroy(hnd)
        print(f"{kind}: {n} files {w}x{h} ({sum(map(len, blobs)) / n / 1e3:.0f} KB each): create {t1 - t0:.3f} s, run {t2 - t1:.3f} s, rc {rc}; {n * w * h / 1e6 / (t2 - t1):.0f} MP/s", flush=True)
