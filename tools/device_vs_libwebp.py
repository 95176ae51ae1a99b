"""The device against libwebp ITSELF (every libwebp on the box, through ctypes): RGB pixels -> PNG -> `cs_batch_convert(.., WebP)` on the MI355X -> bytes,
against WebPEncode with a default WebPConfig at the same quality on the same pixels.  `python tools/device_vs_libwebp.py [n_pictures] [width height]`
prints one line per picture and the count of byte-identical files (the oracle's file is compared too: three-way)."""
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
from PIL import Image                                # noqa: E402
from _util import package, product_api               # noqa: E402
from gen_synth import synth_rgb                      # noqa: E402
from libwebp_pin import libwebp_encode, libwebps      # noqa: E402
from oracle import oracle as O                       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1500, 844)
api, pkg, libs = product_api(), package(), libwebps()
print("libwebp versions on the box:", [v for v, _ in libs])
same = 0
for q in (85, 40, 100, 10):
    k = n if q == 85 else max(1, n // 8)
    rgbs = [np.ascontiguousarray(synth_rgb(seed + 100 * q, w, h)) for seed in range(k)]
    pngs = []
    for rgb in rgbs:
        b = io.BytesIO(); Image.fromarray(rgb).save(b, "PNG", compress_level=1); pngs.append(b.getvalue())
    outs = api.batch_convert(pngs, pkg.default_parameters(webp_quality=q), 3)
    for i, (rgb, out) in enumerate(zip(rgbs, outs)):
        refs = [libwebp_encode(W, rgb, q) for _, W in libs]
        ok = isinstance(out, bytes) and all(out == r for r in refs) and out == O.vp8enc_encode_rgb(rgb, q)
        same += ok
        print("q%-3d picture %2d %dx%d: device %s B, libwebp %d B: %s" % (q, i, w, h, len(out) if isinstance(out, bytes) else out, len(refs[0]), "identical" if ok else "DIFFERENT"))
total = n + 3 * max(1, n // 8)
print("%d of %d files byte-identical: device == oracle == libwebp %s" % (same, total, " / ".join(v for v, _ in libs)))
