"""JPEG -> WebP on the device: `python tools/webp_bench.py [files] [distinct] [quality] [long_edge]` -- configs[3] shape by default
(synthetic 1920x1080 q92 4:2:0 JPEGs, -q 85 --format webp --long-edge 1500), device time per step, sizes against libwebp."""
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
from _util import package, product_api   # noqa: E402
from gen_synth import synth_jpeg          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 8
q = int(sys.argv[3]) if len(sys.argv) > 3 else 85
edge = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
api, pkg = product_api(), package()
src = [synth_jpeg(k) for k in range(distinct)]
blobs = [src[k % distinct] for k in range(n)]
p = pkg.default_parameters(webp_quality=q, jpeg_quality=q, width=edge)
for rep in range(2):
    t0 = time.time()
    b = api.webp_batch(blobs, p)
    t1 = time.time()
    tm = b.run()
    t2 = time.time()
    outs = b.fetch()
    t3 = time.time()
    names = api.kernel_names()
    tail = [i for i in range(len(names)) if tm.kernel_ms[i] > 0][-1]
    print(f"rep {rep}: {n} files -> webp q{q} long edge {edge}: create {t1 - t0:.2f}s run {t2 - t1:.2f}s (device {tm.total_ms:.1f} ms, of which the WebP tail "
          f"{tm.kernel_ms[tail]:.1f} ms) fetch {t3 - t2:.2f}s; {n * 1920 * 1080 / 1e6 / (tm.total_ms / 1e3):.0f} source MP/s, {n / (tm.total_ms / 1e3):.0f} files/s")
    b.close()
bad = [o for o in outs if isinstance(o, Exception)]
print("errors", len(bad), "sizes: input", [len(s) for s in src[:4]], "webp", [len(o) for o in outs[:4]])
try:
    from PIL import Image
    im = Image.open(io.BytesIO(src[0])).resize((edge, round(edge * 1080 / 1920)), Image.LANCZOS)
    bb = io.BytesIO(); im.save(bb, "WEBP", quality=q)
    print("libwebp (Pillow) at the same setting:", len(bb.getvalue()), "bytes")
except Exception as e:
    print("no Pillow comparison:", e)
