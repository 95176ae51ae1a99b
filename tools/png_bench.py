"""The PNG row on the device: `python tools/png_bench.py [files] [distinct] [width] [height] [level] [quality]` (a quality selects the lossy form, `-q` on a PNG) -- a batch of synthetic
RGB8 PNGs (configs[2] shape by default), per-kernel device times, sizes against the input (Pillow level 6) and zlib -9."""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
from _util import package, product_api   # noqa: E402
from gen_synth import synth_png           # noqa: E402

zopfli = "--zopfli" in sys.argv   # png.force_zopfli: more passes of the cost model
if zopfli:
    sys.argv.remove("--zopfli")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
h = int(sys.argv[4]) if len(sys.argv) > 4 else 2160
level = int(sys.argv[5]) if len(sys.argv) > 5 else 3
api, pkg = product_api(), package()
src = [synth_png(100 + k, w, h, "RGB", texture=float(k % 4)) for k in range(distinct)]
blobs = [src[k % distinct] for k in range(n)]
quality = int(sys.argv[6]) if len(sys.argv) > 6 else 0
p = pkg.default_parameters(png_quality=quality) if quality else pkg.default_parameters(png_optimize=True, png_optimization_level=level, png_force_zopfli=zopfli)
for rep in range(2):
    t0 = time.time()
    b = api.png_batch(blobs, p)
    t1 = time.time()
    tm = b.run()
    t2 = time.time()
    outs = b.fetch()
    t3 = time.time()
    names = api.png_kernel_names()
    print(f"rep {rep}: {n} files {w}x{h} level {level}: create {t1 - t0:.2f}s run {t2 - t1:.2f}s (device {tm.total_ms:.1f} ms) fetch {t3 - t2:.2f}s; "
          f"{n * w * h / 1e6 / (tm.total_ms / 1e3):.0f} MP/s, {n / (tm.total_ms / 1e3):.1f} files/s")
    print("   " + ", ".join(f"{names[i]} {tm.kernel_ms[i]:.1f}" for i in range(len(names)) if names[i] and tm.kernel_ms[i] > 0.05))
    tr = b.trials(0)
    b.close()
print("sizes: input", [len(s) for s in src], "output", [len(outs[k]) for k in range(distinct)], "trials of file 0", tr)
