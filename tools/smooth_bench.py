import sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _util import package
from gen_synth import synth_rgb
from PIL import Image, ImageFilter
pkg = package(); api = pkg.load()
for name, radius, q in (("sharp q92", 0, 92), ("blur 2 q92", 2, 92), ("blur 6 q92", 6, 92), ("blur 6 q75", 6, 75), ("flat", -1, 90)):
    blobs = []
    for i in range(4):
        im = Image.fromarray(synth_rgb(i, 1920, 1080, 0)) if radius >= 0 else Image.new("RGB", (1920, 1080), (90, 120, 200))
        if radius > 0: im = im.filter(ImageFilter.GaussianBlur(radius))
        b = io.BytesIO(); im.save(b, format="JPEG", quality=q, subsampling=2); blobs.append(b.getvalue())
    bt = api.batch([blobs[i % 4] for i in range(256)], pkg.default_parameters(jpeg_quality=80), device=0)
    bt.run(); t = bt.run()
    print(f"{name:12s} {len(blobs[0]) / 1e3:7.0f} KB/file  ms={t.total_ms:7.2f}", {k: round(v, 2) for k, v in zip(api.kernel_names(), t.kernel_ms) if v > 0.08 * t.total_ms}, flush=True)
