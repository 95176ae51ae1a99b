import sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _util import package
from gen_synth import synth_jpeg, synth_rgb
from PIL import Image
pkg = package(); api = pkg.load()
def run(name, blobs, n=64, **p):
    b = api.batch([blobs[i % len(blobs)] for i in range(n)], pkg.default_parameters(jpeg_quality=80, **p), device=0)
    b.run(); t = b.run()
    mp = t.pixels / 1e6
    print(f"{name:34s} n={n:5d} ms={t.total_ms:8.2f}  {mp / t.total_ms:7.1f} GP/s  seq={t.n_seq_decoded} fb={t.n_par_fallback} prog={t.n_prog_decoded} fail={t.n_failed}",
          {k: round(v, 1) for k, v in zip(api.kernel_names(), t.kernel_ms) if v > 0.25 * t.total_ms}, flush=True)
run("q50 4:2:0", [synth_jpeg(i, quality=50) for i in range(4)])
run("q20 4:2:0", [synth_jpeg(i, quality=20) for i in range(4)])
run("q92 optimised tables", [synth_jpeg(i, optimize=True) for i in range(4)])
run("q75 optimised tables", [synth_jpeg(i, quality=75, optimize=True) for i in range(4)])
run("q92 4:4:4", [synth_jpeg(i, subsampling=0) for i in range(4)])
run("q92 4:2:2", [synth_jpeg(i, subsampling=1) for i in range(4)])
g = []
for i in range(4):
    b = io.BytesIO(); Image.fromarray(synth_rgb(i, 1920, 1080, 10)).convert("L").save(b, format="JPEG", quality=90); g.append(b.getvalue())
run("grayscale q90", g)
run("thumbnails 160x120", [synth_jpeg(i, 160, 120) for i in range(16)], n=2048)
run("4000x3000 q92", [synth_jpeg(i, 4000, 3000) for i in range(2)], n=8)
run("q92 -> 4:4:4 out", [synth_jpeg(i) for i in range(4)], jpeg_chroma_subsampling=444)
run("q92 lossless", [synth_jpeg(i) for i in range(4)], jpeg_optimize=True)
run("q92 baseline out", [synth_jpeg(i) for i in range(4)], jpeg_progressive=False)
run("q92 resize to 1500 wide", [synth_jpeg(i) for i in range(4)], width=1500)
