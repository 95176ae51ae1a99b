import sys, os, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _util import package
from gen_synth import synth_jpeg
pkg = package(); api = pkg.load()
def run(name, blobs, n=64, **p):
    b = api.batch([blobs[i % len(blobs)] for i in range(n)], pkg.default_parameters(jpeg_quality=80, **p), device=0)
    b.run(); t = b.run()
    print(f"{name:34s} n={n:5d} ms={t.total_ms:8.2f}  {t.pixels / 1e6 / t.total_ms:7.1f} GP/s  seq={t.n_seq_decoded} fb={t.n_par_fallback} prog={t.n_prog_decoded} fail={t.n_failed}",
          {k: round(v, 1) for k, v in zip(api.kernel_names(), t.kernel_ms) if v > 0.25 * t.total_ms}, flush=True)
run("8000x6000 q92", [synth_jpeg(1, 8000, 6000)], n=2)
run("mixed sizes", [synth_jpeg(i, w, h) for i, (w, h) in enumerate([(1920, 1080), (640, 480), (3000, 2000), (333, 777), (64, 64), (1280, 720), (17, 9), (2048, 2048)])], n=256)
run("progressive q75", [synth_jpeg(i, quality=75, progressive=True) for i in range(4)])
run("progressive q95", [synth_jpeg(i, quality=95, progressive=True) for i in range(4)])
blobs = [synth_jpeg(i) for i in range(8)]
for target in (300_000, 150_000, 60_000):
    t0 = time.perf_counter()
    outs = api.batch_compress_to_size(blobs * 32, pkg.default_parameters(), target, True)
    dt = time.perf_counter() - t0
    sizes = [len(o) for o in outs if not isinstance(o, Exception)]
    print(f"--max-size {target}: 256 files in {dt * 1e3:.0f} ms (host call incl. parse/upload/fetch each round), sizes {min(sizes)}..{max(sizes)}, failures {256 - len(sizes)}", flush=True)
