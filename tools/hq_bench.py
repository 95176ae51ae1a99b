import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _util import package
from gen_synth import synth_jpeg
pkg = package(); api = pkg.load()
for q, tex, ss in ((98, 40, 2), (100, 40, 2), (100, 100, 2), (100, 100, 0), (100, 255, 0)):
    u = [synth_jpeg(i, quality=q, texture=tex, subsampling=ss) for i in range(4)]
    b = api.batch([u[i % 4] for i in range(64)], pkg.default_parameters(jpeg_quality=80), device=0)
    b.run(); t = b.run()
    print(f"q{q} texture {tex} ss {ss}: {sum(map(len, u)) / 4 / 1e6:.2f} MB/file, {sum(map(len, u)) / 4 / (48960 if ss == 2 else 97920):.0f} B/block, ms={t.total_ms:.1f} seq={t.n_seq_decoded} fallback={t.n_par_fallback}",
          {k: round(v, 2) for k, v in zip(api.kernel_names(), t.kernel_ms) if v > 2.0}, flush=True)
