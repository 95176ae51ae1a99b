import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _util import package
from gen_synth import synth_jpeg
pkg = package(); api = pkg.load()
for q, tex in ((92, 0), (95, 20), (98, 40)):
    u = [synth_jpeg(i, quality=q, texture=tex) for i in range(2)]
    b = api.batch(u, pkg.default_parameters(jpeg_quality=80), device=0)
    t = b.run()
    print(f"q{q} texture {tex}: {len(u[0]) / 48960:.0f} B/block seq={t.n_seq_decoded}", flush=True)
