import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _util import package
from gen_synth import synth_jpeg
pkg = package(); api = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
u = [synth_jpeg(i, progressive=True) for i in range(4)]
b = api.batch([u[i % 4] for i in range(n)], pkg.default_parameters(jpeg_quality=80, jpeg_optimize=True), device=0)
t = b.run()
print("ms", t.total_ms, "prog", t.n_prog_decoded)
