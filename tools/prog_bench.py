import sys, time, os
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _util import package
from gen_synth import synth_jpeg
pkg = package(); api = pkg.load()
for name, kw in (("baseline", {}), ("progressive", {"progressive": True}), ("dri", {"restart_rows": 1})):
    uniq = [synth_jpeg(i, **kw) for i in range(8)]
    for n in (1, 64, 512):
        blobs = [uniq[i % 8] for i in range(n)]
        for lossless in (False, True):
            p = pkg.default_parameters(jpeg_quality=80, jpeg_optimize=lossless)
            b = api.batch(blobs, p, device=0)
            b.run()
            t = [b.run() for _ in range(3)][-1]
            print(name, n, "lossless" if lossless else "q80", "ms=%.2f" % t.total_ms, "seq=%d" % t.n_seq_decoded, {k: round(v, 2) for k, v in zip(api.kernel_names(), t.kernel_ms) if v > 1.0}, flush=True)
