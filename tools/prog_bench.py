"""Device time of the decode phase by input class: baseline, progressive (Pillow's stock script; and this library's own -q 80 output -- the mozjpeg
script with EOB runs, what caesium itself writes), restart intervals.  CSH_PROG_PAR=0 puts every progressive scan back on the chains, =1 only the
refinement scans (one wave per chain, as before k_decode_refine.hip).
usage: python tools/prog_bench.py [files=512]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _util import package
from gen_synth import synth_jpeg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pkg = package(); api = pkg.load()
names = api.kernel_names()
own = api.batch_compress([synth_jpeg(i) for i in range(8)], pkg.default_parameters(jpeg_quality=80))   # progressive files as this library writes them
classes = (("baseline", [synth_jpeg(i) for i in range(8)]), ("progressive (stock script)", [synth_jpeg(i, progressive=True) for i in range(8)]),
           ("progressive (own -q 80 output)", own), ("dri", [synth_jpeg(i, restart_rows=1) for i in range(8)]))
for name, uniq in classes:
    blobs = [uniq[i % 8] for i in range(n)]
    for lossless in (True,):
        b = api.batch(blobs, pkg.default_parameters(jpeg_quality=80, jpeg_optimize=lossless), device=0)
        b.run()
        t = [b.run() for _ in range(3)][-1]
        dec = sum(t.kernel_ms[i] for i in range(9))
        print(f"{name:32s} files={n} lossless decode_ms={dec:.2f} total_ms={t.total_ms:.2f} seq={t.n_seq_decoded} prog={t.n_prog_decoded}",
              {k: round(v, 2) for k, v in zip(names[:9], t.kernel_ms[:9]) if v > 0.5}, flush=True)
        b.close()
