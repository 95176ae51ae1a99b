"""Does the device finish 2048 files sooner as one batch (one stream) or as K batches run side by side (K streams)?  usage: concurrent_probe.py [files] [unique]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from _util import package
from bench import make_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
uniq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pkg = package()
api = pkg.load()
blobs = make_inputs(0, uniq)
blobs = [blobs[i % uniq] for i in range(n)]
params = pkg.default_parameters(jpeg_quality=80)
for k in (1, 2, 3, 4):
    parts = [blobs[i::k] for i in range(k)]
    batches = [api.batch(p, params) for p in parts]
    for b in batches:
        b.run()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        th = [threading.Thread(target=b.run) for b in batches]
        for t in th: t.start()
        for t in th: t.join()
        best = min(best, time.perf_counter() - t0)
    print(f"{k} batch(es) of {n // k} files side by side: {best * 1e3:.1f} ms wall = {n * 2.0736 / (best * 1e3):.2f} GP/s", flush=True)
    for b in batches:
        b.close()
