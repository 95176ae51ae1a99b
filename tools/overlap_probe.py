"""Does the chip have room for two halves of a batch at once?  One batch of N files on one stream against two batches of N/2 (own streams,
two host threads): the kernels are bound by instruction issue with waves stalled most of their life, so independent work may fill the gaps."""
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
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pkg = package()
api = pkg.load()
u = make_inputs(0, 64)
blobs = [u[i % 64] for i in range(n)]
p = pkg.default_parameters(jpeg_quality=80)
one = api.batch(blobs, p)
one.run()
t0 = time.perf_counter()
for _ in range(3):
    one.run()
t1 = (time.perf_counter() - t0) / 3
one.close()
bs = [api.batch(blobs[k::parts], p) for k in range(parts)]
for b in bs:
    b.run()


def loop(b):
    for _ in range(3):
        b.run()


t0 = time.perf_counter()
th = [threading.Thread(target=loop, args=(b,)) for b in bs]
for t in th:
    t.start()
for t in th:
    t.join()
t2 = (time.perf_counter() - t0) / 3
print(f"profile={os.environ.get('CSH_PROFILE', 'default')} one batch of {n}: {t1 * 1e3:.1f} ms/step ({n * 2.0736 / t1 / 1e3:.2f} GP/s); {parts} concurrent batches of {n // parts}: {t2 * 1e3:.1f} ms/step ({n * 2.0736 / t2 / 1e3:.2f} GP/s)")
