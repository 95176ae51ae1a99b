"""cs_batch_compress from host buffers, warm, with the library's own stage times (CSH_TRACE=1): where the boundary call spends its time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["CSH_TRACE"] = "1"
from _util import package
from bench import make_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pkg = package()
api = pkg.load()
u = make_inputs(0, 64)
blobs = [u[i % 64] for i in range(n)]
p = pkg.default_parameters(jpeg_quality=80)
for k in range(3):
    t0 = time.perf_counter()
    res = api.cs_batch_compress(blobs, p)
    dt = time.perf_counter() - t0
    print(f"call {k}: {n} files in {dt * 1e3:.1f} ms = {n * 2.0736 / dt / 1e3:.2f} GP/s, ok {sum(isinstance(r, bytes) for r in res)}", flush=True)
