"""Where the time of the boundary call goes: cs_batch_compress over N x 1080p files from host buffers, by group size and worker count (CSH_TRACE=1
prints create / run / fetch / destroy per device batch on stderr).  usage: python tools/boundary_probe.py [files=2048]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _util import package
from gen_synth import synth_jpeg
import multiprocessing as mp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pkg = package(); api = pkg.load()
with mp.get_context("fork").Pool(32) as pool:
    uniq = pool.map(synth_jpeg, range(64))
blobs = [uniq[i % 64] for i in range(n)]
params = pkg.default_parameters(jpeg_quality=80)
api.cs_batch_compress(blobs[:256], params, device=0)
for group, workers in ((256, 2), (256, 3), (512, 2), (512, 3), (512, 4)):
    os.environ["CSH_GROUP"] = str(group); os.environ["CSH_WORKERS"] = str(workers)
    best = 1e9
    for rep in range(2):
        tm = []
        res = api.cs_batch_compress(blobs, params, device=0, timing=tm)
        best = min(best, tm[0])
    ok = sum(1 for r in res if isinstance(r, bytes))
    print(f"group={group} workers={workers}: {best * 1e3:.1f} ms for {n} files ({ok} ok) = {n * 2.0736 / best / 1e3:.2f} GP/s", flush=True)
os.environ["CSH_GROUP"] = "256"; os.environ["CSH_WORKERS"] = "2"; os.environ["CSH_TRACE"] = "1"
api.cs_batch_compress(blobs, params, device=0)
