"""Where k_trellis_ac's time goes: CSH_TR_DEBUG switches (1: stop after lambda; 2: no programme; 4: no output) x CSH_TR_SORT (blocks in
order of list length or in raster order).  usage: python tools/trellis_probe.py [files=1024] [unique=64]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from _util import package
from bench import make_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
uniq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pkg = package()
api = pkg.load()
blobs = make_inputs(0, uniq)
blobs = [blobs[i % uniq] for i in range(n)]
names = api.kernel_names()
for sort in ("1", "0"):
    for dbg in ("0", "1", "2", "4"):
        os.environ["CSH_TR_SORT"] = sort
        os.environ["CSH_TR_DEBUG"] = dbg
        b = api.batch(blobs, pkg.default_parameters(jpeg_quality=80))
        b.run()
        tms = [b.run() for _ in range(3)]
        pick = {k: round(sum(x.kernel_ms[names.index(k)] for x in tms) / len(tms), 2) for k in ("trellis_stats", "k_trellis_ac", "k_trellis_dc")}
        print(f"sort={sort} debug={dbg} files={n}", pick, flush=True)
        b.close()
