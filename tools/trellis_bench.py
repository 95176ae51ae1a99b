"""Device time of the JPEG path under each profile (plain / scalar = scan search over the scalar quantiser / mozjpeg = scan search + trellis + deringing, the default), per kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from _util import package
from bench import make_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
uniq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
profiles = sys.argv[3].split(",") if len(sys.argv) > 3 else ["scalar", "mozjpeg"]
pkg = package()
api = pkg.load()
blobs = make_inputs(0, uniq)
blobs = [blobs[i % uniq] for i in range(n)]
for prof in profiles:
    if prof:
        os.environ["CSH_PROFILE"] = prof
    else:
        os.environ.pop("CSH_PROFILE", None)
    b = api.batch(blobs, pkg.default_parameters(jpeg_quality=80))
    b.run()
    tms = [b.run() for _ in range(3)]
    t = tms[-1]
    names = api.kernel_names()
    ms = sum(x.total_ms for x in tms) / len(tms)
    print(f"profile={prof or 'default'} files={n} ms={ms:.2f} GP/s={n * 2.0736 / ms:.2f} out_bytes={t.out_bytes}",
          {names[i]: round(sum(x.kernel_ms[i] for x in tms) / len(tms), 2) for i in range(len(names)) if names[i] and t.kernel_ms[i] > 0.05}, flush=True)
    b.close()
