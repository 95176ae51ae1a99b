import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _util import package, product_api, oracle_jpeg_to_png
from test_pipeline_emul import fuzzed_blobs
api = product_api(); PNG = 1
from test_jpeg_png_emul import PNG as PNGT
blobs = fuzzed_blobs(13, 20, True)
for mode in (None, "1", "0"):
    if mode is None: os.environ.pop("CSH_PROG_PAR", None)
    else: os.environ["CSH_PROG_PAR"] = mode
    outs = api.batch_convert(blobs, package().default_parameters(png_optimize=True, png_optimization_level=1), PNGT)
    bad = []
    for i, (b, o) in enumerate(zip(blobs, outs)):
        try: want = oracle_jpeg_to_png(b, True, 1)
        except Exception as ex:
            want = None
            if mode is None and i < 3: print('oracle raised', i, repr(ex)[:200])
        if want is None:
            if not isinstance(o, Exception): bad.append((i, "should fail"))
        elif isinstance(o, Exception): bad.append((i, 'device failed', repr(o)[:80]))
        elif o != want: bad.append((i, "differs", i % 4, len(o), len(want)))
    print("mode", mode, "bad", bad, flush=True)
