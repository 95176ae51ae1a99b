"""damaged files through the product library on the GPU vs the oracle; usage: gpu_fuzz.py <first seed> <last seed> [size scale]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, ROOT)
import test_pipeline_emul as E
from _util import product_api, oracle_lossy, oracle_lossless, package
api = product_api()
bad = total = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    for lossless in (True, False):
        for whole in (False, True):
            blobs = E.fuzzed_blobs(seed, 64, whole, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
            outs = api.batch_compress(blobs, package().default_parameters(jpeg_optimize=lossless))
            for i, (src, out) in enumerate(zip(blobs, outs)):
                total += 1
                try:
                    want = oracle_lossless(src) if lossless else oracle_lossy(src)
                except Exception as e:
                    want = e
                if isinstance(want, Exception) != isinstance(out, Exception) or (not isinstance(out, Exception) and out != want):
                    bad += 1; print("MISMATCH seed", seed, "lossless", lossless, "whole", whole, "case", i, "kind", i % 4, str(out)[:60] if isinstance(out, Exception) else len(out), flush=True)
print("total", total, "bad", bad)
