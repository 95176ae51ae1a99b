"""sha256 of the ORACLE's -o3 output for the three 3840x2160 files of tests/test_png_gpu.py::test_full_size_batch_by_properties (configs[2]'s
real size): the oracle needs about a minute per file on one core, too long for the test run, so its answer is committed here and the
device's files are compared with it byte for byte (by digest).  Like oracle_digests.json this pins the device to the oracle at that
size, not the oracle to oxipng.   Run here: python tests/golden/make_oracle_digests_4k.py"""
import hashlib
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]


def one(k):
    from _util import oracle_png
    from gen_synth import synth_png
    src = synth_png(40 + k, 3840, 2160, "RGB", texture=float(k))
    out = oracle_png(src, 3)
    return f"png_optimize/synth_png({40 + k},3840,2160,RGB,texture={float(k)})/o3", {"in_sha256": hashlib.sha256(src).hexdigest(), "in_bytes": len(src),
                                                                                      "out_sha256": hashlib.sha256(out).hexdigest(), "out_bytes": len(out)}


if __name__ == "__main__":
    import PIL
    import zlib
    with ProcessPoolExecutor(3) as ex:
        res = dict(ex.map(one, range(3)))
    json.dump({"made_with": {"pillow": PIL.__version__, "zlib": zlib.ZLIB_RUNTIME_VERSION}, "digests": res}, open(os.path.join(HERE, "oracle_digests_4k.json"), "w"), indent=1, sort_keys=True)
    print(res)
