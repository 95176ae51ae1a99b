"""JPEG -> WebP on the device against the oracle, stage by stage (header, segments, modes, levels): `python tools/vp8_device_diff.py [emul]`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
from _util import emul_api, oracle_jpeg_to_webp, package, product_api   # noqa: E402
from gen_synth import synth_jpeg                                          # noqa: E402
from libwebp_pin import compare                                           # noqa: E402

api = emul_api() if len(sys.argv) > 1 and sys.argv[1] == "emul" else product_api()
pkg = package()
cases = [("420_160x96", synth_jpeg(1, 160, 96, texture=10)), ("flat_64x48", synth_jpeg(3, 64, 48)), ("tiny_9x5", synth_jpeg(6, 9, 5, texture=3)), ("640x480", synth_jpeg(21, 640, 480, texture=20)),
         ("1500x844", synth_jpeg(23, 1500, 844))]
for q in (85, 40):
    p = pkg.default_parameters(webp_quality=q, jpeg_quality=q)
    outs = api.batch_convert([c[1] for c in cases], p, 3)
    for (name, src), out in zip(cases, outs):
        want = oracle_jpeg_to_webp(src, q)
        if isinstance(out, Exception):
            print(q, name, "ERROR", out)
            continue
        print(q, name, len(want), len(out), "identical" if out == want else "DIFFERENT")
        if out != want:
            try:
                print("   ", compare(want, out))
            except Exception as e:   # an unparsable stream
                print("    unparsable:", e)
