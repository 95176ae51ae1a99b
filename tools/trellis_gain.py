"""Bytes at equal PSNR: mozjpeg's trellis quantiser (CSH_PROFILE=mozjpeg-trellis / mozjpeg) against the scalar quantiser (CSH_PROFILE=scalar),
on the bench's synthetic 1080p set, made on the device (the outputs are byte-identical to the oracle's: tests/test_trellis_gpu.py).

For every image: the profile's file at -q 80 (bytes, PSNR against the decoded source), and the scalar quantiser's files at -q 60..80; the scalar
curve is interpolated at the profile's PSNR.  mozjpeg's published gain for trellis quantisation is several per cent at equal quality; a
restatement that saved < 2 % would be wrong.   usage: python tools/trellis_gain.py [images=256] > profiles/r03_trellis_gain.txt"""
import io
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from PIL import Image

from _util import package
from bench import make_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pkg = package()
api = pkg.load()
srcs = make_inputs(0, n)
pool = ThreadPoolExecutor(os.cpu_count() or 8)
ref = list(pool.map(lambda s: np.asarray(Image.open(io.BytesIO(s)).convert("RGB")).astype(np.float32), srcs))


def psnr_all(outs):
    def one(i):
        d = np.asarray(Image.open(io.BytesIO(outs[i])).convert("RGB")).astype(np.float32) - ref[i]
        return 10 * np.log10(255.0 ** 2 / float(np.mean(d * d)))
    return np.array(list(pool.map(one, range(n))))


def run(profile, qualities):
    os.environ["CSH_PROFILE"] = profile or "scalar"   # (unset = the library's default = the whole mozjpeg profile)
    b = api.batch(srcs, pkg.default_parameters(jpeg_quality=80))
    b.retain_dct()
    b.run()
    res = {}
    for q in qualities:
        b.set_quality([q] * n)
        b.rerun_encode()
        outs = b.fetch()
        res[q] = (np.array([len(o) for o in outs]), psnr_all(outs))
    b.close()
    return res


scalar = run("", list(range(60, 81, 2)))
qs = sorted(scalar)
print(f"# {n} synthetic 1080p images (SURVEY 8d recipe, q92 4:2:0 sources), PSNR of the decoded RGB against the decoded source; device outputs (== oracle)")
print(f"# scalar quantiser (CSH_PROFILE=scalar) at -q 80: {scalar[80][0].mean() / 1e3:.1f} KB, {scalar[80][1].mean():.3f} dB")
print("profile              bytes@q80(KB)  PSNR(dB)  vs scalar@q80  scalar bytes at equal PSNR(KB)  gain at equal PSNR   (per-image gain: min / median / max)")
for prof in ("mozjpeg-trellis", "mozjpeg-dering", "mozjpeg"):
    by, ps = run(prof, [80])[80]
    eq = np.empty(n)
    for i in range(n):
        xs = np.array([scalar[q][1][i] for q in qs]); ys = np.array([scalar[q][0][i] for q in qs], dtype=np.float64)
        o = np.argsort(xs)
        eq[i] = np.interp(ps[i], xs[o], ys[o])
    gain = 1.0 - by / eq
    print(f"{prof:20s} {by.mean() / 1e3:10.1f} {ps.mean():10.3f} {100 * (by.mean() / scalar[80][0].mean() - 1):+10.2f} % {eq.mean() / 1e3:22.1f} {100 * (1 - by.sum() / eq.sum()):18.2f} %"
          f"   ({100 * gain.min():.2f} / {100 * np.median(gain):.2f} / {100 * gain.max():.2f} %)")
