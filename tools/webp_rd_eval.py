"""Lossy WebP oracle against libwebp (Pillow, method 4) at EQUAL PSNR: bytes of oracle/webp_oracle.c at q 50/75/85/92 over libwebp's bytes interpolated (log-linear over a
quality sweep) at the same RGB PSNR.  `python tools/webp_rd_eval.py` -- the reference sample photographs + two synthetic 1500 x 844 pictures."""
import io, os, sys, numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/tools"]
from PIL import Image
from oracle import oracle as O
from gen_synth import synth_rgb
imgs = []
for name in ("j0.JPG", "j1.jpg", "p0.png", "p2.png"):
    p = os.path.join("/root/repo/tests/golden/reference_samples", name)
    if os.path.exists(p):
        im = Image.open(p).convert("RGB"); w, h = im.size
        im = im.resize((1500, max(1, round(h * 1500 / w))), Image.LANCZOS) if w > 1500 else im
        imgs.append((name, np.ascontiguousarray(np.asarray(im))))
for k in range(2):
    imgs.append(("synth%d" % k, np.ascontiguousarray(synth_rgb(k, 1500, 844))))
def psnr(data, rgb):
    a = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(np.float64)
    return 10 * np.log10(255.0 ** 2 / ((a - rgb) ** 2).mean())
tot = []
for name, rgb in imgs:
    ref = []
    for q in list(range(2, 100, 4)) + [99, 100]:
        b = io.BytesIO(); Image.fromarray(rgb).save(b, "WEBP", quality=q, method=4); ref.append((psnr(b.getvalue(), rgb), len(b.getvalue())))
    ref.sort()
    rp = np.array([r[0] for r in ref]); rb = np.log(np.array([r[1] for r in ref], dtype=np.float64))
    line = []
    for q in (50, 75, 85, 92):
        d = O.webp_encode_rgb(rgb, q); ps = psnr(d, rgb)
        lib_at = float(np.exp(np.interp(ps, rp, rb)))
        line.append("q%d %dB %.2fdB x%.3f" % (q, len(d), ps, len(d) / lib_at)); tot.append(len(d) / lib_at)
    print(name, rgb.shape, " | ".join(line))
print("geomean ratio at equal PSNR: %.4f" % float(np.exp(np.mean(np.log(tot)))))
