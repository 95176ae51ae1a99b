"""Deterministic synthetic photo-like images.

image i uses numpy PCG64(seed = 0xCAE50000 + i): per channel a base level + linear gradient +
6 sinusoids + shared luma noise + 8 random solid rectangles, clipped to u8.
"""
import io

import numpy as np


def synth_rgb(i, w=1920, h=1080, texture=0.0):
    rng = np.random.Generator(np.random.PCG64(0xCAE50000 + i))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    noise = rng.normal(0.0, 6.0, (h, w)).astype(np.float32)
    img = np.empty((h, w, 3), dtype=np.float32)
    for c in range(3):
        base = rng.uniform(40, 200)
        sx, sy = rng.uniform(-0.05, 0.05, 2)
        ch = base + sx * xx + sy * yy
        for _ in range(6):
            fx, fy = rng.uniform(0.002, 0.08, 2)
            ph = rng.uniform(0, 2 * np.pi)
            ch = ch + rng.uniform(5, 40) * np.sin(fx * xx + fy * yy + ph)
        img[:, :, c] = ch + noise
    for _ in range(8):
        rw = int(rng.integers(40, 401)); rh = int(rng.integers(40, 301))
        rw = min(rw, w); rh = min(rh, h)
        x0 = int(rng.integers(0, max(1, w - rw + 1))); y0 = int(rng.integers(0, max(1, h - rh + 1)))
        img[y0:y0 + rh, x0:x0 + rw, :] = rng.uniform(0, 255, 3).astype(np.float32)
    if texture:
        # extra high-frequency content (not part of the SURVEY 8d recipe): exercises ZRL / long codes
        img += rng.normal(0.0, texture, (h, w, 3)).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_jpeg(i, w=1920, h=1080, quality=92, subsampling=2, progressive=False, optimize=False, restart_rows=0, texture=0.0):
    """Source JPEG as config 2 uses: Pillow/libjpeg-turbo, q92, 4:2:0, baseline, no optimise."""
    from PIL import Image
    b = io.BytesIO()
    kw = {}
    if restart_rows:
        kw["restart_marker_rows"] = restart_rows
    Image.fromarray(synth_rgb(i, w, h, texture), "RGB").save(b, format="JPEG", quality=quality, subsampling=subsampling,
                                                    progressive=progressive, optimize=optimize, **kw)
    return b.getvalue()


def synth_png(seed, width, height, mode="RGB", compress_level=6, texture=3.0, **save_kw):
    """PNG of the SURVEY 8d synthetic picture in a Pillow mode ("RGB", "RGBA", "L", "LA", "P", "1", "I;16"), written by
    Pillow/libpng (adaptive filtering, zlib `compress_level`)."""
    import io

    from PIL import Image
    # cut out of a larger picture: the recipe's solid rectangles (40-400 px) would otherwise cover a small one completely
    im = Image.fromarray(np.ascontiguousarray(synth_rgb(seed, width + 400, height + 300, texture=texture)[150:150 + height, 200:200 + width]), "RGB")
    if mode == "I;16":
        rng = np.random.default_rng(seed)
        a = np.asarray(im.convert("L")).astype(np.uint16) * 256 + rng.integers(0, 256, (height, width), dtype=np.uint16)
        im = Image.frombytes("I;16", (width, height), a.astype("<u2").tobytes())
    elif mode == "RGBA":
        im = im.convert("RGBA")
        a = np.asarray(im).copy()
        a[height // 4: height // 2, width // 4: width // 2, 3] = 128
        im = Image.fromarray(a, "RGBA")
    elif mode != "RGB":
        im = im.convert(mode)
    b = io.BytesIO()
    im.save(b, "PNG", compress_level=compress_level, **save_kw)
    return b.getvalue()
