"""Regenerates tests/golden/oracle_digests.json: sha256 of what the ORACLE produces for named synthetic inputs on the rows that have no
external byte truth in this container (PNG recode, lossy PNG, the minimal VP8 encoder, conversions, PNG resize).

This is a tripwire, not a pin: the device is compared with the oracle everywhere else, so an accidental change of the oracle would move
both sides together and go unseen -- these digests make such a change show up as a diff of this file.  What the oracle itself is pinned
to: libjpeg-turbo (JPEG), libpng / zlib on the decode side (PNG), libwebp's decoder (WebP); see DESIGN.md section 2.
Run here:  python tests/golden/make_oracle_digests.py      (inputs come from tools/gen_synth.py and Pillow's PNG writer; their versions
are recorded, a different Pillow / zlib may legitimately change the lossless-PNG digests through the "not smaller" rule)
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]


def digests():
    import _util as U
    from gen_synth import synth_jpeg
    from oracle import oracle as O
    out = {}

    def put(key, data):
        out[key] = hashlib.sha256(data).hexdigest()[:32]
    cases = dict(U.png_cases())
    for name in ("RGB_97x61", "RGBA_97x61", "L_97x61", "P_97x61", "1_97x61", "I;16_97x61", "RGB_200x150_3chunks", "reduce_rgba_grey_opaque", "palette_rgba_translucent",
                 "greydepth_16_levels", "adam7_RGB_33x21"):
        src = cases[name]
        for level in (1, 3, 6):
            put(f"png_optimize/{name}/o{level}", O.png_optimize(src, level)[0])
        put(f"png_lossy/{name}", O.png_lossy(src, 3))
        try:
            put(f"png_to_webp/{name}/q80", O.png_to_webp(src, 80))
        except O.PngError as e:
            out[f"png_to_webp/{name}/q80"] = f"refused {e.code}"
        put(f"png_to_jpeg/{name}/q80", U.oracle_png_to_jpeg(src, 80))
        try:
            put(f"png_resized/{name}/w40", U.oracle_png_resized(src, True, 2, 40, 0))
        except O.PngError as e:
            out[f"png_resized/{name}/w40"] = f"refused {e.code}"
    jpegs = {"420_160x96": synth_jpeg(1, 160, 96, texture=10), "444_97x61": synth_jpeg(2, 97, 61, subsampling=0, texture=5), "prog_104x72": synth_jpeg(5, 104, 72, progressive=True, texture=6)}
    for name, src in jpegs.items():
        for q in (30, 85):
            put(f"jpeg_to_webp/{name}/q{q}", U.oracle_jpeg_to_webp(src, q))
        put(f"jpeg_to_webp/{name}/q85/w60", U.oracle_jpeg_to_webp(src, 85, 60, 0))
        put(f"jpeg_to_png/{name}/lossless", U.oracle_jpeg_to_png(src, True, 3))
        put(f"jpeg_to_png/{name}/quantised", U.oracle_jpeg_to_png(src, False, 3))
    return out


if __name__ == "__main__":
    import PIL
    import zlib
    doc = {"made_with": {"pillow": PIL.__version__, "zlib": zlib.ZLIB_RUNTIME_VERSION}, "digests": digests()}
    json.dump(doc, open(os.path.join(HERE, "oracle_digests.json"), "w"), indent=1, sort_keys=True)
    print(len(doc["digests"]), "digests written")
