"""How far the device's DEFLATE parse is from libdeflate's (what oxipng -o3 links, level 11 / 12): the IDAT stream of this repo's -o3 output (the oracle's
bytes = the device's) is inflated and packed again by the system libdeflate at levels 6, 9, 11, 12 and by zlib 6 / 9 -- same filtered bytes, so only the
parse differs.  `python tools/png_parse_gap.py [width height] [--zopfli]`"""
import ctypes as C
import ctypes.util
import io
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
from oracle import oracle as O        # noqa: E402
from gen_synth import synth_png        # noqa: E402

D = C.CDLL(ctypes.util.find_library("deflate"))
D.libdeflate_alloc_compressor.restype = C.c_void_p
D.libdeflate_zlib_compress.restype = C.c_size_t
D.libdeflate_zlib_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
D.libdeflate_zlib_compress_bound.restype = C.c_size_t
D.libdeflate_zlib_compress_bound.argtypes = [C.c_void_p, C.c_size_t]


def libdeflate(raw, level):
    c = D.libdeflate_alloc_compressor(level)
    cap = D.libdeflate_zlib_compress_bound(c, len(raw))
    out = C.create_string_buffer(cap)
    n = D.libdeflate_zlib_compress(c, raw, len(raw), out, cap)
    D.libdeflate_free_compressor(C.c_void_p(c))
    return n


def idat(png):
    at, out = 8, b""
    while at < len(png):
        n = int.from_bytes(png[at:at + 4], "big")
        if png[at + 4:at + 8] == b"IDAT":
            out += png[at + 8:at + 8 + n]
        at += 12 + n
    return out


nums = [a for a in sys.argv[1:] if a.isdigit()]
w, h = (int(nums[0]), int(nums[1])) if len(nums) > 1 else (1280, 720)
for k, (mode, tex) in enumerate([("RGB", 3.0), ("RGB", 0.5), ("RGBA", 3.0), ("L", 3.0)]):
    src = synth_png(40 + k, w, h, mode, texture=tex)
    out, _ = O.png_optimize(src, 3)
    z = idat(out)
    raw = zlib.decompress(z)
    line = ["%s tex %.1f: in %d out %d | idat %d" % (mode, tex, len(src), len(out), len(z))]
    line.append("zlib6 %.3f zlib9 %.3f" % (len(zlib.compress(raw, 6)) / len(z), len(zlib.compress(raw, 9)) / len(z)))
    line.append(" ".join("ld%d %.3f" % (lv, libdeflate(raw, lv) / len(z)) for lv in (6, 9, 11, 12)))
    zopfli = next((c for c in ("/opt/conda/bin/zopfli", "/usr/bin/zopfli") if os.path.exists(c)), None)
    if zopfli and "--zopfli" in sys.argv:   # what --zopfli (refused by this build) would have bought: slow, so only on request
        import subprocess
        import tempfile
        f = os.path.join(tempfile.mkdtemp(), "raw.bin")
        open(f, "wb").write(raw)
        subprocess.run([zopfli, "--zlib", "--i15", f], check=True)
        line.append("zopfli-i15 %.3f" % (os.path.getsize(f + ".zlib") / len(z)))
    print(" | ".join(line))
