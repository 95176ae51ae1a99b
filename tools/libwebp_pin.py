"""libwebp itself (every libwebp this container carries, through ctypes) against oracle/vp8enc_oracle.c, stage by stage: header fields, segment map,
modes, levels, bytes.  `python tools/libwebp_pin.py [n_pictures] [width height] [quality]`"""
import ctypes as C
import glob
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
from gen_synth import synth_rgb   # noqa: E402
from oracle import oracle as O    # noqa: E402

CANDIDATES = ["/usr/lib/x86_64-linux-gnu/libwebp.so.7", "/opt/conda/lib/libwebp.so.7"] + sorted(glob.glob("/usr/local/lib/python3*/dist-packages/pillow.libs/libwebp-*.so*"))


def libwebps():
    """[(version string, CDLL)] of every loadable libwebp with the encoder API"""
    out = []
    for dep in glob.glob("/usr/local/lib/python3*/dist-packages/pillow.libs/libsharpyuv*"):
        try:
            C.CDLL(dep, mode=C.RTLD_GLOBAL)
        except OSError:
            pass
    for path in CANDIDATES:
        try:
            W = C.CDLL(path)
        except OSError:
            continue
        if not hasattr(W, "WebPEncode"):
            continue
        W.WebPGetEncoderVersion.restype = C.c_int
        v = W.WebPGetEncoderVersion()
        out.append(("%d.%d.%d" % (v >> 16, (v >> 8) & 255, v & 255), W))
    return out


def libwebp_encode(W, rgb, quality, use_argb=False, **cfg_fields):
    """WebPEncode with a default WebPConfig at `quality` (what crate webp 0.3.1's Encoder::encode does; use_argb=True is its picture set-up)"""
    h, w, _ = rgb.shape
    cfg = (C.c_int32 * 64)()
    W.WebPConfigInitInternal.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int]
    assert W.WebPConfigInitInternal(cfg, 0, float(quality), 0x020f)
    for k, v in cfg_fields.items():
        cfg[int(k)] = v
    pic = (C.c_uint8 * 1024)()
    assert W.WebPPictureInitInternal(pic, 0x020f)
    ints = C.cast(pic, C.POINTER(C.c_int32))
    ints[0], ints[2], ints[3] = int(use_argb), w, h
    rgb = np.ascontiguousarray(rgb)
    W.WebPPictureImportRGB.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert W.WebPPictureImportRGB(pic, rgb.ctypes.data, w * 3)
    wr = (C.c_uint8 * 64)()
    W.WebPMemoryWriterInit.argtypes = [C.c_void_p]
    W.WebPMemoryWriterInit(wr)
    ptrs = C.cast(pic, C.POINTER(C.c_void_p))
    ptrs[12] = C.cast(W.WebPMemoryWrite, C.c_void_p).value
    ptrs[13] = C.addressof(wr)
    W.WebPEncode.argtypes = [C.c_void_p, C.c_void_p]
    assert W.WebPEncode(cfg, pic)
    data = C.string_at(C.cast(wr, C.POINTER(C.c_void_p))[0], C.cast(wr, C.POINTER(C.c_size_t))[1])
    W.WebPPictureFree(pic)
    return data


def compare(ref, mine, verbose=True):
    """-> dict of stage -> bool; prints the first differences"""
    fr, mr = O.vp8_parse(ref)
    fm, mm = O.vp8_parse(mine)
    res = {}
    res["header"] = fr.header() == fm.header()
    if verbose and not res["header"]:
        print("  header ref ", fr.header())
        print("  header mine", fm.header())
    for k in ("segment", "is_i4", "ymode", "uvmode", "bmodes"):
        eq = mr[k] == mm[k]
        if k == "ymode":
            eq = eq | (mr["is_i4"] == 1)
        res[k] = bool(np.all(eq))
        if verbose and not res[k]:
            bad = np.argwhere(~(eq if eq.ndim == 1 else eq.all(axis=1)))[:, 0]
            print("  %s: %d of %d macroblocks differ, first at %d (x %d, y %d): ref %s mine %s" % (k, len(bad), len(mr), bad[0], bad[0] % fr.mbw, bad[0] // fr.mbw, mr[k][bad[0]], mm[k][bad[0]]))
    eq = (mr["levels"] == mm["levels"]).all(axis=(1, 2))
    res["levels"] = bool(eq.all())
    if verbose and not res["levels"]:
        bad = np.argwhere(~eq)[:, 0]
        print("  levels: %d of %d macroblocks differ, first at %d (x %d, y %d)" % (len(bad), len(mr), bad[0], bad[0] % fr.mbw, bad[0] // fr.mbw))
    res["probas"] = bytes(fr.probas) == bytes(fm.probas)
    res["bytes"] = ref == mine
    return res


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1500, 844)
    q = float(sys.argv[4]) if len(sys.argv) > 4 else 85.0
    libs = libwebps()
    print("libwebp versions:", [v for v, _ in libs])
    same = 0
    for seed in range(n):
        rgb = np.ascontiguousarray(synth_rgb(seed, w, h))
        refs = [libwebp_encode(W, rgb, q) for _, W in libs]
        assert all(r == refs[0] for r in refs), "libwebp versions disagree"
        assert libwebp_encode(libs[0][1], rgb, q, use_argb=True) == refs[0], "use_argb import differs"
        mine = O.vp8enc_encode_rgb(rgb, q)
        res = compare(refs[0], mine)
        same += res["bytes"]
        print("seed %d %dx%d q%g: libwebp %d B (md5 %s), oracle %d B: %s" % (seed, w, h, q, len(refs[0]), hashlib.md5(refs[0]).hexdigest()[:8], len(mine), " ".join("%s=%s" % (k, "ok" if v else "DIFF") for k, v in res.items())))
    print("%d of %d byte-identical" % (same, n))


if __name__ == "__main__":
    main()
