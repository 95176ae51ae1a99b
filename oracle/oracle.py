"""ctypes binding of the CPU ORACLE (oracle/libcsoracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product path (caesium-clt_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcsoracle.so")

ZZ = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
               41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)


class Comp(C.Structure):
    _fields_ = [("id", C.c_int), ("h", C.c_int), ("v", C.c_int), ("tq", C.c_int),
                ("comp_w", C.c_int), ("comp_h", C.c_int), ("real_bw", C.c_int), ("real_bh", C.c_int),
                ("bw", C.c_int), ("bh", C.c_int), ("coef", C.POINTER(C.c_int16))]


class Scan(C.Structure):
    _fields_ = [("ncomp_in_scan", C.c_int), ("comp_idx", C.c_int * 4),
                ("Ss", C.c_int), ("Se", C.c_int), ("Ah", C.c_int), ("Al", C.c_int)]

    def astuple(self):
        return (tuple(self.comp_idx[i] for i in range(self.ncomp_in_scan)), self.Ss, self.Se, self.Ah, self.Al)


class Image(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("ncomp", C.c_int), ("precision", C.c_int),
                ("progressive", C.c_int), ("hmax", C.c_int), ("vmax", C.c_int), ("mcus_x", C.c_int), ("mcus_y", C.c_int),
                ("restart_interval", C.c_int), ("comp", Comp * 4), ("qt", (C.c_uint16 * 64) * 4), ("qt_present", C.c_int * 4),
                ("nscans", C.c_int), ("scans", Scan * 64), ("meta", C.POINTER(C.c_uint8)), ("meta_len", C.c_size_t),
                ("saw_jfif", C.c_int), ("adobe_transform", C.c_int)]


class EncParams(C.Structure):
    _fields_ = [("quality", C.c_int), ("progressive", C.c_int), ("subsampling", C.c_int), ("qtable_profile", C.c_int),
                ("marker_style", C.c_int), ("scan_script", C.c_int), ("keep_metadata", C.c_int), ("force_baseline", C.c_int), ("preserve_icc", C.c_int),
                ("trellis", C.c_int), ("deringing", C.c_int)]


class Vp8Mb(C.Structure):
    _fields_ = [("segment", C.c_uint8), ("is_i4", C.c_uint8), ("ymode", C.c_uint8), ("uvmode", C.c_uint8), ("bmodes", C.c_uint8 * 16),
                ("skip", C.c_uint8), ("alpha", C.c_uint8), ("pad", C.c_uint8 * 2), ("levels", (C.c_int16 * 16) * 25)]


class Vp8Frame(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("mbw", C.c_int), ("mbh", C.c_int),
                ("num_segments", C.c_int), ("update_map", C.c_int), ("seg_quant", C.c_int * 4), ("seg_filter", C.c_int * 4), ("seg_probs", C.c_int * 3),
                ("filter_simple", C.c_int), ("filter_level", C.c_int), ("filter_sharpness", C.c_int),
                ("num_parts_log2", C.c_int), ("base_quant", C.c_int), ("dq", C.c_int * 5), ("use_skip", C.c_int), ("skip_proba", C.c_int),
                ("probas", C.c_uint8 * 1056), ("part0_size", C.c_size_t), ("vp8_size", C.c_size_t),
                ("alpha_avg", C.c_int), ("uv_alpha_avg", C.c_int), ("seg_alpha", C.c_int * 4), ("seg_beta", C.c_int * 4), ("seg_max_edge", C.c_int * 4)]

    def header(self):
        return dict(size=(self.width, self.height), nseg=self.num_segments, update_map=self.update_map, quant=list(self.seg_quant), filt=list(self.seg_filter),
                    seg_probs=list(self.seg_probs), filter=(self.filter_simple, self.filter_level, self.filter_sharpness), parts=self.num_parts_log2,
                    base_quant=self.base_quant, dq=list(self.dq), skip=(self.use_skip, self.skip_proba))


class Png(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("depth", C.c_int), ("ctype", C.c_int), ("interlace", C.c_int),
                ("channels", C.c_int), ("bpp", C.c_int), ("nplte", C.c_int), ("rowbytes", C.c_size_t),
                ("pix", C.POINTER(C.c_uint8)), ("chunks", C.POINTER(C.c_uint8)), ("chunks_len", C.c_size_t), ("idat_at", C.c_size_t), ("no_reduce", C.c_int), ("pal_tied", C.c_int)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("jpeg_oracle.c", "jpeg_oracle.h", "png_oracle.c", "png_oracle.h", "webp_oracle.c", "webp_oracle.h", "vp8enc_oracle.c", "vp8enc_oracle.h")] + [os.path.join(_HERE, "..", "include", f) for f in ("png_quality_table.h", "vp8_tables.h", "vp8_cost_tables.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.cso_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(Image))]
        L.cso_image_free.argtypes = [C.POINTER(Image)]
        L.cso_decode_pixels.argtypes = [C.POINTER(Image), C.c_void_p]
        L.cso_decode_plane.argtypes = [C.POINTER(Image), C.c_int, C.c_void_p]
        L.cso_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(EncParams), C.c_void_p, C.POINTER(C.POINTER(Image))]
        L.cso_encode.argtypes = [C.POINTER(Image), C.POINTER(EncParams), C.POINTER(Scan), C.c_int,
                                 C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_jpeg_compress.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(EncParams), C.c_int,
                                        C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_free.argtypes = [C.c_void_p]
        L.cso_quality_tables.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.cso_fdct_islow.argtypes = [C.c_void_p, C.c_void_p]
        L.cso_idct_islow.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cso_stock_script.argtypes = [C.c_int, C.c_int, C.POINTER(Scan)]
        L.cso_gen_optimal_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cso_compute_dimensions.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.cso_lanczos3_resize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.cso_lanczos3_resize16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.cso_ycc_to_rgb.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.cso_rgb_to_ycc.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.cso_pixels_to_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(EncParams), C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_jpeg_compress_resized.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(EncParams), C.c_int, C.c_int,
                                                C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_last_error.restype = C.c_char_p
        L.cso_crc32.restype = C.c_uint32
        L.cso_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.cso_adler32.restype = C.c_uint32
        L.cso_adler32.argtypes = [C.c_char_p, C.c_size_t]
        L.cso_inflate_zlib.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.cso_png_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(Png))]
        L.cso_png_free.argtypes = [C.POINTER(Png)]
        L.cso_png_reduce.argtypes = [C.POINTER(Png)]
        L.cso_png_quantize.argtypes = [C.POINTER(Png), C.c_int]
        L.cso_png_lossy.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_png_to_webp.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_png_scores.argtypes = [C.POINTER(Png), C.c_void_p]
        L.cso_png_filter.argtypes = [C.POINTER(Png), C.c_int, C.c_void_p, C.c_void_p]
        L.cso_deflate_zlib.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_png_trials.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.cso_png_optimize.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.cso_png_optimize_zopfli.argtypes = L.cso_png_optimize.argtypes
        L.cso_png_optimize_iters.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_deflate_zlib_iters.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_png_deep_div.argtypes = [C.c_int]
        L.cso_png_deep_div.restype = None
        L.cso_vp8enc_encode_yuv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p]
        L.cso_vp8enc_encode_rgb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_vp8_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        L.cso_webp_rgb_to_yuv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cso_webp_encode_rgb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.cso_vp8l_encode.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise OracleError(lib().cso_last_error().decode())


def params(quality=80, progressive=1, subsampling=0, qtable_profile=3, marker_style=1, scan_script=0,
           keep_metadata=0, force_baseline=0, preserve_icc=1, trellis=0, deringing=0):
    return EncParams(quality, progressive, subsampling, qtable_profile, marker_style, scan_script, keep_metadata, force_baseline, preserve_icc, trellis, deringing)


class CoefImage:
    """Owns a cso_image*: quantised coefficients per component (natural order)."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().cso_image_free(self.ptr)
            self.ptr = None

    @property
    def im(self):
        return self.ptr.contents

    def coefs(self, ci):
        """[bh][bw][64] int16 natural order (copy)."""
        k = self.im.comp[ci]
        n = k.bw * k.bh * 64
        return np.ctypeslib.as_array(k.coef, shape=(n,)).reshape(k.bh, k.bw, 64).copy()

    def coefs_view(self, ci):
        """writable [bh][bw][64] view (natural order) of the oracle's own buffer -- for crafting test streams"""
        k = self.im.comp[ci]
        return np.ctypeslib.as_array(k.coef, shape=(k.bw * k.bh * 64,)).reshape(k.bh, k.bw, 64)

    def coefs_zigzag(self, ci):
        return self.coefs(ci)[:, :, ZZ]

    def qtable(self, tq):
        return np.array(self.im.qt[tq], dtype=np.uint16)

    def scans(self):
        return [self.im.scans[i].astuple() for i in range(self.im.nscans)]

    def pixels(self):
        im = self.im
        out = np.empty((im.height, im.width, im.ncomp), dtype=np.uint8)
        _check(lib().cso_decode_pixels(self.ptr, out.ctypes.data))
        return out

    def plane(self, ci):
        k = self.im.comp[ci]
        out = np.empty((k.comp_h, k.comp_w), dtype=np.uint8)
        _check(lib().cso_decode_plane(self.ptr, ci, out.ctypes.data))
        return out

    def encode(self, p, script=None):
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        if script is None:
            _check(lib().cso_encode(self.ptr, C.byref(p), None, 0, C.byref(out), C.byref(n)))
        else:
            arr = (Scan * len(script))()
            for i, (comps, ss, se, ah, al) in enumerate(script):
                arr[i].ncomp_in_scan = len(comps)
                for j, c in enumerate(comps):
                    arr[i].comp_idx[j] = c
                arr[i].Ss, arr[i].Se, arr[i].Ah, arr[i].Al = ss, se, ah, al
            _check(lib().cso_encode(self.ptr, C.byref(p), arr, len(script), C.byref(out), C.byref(n)))
        data = C.string_at(out, n.value)
        lib().cso_free(out)
        return data


def decode(data):
    ptr = C.POINTER(Image)()
    _check(lib().cso_decode(data, len(data), C.byref(ptr)))
    return CoefImage(ptr)


def forward(pix, p, qtables=None):
    """pix: HxWxC u8 in the JPEG colour space -> CoefImage (new quantised coefficients)."""
    pix = np.ascontiguousarray(pix, dtype=np.uint8)
    if pix.ndim == 2:
        pix = pix[:, :, None]
    h, w, c = pix.shape
    qt = None
    if qtables is not None:
        qt = np.ascontiguousarray(qtables, dtype=np.uint16).reshape(2, 64)
    ptr = C.POINTER(Image)()
    _check(lib().cso_forward(pix.ctypes.data, w, h, c, C.byref(p), qt.ctypes.data if qt is not None else None, C.byref(ptr)))
    return CoefImage(ptr)


def jpeg_compress(data, p, lossless=False):
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    _check(lib().cso_jpeg_compress(data, len(data), C.byref(p), 1 if lossless else 0, C.byref(out), C.byref(n)))
    res = C.string_at(out, n.value)
    lib().cso_free(out)
    return res


def quality_tables(q, profile=3, force_baseline=0):
    out = np.zeros((2, 64), dtype=np.uint16)
    lib().cso_quality_tables(q, profile, force_baseline, out.ctypes.data)
    return out


def stock_script(ncomp, which=0):
    arr = (Scan * 64)()
    n = lib().cso_stock_script(ncomp, which, arr)
    return [arr[i].astuple() for i in range(n)]


def gen_optimal_table(freq):
    f = np.zeros(257, dtype=np.int64)
    f[:len(freq)] = freq
    bits = np.zeros(17, dtype=np.uint8)
    hv = np.zeros(256, dtype=np.uint8)
    n = lib().cso_gen_optimal_table(f.ctypes.data, bits.ctypes.data, hv.ctypes.data)
    return bits, hv[:n]


def compute_dimensions(ow, oh, dw, dh):
    nw, nh = C.c_int(), C.c_int()
    lib().cso_compute_dimensions(ow, oh, dw, dh, C.byref(nw), C.byref(nh))
    return nw.value, nh.value


def lanczos3_resize(img, nw, nh):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    out = np.empty((nh, nw, c), dtype=np.uint8)
    lib().cso_lanczos3_resize(img.ctypes.data, w, h, c, nw, nh, out.ctypes.data)
    return out


def ycc_to_rgb(a):
    a = np.ascontiguousarray(a, dtype=np.uint8); out = np.empty_like(a)
    lib().cso_ycc_to_rgb(a.ctypes.data, a.size // 3, out.ctypes.data)
    return out


def rgb_to_ycc(a):
    a = np.ascontiguousarray(a, dtype=np.uint8); out = np.empty_like(a)
    lib().cso_rgb_to_ycc(a.ctypes.data, a.size // 3, out.ctypes.data)
    return out


def jpeg_compress_resized(data, p, width, height):
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    _check(lib().cso_jpeg_compress_resized(data, len(data), C.byref(p), width, height, C.byref(out), C.byref(n)))
    res = C.string_at(out, n.value)
    lib().cso_free(out)
    return res


def pixels_to_jpeg(pix, p, width=0, height=0):
    """(h, w, 1 or 3) uint8 -> JPEG file bytes"""
    pix = np.ascontiguousarray(pix, dtype=np.uint8)
    h, w, nc = pix.shape
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    _check(lib().cso_pixels_to_jpeg(pix.ctypes.data, w, h, nc, C.byref(p), width, height, C.byref(out), C.byref(n)))
    res = C.string_at(out, n.value)
    lib().cso_free(out)
    return res


# ---------------------------------------------------------------- lossless PNG row (png_oracle.c; parity unpinned, see its header)
class PngError(OracleError):
    def __init__(self, code):
        super().__init__("png oracle: code %d" % code)
        self.code = code


class PngImage:
    """Owns a cso_png*: the unfiltered rows and the chunks that are carried over."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().cso_png_free(self.ptr)
            self.ptr = None

    @property
    def im(self):
        return self.ptr.contents

    def rows(self):
        im = self.im
        return np.ctypeslib.as_array(im.pix, shape=(im.height * im.rowbytes,)).reshape(im.height, im.rowbytes).copy()

    def reduce(self):
        """P2 reductions in place; -> bit mask of what was applied"""
        return lib().cso_png_reduce(self.ptr)

    def quantize(self, quality=80):
        """lossy: median cut to at most 256 colours (fewer when the quality allows), in place; -> 16 when applied"""
        return lib().cso_png_quantize(self.ptr, quality)

    def scores(self):
        """[height][5 filters][5 scores: MinSum, Entropy, Bigrams, BigEnt, Brute]"""
        out = np.empty((self.im.height, 5, 5), dtype=np.uint64)
        lib().cso_png_scores(self.ptr, out.ctypes.data)
        return out

    def filtered(self, strategy):
        """(stream of height*(1+rowbytes) bytes, per-row filter choice)"""
        im = self.im
        out = np.empty(im.height * (1 + im.rowbytes), dtype=np.uint8)
        choice = np.empty(im.height, dtype=np.uint8)
        rc = lib().cso_png_filter(self.ptr, strategy, out.ctypes.data, choice.ctypes.data)
        if rc:
            raise PngError(rc)
        return out, choice


def png_decode(data, keep_metadata=False):
    ptr = C.POINTER(Png)()
    rc = lib().cso_png_decode(data, len(data), 1 if keep_metadata else 0, C.byref(ptr))
    if rc:
        raise PngError(rc)
    return PngImage(ptr)


def deflate_zlib(data, iters=None):
    """the coder over one stream; iters: passes of the min-cost-path parse over the chunks that qualify (None: the default, 0: the greedy parse everywhere)"""
    data = bytes(data)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    if iters is None:
        lib().cso_deflate_zlib(data, len(data), C.byref(out), C.byref(n))
    else:
        lib().cso_deflate_zlib_iters(data, len(data), int(iters), C.byref(out), C.byref(n))
    res = C.string_at(out, n.value)
    lib().cso_free(out)
    return res


def inflate_zlib(data, cap):
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().cso_inflate_zlib(data, len(data), out.ctypes.data, cap, C.byref(n))
    if rc:
        raise PngError(rc)
    return out[:n.value].tobytes()


def png_trials(level):
    arr = (C.c_int * 10)()
    n = lib().cso_png_trials(level, arr)
    return [arr[i] for i in range(n)]


def png_optimize(data, level=3, keep_metadata=False, zopfli=False):
    """-> (file bytes, winning strategy or -1 when the input is returned unchanged); zopfli: png.force_zopfli (more passes of the cost model)"""
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    chosen = C.c_int()
    fn = lib().cso_png_optimize_zopfli if zopfli else lib().cso_png_optimize
    rc = fn(data, len(data), level, 1 if keep_metadata else 0, C.byref(out), C.byref(n), C.byref(chosen))
    if rc:
        raise PngError(rc)
    res = C.string_at(out, n.value)
    lib().cso_free(out)
    return res, chosen.value


def png_optimize_iters(data, level, iters, deep_div=0):
    """tools: the lossless recode with `iters` passes of the min-cost-path parse (0: the greedy parse everywhere); deep_div overrides which chunks qualify"""
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    lib().cso_png_deep_div(deep_div)
    rc = lib().cso_png_optimize_iters(data, len(data), level, iters, C.byref(out), C.byref(n))
    lib().cso_png_deep_div(0)
    if rc:
        raise PngError(rc)
    res = C.string_at(out, n.value)
    lib().cso_free(out)
    return res


# ---------------------------------------------------------------- lossy WebP row (webp_oracle.c: libwebp's import; the encoder is vp8enc_oracle.c below)
def webp_rgb_to_yuv(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    y = np.empty((mbh * 16, mbw * 16), np.uint8); u = np.empty((mbh * 8, mbw * 8), np.uint8); v = np.empty_like(u)
    lib().cso_webp_rgb_to_yuv(rgb.ctypes.data, w, h, y.ctypes.data, u.ctypes.data, v.ctypes.data)
    return y, u, v


def webp_encode_rgb(rgb, quality):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t()
    rc = lib().cso_webp_encode_rgb(rgb.ctypes.data, w, h, quality, C.byref(out), C.byref(n))
    if rc:
        raise OracleError("webp oracle: %d" % rc)
    data = C.string_at(out, n.value)
    lib().cso_free(out)
    return data


# ---------------------------------------------------------------- libwebp's lossy encoder restated (vp8enc_oracle.c; pinned to libwebp itself)
def _mbs_array(mbs, n):
    a = np.frombuffer(mbs, dtype=np.dtype([("segment", "u1"), ("is_i4", "u1"), ("ymode", "u1"), ("uvmode", "u1"), ("bmodes", "u1", 16), ("skip", "u1"), ("alpha", "u1"), ("pad", "u1", 2), ("levels", "<i2", (25, 16))]), count=n)
    return a.copy()


def vp8enc_encode_yuv(y, u, v, width, height, quality, trace=False):
    """-> bytes [, Vp8Frame, per-macroblock structured array] for padded planes (webp_rgb_to_yuv)"""
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    nmb = ((width + 15) // 16) * ((height + 15) // 16)
    frame, mbs = Vp8Frame(), (Vp8Mb * nmb)()
    rc = lib().cso_vp8enc_encode_yuv(y.ctypes.data, u.ctypes.data, v.ctypes.data, width, height, float(quality), C.byref(out), C.byref(n), C.byref(frame), C.byref(mbs))
    if rc:
        raise OracleError("vp8enc oracle: %d" % rc)
    data = C.string_at(out, n.value)
    lib().cso_free(out)
    return (data, frame, _mbs_array(mbs, nmb)) if trace else data


def vp8enc_encode_rgb(rgb, quality, trace=False):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    y, u, v = webp_rgb_to_yuv(rgb)
    return vp8enc_encode_yuv(y, u, v, w, h, quality, trace)


def vp8_parse(data):
    """-> Vp8Frame, per-macroblock structured array of any lossy WebP / VP8 key frame"""
    frame = Vp8Frame()
    cap = 1 << 20
    w = h = 0
    off = 20 if data[:4] == b"RIFF" else 0
    w, h = int.from_bytes(data[off + 6:off + 8], "little") & 0x3fff, int.from_bytes(data[off + 8:off + 10], "little") & 0x3fff
    cap = ((w + 15) // 16) * ((h + 15) // 16)
    mbs = (Vp8Mb * cap)()
    rc = lib().cso_vp8_parse(data, len(data), C.byref(frame), C.byref(mbs), cap)
    if rc:
        raise OracleError("vp8 parse: %d" % rc)
    return frame, _mbs_array(mbs, cap)


def png_to_webp(data, quality):
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t()
    rc = lib().cso_png_to_webp(data, len(data), quality, C.byref(out), C.byref(n))
    if rc:
        raise PngError(rc)
    res = C.string_at(out, n.value)
    lib().cso_free(out)
    return res


def vp8l_encode(pixels, width, height, channels):
    """lossless WebP (VP8L) of `pixels` (bytes: width * height * channels; 1 grey, 2 grey + alpha, 3 RGB, 4 RGBA) as the device coder writes it
    (oracle/png_oracle.c cso_vp8l_encode)"""
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t()
    rc = lib().cso_vp8l_encode(bytes(pixels), width, height, channels, C.byref(out), C.byref(n))
    if rc:
        raise PngError(rc)
    res = C.string_at(out, n.value)
    lib().cso_free(out)
    return res


def png_lossy(data, level=3, keep_metadata=False, quality=80):
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    rc = lib().cso_png_lossy(data, len(data), level, 1 if keep_metadata else 0, quality, C.byref(out), C.byref(n))
    if rc:
        raise PngError(rc)
    res = C.string_at(out, n.value)
    lib().cso_free(out)
    return res
