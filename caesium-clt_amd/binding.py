"""ctypes mirror of include/caesium_hip.h (same names, same argument meaning)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
NPHASES = 8
NKERNELS = 36
PHASE_NAMES = ["decode", "pixel", "masks_flags_runs", "stats_tables", "sizes_scan", "pack", "stuff_assemble", "reserved"]


class CCSParameters(C.Structure):
    _fields_ = [("keep_metadata", C.c_bool), ("jpeg_quality", C.c_uint32), ("jpeg_chroma_subsampling", C.c_uint32),
                ("jpeg_progressive", C.c_bool), ("jpeg_optimize", C.c_bool), ("jpeg_preserve_icc", C.c_bool),
                ("png_quality", C.c_uint32), ("png_optimization_level", C.c_uint32), ("png_force_zopfli", C.c_bool),
                ("png_optimize", C.c_bool), ("gif_quality", C.c_uint32), ("webp_quality", C.c_uint32), ("webp_lossless", C.c_bool),
                ("tiff_compression", C.c_uint32), ("tiff_deflate_level", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


class CCSResult(C.Structure):
    _fields_ = [("success", C.c_bool), ("code", C.c_uint32), ("error_message", C.c_char_p)]


class CByteArray(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("length", C.c_size_t)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("phase_ms", C.c_float * NPHASES), ("kernel_ms", C.c_float * NKERNELS), ("in_bytes", C.c_uint64), ("out_bytes", C.c_uint64),
                ("pixels", C.c_uint64), ("coef_bytes", C.c_uint64), ("n_images", C.c_uint32), ("n_failed", C.c_uint32), ("n_seq_decoded", C.c_uint32), ("n_par_fallback", C.c_uint32), ("n_par_short", C.c_uint32), ("n_prog_decoded", C.c_uint32), ("n_refine_chains", C.c_uint32), ("n_search_extra", C.c_uint32)]


class PngTiming(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("kernel_ms", C.c_float * 16), ("in_bytes", C.c_uint64), ("out_bytes", C.c_uint64), ("pixels", C.c_uint64),
                ("raw_bytes", C.c_uint64), ("n_images", C.c_uint32), ("n_failed", C.c_uint32), ("n_trials", C.c_uint32)]


class CaesiumError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


def library_path():
    return os.path.join(_HERE, "libcaesium_hip.so")


EXPORTS = ["cs_default_parameters", "cs_compress_in_memory", "cs_compress_to_size_in_memory", "cs_convert_in_memory",
           "cs_batch_compress", "cs_batch_extent", "cs_free_bytes", "cs_free_result", "csh_device_count", "csh_last_error", "csh_kernel_name", "csh_kernel_name_webp", "csh_release_cached_memory", "csh_warmup", "csh_batch_create",
           "csh_batch_run", "csh_batch_fetch", "csh_batch_destroy", "csh_batch_retain_dct", "csh_batch_set_quality", "csh_batch_rerun_encode", "cs_batch_compress_to_size", "csh_batch_geometry", "csh_batch_read_coefs",
           "csp_kernel_name", "csp_batch_create", "csp_batch_create_webp", "csp_batch_create_pixels", "csh_batch_create_pixels", "csh_batch_pixels", "csh_batch_create_from_pixels", "csp_png_to_jpeg", "csp_png_to_lossless_webp", "csp_batch_run", "csp_batch_fetch", "csp_batch_destroy", "csp_batch_geometry", "csp_batch_read_rows", "csp_batch_read_stream",
           "csp_batch_trials", "csp_batch_read_scores", "csp_batch_chunk_bits", "csh_batch_create_webp", "cs_batch_convert",
           "cswd_batch_create", "cswd_batch_run", "cswd_batch_pixels", "cswd_batch_read_pixels", "cswd_batch_alpha", "cswd_batch_read_rgba", "cswd_batch_destroy", "cswd_rgba_join", "cswd_rgba_destroy", "csh_batch_create_webp_from_pixels", "csh_batch_create_from_pixels_rgb", "csl_encode_pixels", "csl_attach_alpha"]


def _declare(L):
    P = C.POINTER
    L.cs_default_parameters.argtypes = [P(CCSParameters)]
    L.cs_default_parameters.restype = None
    L.cs_compress_in_memory.argtypes = [C.c_char_p, C.c_size_t, P(CCSParameters), P(CByteArray)]
    L.cs_compress_in_memory.restype = CCSResult
    L.cs_compress_to_size_in_memory.argtypes = [C.c_char_p, C.c_size_t, P(CCSParameters), C.c_size_t, C.c_bool, P(CByteArray)]
    L.cs_compress_to_size_in_memory.restype = CCSResult
    L.cs_convert_in_memory.argtypes = [C.c_char_p, C.c_size_t, P(CCSParameters), C.c_uint32, P(CByteArray)]
    L.cs_convert_in_memory.restype = CCSResult
    L.cs_batch_compress.argtypes = [P(CByteArray), C.c_size_t, P(CCSParameters), C.c_int, P(CByteArray), P(CCSResult)]
    L.cs_batch_extent.argtypes = [P(CByteArray), C.c_size_t]
    L.cs_batch_extent.restype = C.c_size_t
    L.cs_free_bytes.argtypes = [P(CByteArray)]
    L.cs_free_bytes.restype = None
    L.cs_free_result.argtypes = [P(CCSResult)]
    L.cs_free_result.restype = None
    L.csh_last_error.restype = C.c_char_p
    L.csh_kernel_name.argtypes = [C.c_int]
    L.csh_kernel_name.restype = C.c_char_p
    L.csh_kernel_name_webp.argtypes = [C.c_int]
    L.csh_kernel_name_webp.restype = C.c_char_p
    L.csh_batch_create.argtypes = [P(CByteArray), C.c_size_t, P(CCSParameters), C.c_int, P(C.c_void_p)]
    L.csh_batch_run.argtypes = [C.c_void_p, P(Timing)]
    L.csh_batch_fetch.argtypes = [C.c_void_p, P(CByteArray), P(CCSResult)]
    L.csh_batch_retain_dct.argtypes = [C.c_void_p, C.c_int]
    L.csh_batch_set_quality.argtypes = [C.c_void_p, P(C.c_uint32)]
    L.csh_batch_rerun_encode.argtypes = [C.c_void_p, P(Timing)]
    L.cs_batch_compress_to_size.argtypes = [P(CByteArray), C.c_size_t, P(CCSParameters), C.c_size_t, C.c_bool, C.c_int, P(CByteArray), P(CCSResult)]
    L.csh_batch_destroy.argtypes = [C.c_void_p]
    L.csh_batch_destroy.restype = None
    L.csh_batch_geometry.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int)]
    L.csh_batch_read_coefs.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    L.csh_batch_create_webp.argtypes = [P(CByteArray), C.c_size_t, P(CCSParameters), C.c_int, P(C.c_void_p)]
    L.cs_batch_convert.argtypes = [P(CByteArray), C.c_size_t, P(CCSParameters), C.c_uint32, C.c_int, P(CByteArray), P(CCSResult)]
    L.csp_kernel_name.argtypes = [C.c_int]
    L.csp_kernel_name.restype = C.c_char_p
    L.csp_batch_create.argtypes = [P(CByteArray), C.c_size_t, P(CCSParameters), C.c_int, P(C.c_void_p)]
    L.csp_batch_run.argtypes = [C.c_void_p, P(PngTiming)]
    L.csp_batch_fetch.argtypes = [C.c_void_p, P(CByteArray), P(CCSResult)]
    L.csp_batch_destroy.argtypes = [C.c_void_p]
    L.csp_batch_destroy.restype = None
    L.csp_batch_geometry.argtypes = [C.c_void_p, C.c_size_t, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
    L.csp_batch_read_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.csp_batch_read_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.csp_batch_read_scores.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, P(C.c_int)]
    L.csp_batch_chunk_bits.argtypes = [C.c_void_p, C.c_size_t, C.c_int, P(C.c_uint64), C.c_size_t, P(C.c_size_t)]
    L.csp_batch_trials.argtypes = [C.c_void_p, C.c_size_t, P(C.c_int), P(C.c_uint64), P(C.c_int), P(C.c_int)]
    L.cswd_batch_create.argtypes = [P(CByteArray), C.c_size_t, C.c_int, P(C.c_void_p)]
    L.cswd_batch_run.argtypes = [C.c_void_p]
    L.cswd_batch_pixels.argtypes = [C.c_void_p, C.c_size_t, P(C.c_void_p), P(C.c_uint32), P(C.c_uint32), P(C.c_uint32), P(C.c_char_p)]
    L.cswd_batch_read_pixels.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.cswd_batch_alpha.argtypes = [C.c_void_p, C.c_size_t, P(C.c_void_p), P(C.c_void_p)]
    L.cswd_batch_read_rgba.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.cswd_batch_destroy.argtypes = [C.c_void_p]
    L.cswd_batch_destroy.restype = None
    L.cswd_rgba_join.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.cswd_rgba_join.restype = C.c_int
    L.cswd_rgba_destroy.argtypes = [C.c_void_p]
    L.cswd_rgba_destroy.restype = None
    return L


def default_parameters(lib=None, **kw):
    p = CCSParameters()
    p.jpeg_quality = p.png_quality = p.webp_quality = p.gif_quality = 80
    p.jpeg_progressive = True
    p.jpeg_preserve_icc = True
    p.png_optimization_level = 3
    p.tiff_deflate_level = 6
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class Batch:
    """csh_batch: a group of input files resident in HBM."""

    def __init__(self, api, blobs, params, device=0, webp=False):
        self.api = api
        L = api.L
        self.n = len(blobs)
        self._keep = [C.create_string_buffer(b, len(b)) for b in blobs]
        self._in = (CByteArray * self.n)()
        for i, buf in enumerate(self._keep):
            self._in[i].data = C.cast(buf, C.POINTER(C.c_uint8))
            self._in[i].length = len(blobs[i])
        self.h = C.c_void_p()
        rc = (L.csh_batch_create_webp if webp else L.csh_batch_create)(self._in, self.n, C.byref(params), device, C.byref(self.h))
        if rc:
            raise CaesiumError(rc, L.csh_last_error().decode())

    def run(self):
        t = Timing()
        rc = self.api.L.csh_batch_run(self.h, C.byref(t))
        if rc:
            raise CaesiumError(rc, self.api.L.csh_last_error().decode())
        return t

    def retain_dct(self, on=True):
        if self.api.L.csh_batch_retain_dct(self.h, 1 if on else 0):
            raise CaesiumError(-1, self.api.L.csh_last_error().decode())

    def set_quality(self, qualities):
        arr = (C.c_uint32 * self.n)(*qualities)
        if self.api.L.csh_batch_set_quality(self.h, arr):
            raise CaesiumError(-1, self.api.L.csh_last_error().decode())

    def rerun_encode(self):
        t = Timing()
        if self.api.L.csh_batch_rerun_encode(self.h, C.byref(t)):
            raise CaesiumError(-1, self.api.L.csh_last_error().decode())
        return t

    def fetch(self):
        """-> list of bytes (or CaesiumError instances for failed items), input order."""
        L = self.api.L
        outs = (CByteArray * self.n)()
        res = (CCSResult * self.n)()
        rc = L.csh_batch_fetch(self.h, outs, res)
        if rc < 0:
            raise CaesiumError(rc, L.csh_last_error().decode())
        result = []
        for i in range(self.n):
            if res[i].success:
                result.append(C.string_at(outs[i].data, outs[i].length))
            else:
                result.append(CaesiumError(res[i].code, (res[i].error_message or b"").decode()))
            L.cs_free_bytes(C.byref(outs[i]))
            L.cs_free_result(C.byref(res[i]))
        return result

    def coefs(self, image, comp, which):
        """[bh][bw][64] int16, zig-zag order.  which: 0 decoded, 1 re-quantised."""
        import numpy as np
        L = self.api.L
        bw, bh, rbw, rbh = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        if L.csh_batch_geometry(self.h, image, comp, which, C.byref(bw), C.byref(bh), C.byref(rbw), C.byref(rbh)):
            raise CaesiumError(-1, L.csh_last_error().decode())
        a = np.empty((bh.value, bw.value, 64), dtype=np.int16)
        if L.csh_batch_read_coefs(self.h, image, comp, which, a.ctypes.data):
            raise CaesiumError(-1, L.csh_last_error().decode())
        return a, (rbw.value, rbh.value)

    def close(self):
        if self.h:
            self.api.L.csh_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PngBatch:
    """csp_batch: a group of PNG files resident in HBM (the lossless PNG row)."""

    def __init__(self, api, blobs, params, device=0):
        self.api = api
        L = api.L
        self.n = len(blobs)
        self._keep = [C.create_string_buffer(b, len(b)) for b in blobs]
        self._in = (CByteArray * self.n)()
        for i, buf in enumerate(self._keep):
            self._in[i].data = C.cast(buf, C.POINTER(C.c_uint8))
            self._in[i].length = len(blobs[i])
        self.h = C.c_void_p()
        rc = L.csp_batch_create(self._in, self.n, C.byref(params), device, C.byref(self.h))
        if rc:
            raise CaesiumError(rc, L.csh_last_error().decode())

    def run(self):
        t = PngTiming()
        rc = self.api.L.csp_batch_run(self.h, C.byref(t))
        if rc:
            raise CaesiumError(rc, self.api.L.csh_last_error().decode())
        return t

    def fetch(self):
        L = self.api.L
        outs = (CByteArray * self.n)()
        res = (CCSResult * self.n)()
        if L.csp_batch_fetch(self.h, outs, res) < 0:
            raise CaesiumError(-1, L.csh_last_error().decode())
        result = []
        for i in range(self.n):
            result.append(C.string_at(outs[i].data, outs[i].length) if res[i].success else CaesiumError(res[i].code, (res[i].error_message or b"").decode()))
            L.cs_free_bytes(C.byref(outs[i])); L.cs_free_result(C.byref(res[i]))
        return result

    def geometry(self, image):
        w, h, rb = C.c_uint32(), C.c_uint32(), C.c_uint32()
        if self.api.L.csp_batch_geometry(self.h, image, C.byref(w), C.byref(h), C.byref(rb)):
            raise CaesiumError(-1, self.api.L.csh_last_error().decode())
        return w.value, h.value, rb.value

    def rows(self, image):
        import numpy as np
        w, h, rb = self.geometry(image)
        out = np.empty((h, rb), dtype=np.uint8)
        if self.api.L.csp_batch_read_rows(self.h, image, out.ctypes.data):
            raise CaesiumError(-1, self.api.L.csh_last_error().decode())
        return out

    def stream(self, image, strategy):
        import numpy as np
        w, h, rb = self.geometry(image)
        out = np.empty(h * (rb + 1), dtype=np.uint8)
        if self.api.L.csp_batch_read_stream(self.h, image, strategy, out.ctypes.data):
            raise CaesiumError(-1, self.api.L.csh_last_error().decode())
        return out

    def scores(self, image):
        """-> ([height][5][5] scores, bit mask of the score columns this plan computed)"""
        import numpy as np
        w, h, rb = self.geometry(image)
        out = np.zeros((h, 5, 5), dtype=np.uint64)
        have = C.c_int()
        if self.api.L.csp_batch_read_scores(self.h, image, out.ctypes.data, C.byref(have)):
            raise CaesiumError(-1, self.api.L.csh_last_error().decode())
        return out, have.value

    def chunk_bits(self, image, trial):
        arr = (C.c_uint64 * 65536)(); n = C.c_size_t()
        if self.api.L.csp_batch_chunk_bits(self.h, image, trial, arr, 65536, C.byref(n)):
            raise CaesiumError(-1, self.api.L.csh_last_error().decode())
        return [arr[i] for i in range(n.value)]

    def trials(self, image):
        """-> ([(strategy, zlib bytes)], index of the winner)"""
        st = (C.c_int * 10)(); zb = (C.c_uint64 * 10)(); n = C.c_int(); w = C.c_int()
        if self.api.L.csp_batch_trials(self.h, image, st, zb, C.byref(n), C.byref(w)):
            raise CaesiumError(-1, self.api.L.csh_last_error().decode())
        return [(st[i], zb[i]) for i in range(n.value)], w.value

    def close(self):
        if getattr(self, "h", None):
            self.api.L.csp_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CaesiumHip:
    def __init__(self, path=None):
        path = path or library_path()
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
        self.path = path
        self.L = _declare(C.CDLL(path))

    def kernel_names(self):
        return [self.L.csh_kernel_name(i).decode() for i in range(NKERNELS)]

    def release_cached_memory(self):
        self.L.csh_release_cached_memory()

    def webp_kernel_names(self):
        return [self.L.csh_kernel_name_webp(i).decode() for i in range(NKERNELS)]

    def batch_extent(self, sizes_and_heads):
        """cs_batch_extent over inputs described as (declared length, header bytes): the header bytes are what the probe reads; the
        declared length may exceed them (the probe never reads past the header of a well-formed file)"""
        n = len(sizes_and_heads)
        keep = [C.create_string_buffer(h, len(h)) for _, h in sizes_and_heads]
        ins = (CByteArray * n)()
        for i, (length, _) in enumerate(sizes_and_heads):
            ins[i].data = C.cast(keep[i], C.POINTER(C.c_uint8)); ins[i].length = length
        return self.L.cs_batch_extent(ins, n)

    def device_count(self):
        return self.L.csh_device_count()

    def _call(self, fn, *args):
        out = CByteArray()
        r = fn(*args, C.byref(out))
        try:
            if not r.success:
                raise CaesiumError(r.code, (r.error_message or b"").decode())
            return C.string_at(out.data, out.length)
        finally:
            self.L.cs_free_bytes(C.byref(out))
            self.L.cs_free_result(C.byref(r))

    def compress_in_memory(self, data, params):
        return self._call(self.L.cs_compress_in_memory, data, len(data), C.byref(params))

    def compress_to_size_in_memory(self, data, params, max_output_size, return_smallest=True):
        return self._call(self.L.cs_compress_to_size_in_memory, data, len(data), C.byref(params), max_output_size, return_smallest)

    def convert_in_memory(self, data, params, fmt):
        return self._call(self.L.cs_convert_in_memory, data, len(data), C.byref(params), fmt)

    def batch_compress_to_size(self, blobs, params, max_output_size, return_smallest=True, device=0):
        n = len(blobs)
        keep = [C.create_string_buffer(x, len(x)) for x in blobs]
        ins = (CByteArray * n)()
        for i, buf in enumerate(keep):
            ins[i].data = C.cast(buf, C.POINTER(C.c_uint8)); ins[i].length = len(blobs[i])
        outs = (CByteArray * n)(); res = (CCSResult * n)()
        self.L.cs_batch_compress_to_size(ins, n, C.byref(params), max_output_size, return_smallest, device, outs, res)
        result = []
        for i in range(n):
            result.append(C.string_at(outs[i].data, outs[i].length) if res[i].success else CaesiumError(res[i].code, (res[i].error_message or b"").decode()))
            self.L.cs_free_bytes(C.byref(outs[i])); self.L.cs_free_result(C.byref(res[i]))
        return result

    def batch(self, blobs, params, device=0):
        return Batch(self, blobs, params, device)

    def webp_decode(self, blobs, device=0):
        """cswd_batch: WebP files -> [H][W][3] uint8 arrays ([H][W][4] for a picture with transparency; CaesiumError per file), decoded on the device"""
        import numpy as np
        L = self.L
        n = len(blobs)
        keep = [C.create_string_buffer(x, len(x)) for x in blobs]
        ins = (CByteArray * n)()
        for i, buf in enumerate(keep):
            ins[i].data = C.cast(buf, C.POINTER(C.c_uint8)); ins[i].length = len(blobs[i])
        h = C.c_void_p()
        rc = L.cswd_batch_create(ins, n, device, C.byref(h))
        if rc:
            raise CaesiumError(rc, L.csh_last_error().decode())
        try:
            rc = L.cswd_batch_run(h)
            if rc:
                raise CaesiumError(rc, L.csh_last_error().decode())
            out = []
            for i in range(n):
                p = C.c_void_p(); w = C.c_uint32(); hh = C.c_uint32(); ch = C.c_uint32(); msg = C.c_char_p()
                rc = L.cswd_batch_pixels(h, i, C.byref(p), C.byref(w), C.byref(hh), C.byref(ch), C.byref(msg))
                if rc:
                    out.append(CaesiumError(rc, (msg.value or b"").decode()))
                    continue
                rgba, plane = C.c_void_p(), C.c_void_p()
                L.cswd_batch_alpha(h, i, C.byref(rgba), C.byref(plane))
                a = np.empty((hh.value, w.value, 4 if rgba.value else 3), dtype=np.uint8)   # a picture with transparency comes back as RGBA
                if (L.cswd_batch_read_rgba if rgba.value else L.cswd_batch_read_pixels)(h, i, a.ctypes.data):
                    raise CaesiumError(-1, L.csh_last_error().decode())
                out.append(a)
            return out
        finally:
            L.cswd_batch_destroy(h)

    def webp_batch(self, blobs, params, device=0):
        """JPEG in, WebP out: the same batch object with the VP8 encoder as its tail"""
        return Batch(self, blobs, params, device, webp=True)

    def batch_convert(self, blobs, params, fmt, device=0):
        """cs_batch_convert: -> list of bytes / CaesiumError, input order"""
        n = len(blobs)
        keep = [C.create_string_buffer(x, len(x)) for x in blobs]
        ins = (CByteArray * n)()
        for i, buf in enumerate(keep):
            ins[i].data = C.cast(buf, C.POINTER(C.c_uint8)); ins[i].length = len(blobs[i])
        outs = (CByteArray * n)(); res = (CCSResult * n)()
        self.L.cs_batch_convert(ins, n, C.byref(params), fmt, device, outs, res)
        result = []
        for i in range(n):
            result.append(C.string_at(outs[i].data, outs[i].length) if res[i].success else CaesiumError(res[i].code, (res[i].error_message or b"").decode()))
            self.L.cs_free_bytes(C.byref(outs[i])); self.L.cs_free_result(C.byref(res[i]))
        return result

    def png_batch(self, blobs, params, device=0):
        return PngBatch(self, blobs, params, device)

    def png_kernel_names(self):
        return [self.L.csp_kernel_name(i).decode() for i in range(16)]

    def cs_batch_compress(self, blobs, params, device=0, timing=None):
        """the C entry point itself: mixed inputs, routed per file type.  timing: a list that receives the seconds spent inside the C call
        (host buffers in -> host buffers out: parse, upload, kernels, download), without this wrapper's own copies"""
        import time
        n = len(blobs)
        keep = [C.create_string_buffer(x, len(x)) for x in blobs]
        ins = (CByteArray * n)()
        for i, buf in enumerate(keep):
            ins[i].data = C.cast(buf, C.POINTER(C.c_uint8)); ins[i].length = len(blobs[i])
        outs = (CByteArray * n)(); res = (CCSResult * n)()
        t0 = time.perf_counter()
        self.L.cs_batch_compress(ins, n, C.byref(params), device, outs, res)
        if timing is not None:
            timing.append(time.perf_counter() - t0)
        result = []
        for i in range(n):
            result.append(C.string_at(outs[i].data, outs[i].length) if res[i].success else CaesiumError(res[i].code, (res[i].error_message or b"").decode()))
            self.L.cs_free_bytes(C.byref(outs[i])); self.L.cs_free_result(C.byref(res[i]))
        return result

    def batch_compress(self, blobs, params, device=0):
        b = self.batch(blobs, params, device)
        try:
            b.run()
            return b.fetch()
        finally:
            b.close()


_api = None


def load():
    global _api
    if _api is None:
        _api = CaesiumHip()
    return _api
