"""shared helpers for the tests: load the product library / the emulation build, reference outputs."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "caesium-clt_amd")


def package():
    """import caesium-clt_amd (the directory name is not a Python identifier) as caesium_clt_amd"""
    if "caesium_clt_amd" not in sys.modules:
        spec = importlib.util.spec_from_file_location("caesium_clt_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                      submodule_search_locations=[PKG_DIR])
        m = importlib.util.module_from_spec(spec)
        sys.modules["caesium_clt_amd"] = m
        spec.loader.exec_module(m)
    return sys.modules["caesium_clt_amd"]


def product_api():
    return package().load()


def emul_api():
    """CPU emulation build of the SAME kernel sources (logic tests only; see tests/emul/README.md)."""
    so = os.path.join(ROOT, "tests", "emul", "libcaesium_emul.so")
    srcs = [os.path.join(PKG_DIR, "csrc", f) for f in os.listdir(os.path.join(PKG_DIR, "csrc")) if f.endswith((".hip", ".cpp", ".h", ".hpp"))]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(PKG_DIR, "csrc"), "emul"])
    return package().CaesiumHip(so)


def oracle_lossy(src, quality=80, progressive=1, subsampling=420, keep_metadata=0, preserve_icc=1):
    from oracle import oracle as O
    return O.jpeg_compress(src, O.params(quality=quality, progressive=progressive, subsampling=subsampling, qtable_profile=3, marker_style=1,
                                         keep_metadata=keep_metadata, preserve_icc=preserve_icc))


def oracle_lossless(src, progressive=1, keep_metadata=0, preserve_icc=1):
    from oracle import oracle as O
    return O.jpeg_compress(src, O.params(progressive=progressive, marker_style=1, keep_metadata=keep_metadata, preserve_icc=preserve_icc), lossless=True)


def oracle_resized(src, width, height, quality=80, subsampling=420):
    from oracle import oracle as O
    return O.jpeg_compress_resized(src, O.params(quality=quality, progressive=1, subsampling=subsampling, qtable_profile=3, marker_style=1), width, height)
