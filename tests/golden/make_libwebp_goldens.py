"""Writes tests/golden/libwebp_vp8enc.json: SHA-256 and length of what libwebp's WebPEncode (default WebPConfig at the given quality: the reference's
call, /root/reference/src/compressor.rs:417, :429 -> crate webp 0.3.1) makes of the pictures tests/test_oracle_webp.py builds, for every libwebp this container
carries (1.2.0, 1.2.2, 1.6.0: they agree byte for byte, which the script asserts).  A digest is a known-answer vector: the oracle (oracle/vp8enc_oracle.c) must
reproduce it without libwebp being present.  `python tests/golden/make_libwebp_goldens.py`"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
from libwebp_pin import libwebp_encode, libwebps   # noqa: E402
from test_oracle_webp import pictures             # noqa: E402

libs = libwebps()
assert libs, "no libwebp with the encoder API here"
out = {"libwebp_versions": [v for v, _ in libs], "cases": {}}
for name, rgb, q in pictures(big=True):
    datas = [libwebp_encode(W, rgb, q) for _, W in libs]
    assert all(d == datas[0] for d in datas), name
    assert libwebp_encode(libs[0][1], rgb, q, use_argb=True) == datas[0], name   # the webp crate's picture set-up (use_argb = 1) imports alike
    out["cases"][name] = {"bytes": len(datas[0]), "sha256": hashlib.sha256(datas[0]).hexdigest()}
    print(name, len(datas[0]))
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "libwebp_vp8enc.json"), "w"), indent=1, sort_keys=True)
