"""The pixel kernels' block arithmetic, function by function, against the oracle's plain ISLOW routines: tests/xform_block_check.cpp compiles
caesium-clt_amd/csrc/k_pixel.hip as the emulation build does and drives the packed (v_dot2 form) forward transform, the multiply-add inverse
transform and the one-fma quantiser with range corners, deringing overshoot values and random blocks."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_functions_against_the_oracle():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "xcheck")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-DCSH_EMUL", "-Wno-attributes", "-Wno-unknown-pragmas", "-Wno-unused-function", "-x", "c++",
                               os.path.join(ROOT, "tests", "xform_block_check.cpp"), "-x", "c", os.path.join(ROOT, "oracle", "jpeg_oracle.c"),
                               "-I", os.path.join(ROOT, "oracle"), "-o", exe, "-lm", "-lpthread"])
        out = subprocess.run([exe], capture_output=True, timeout=600)
    assert out.returncode == 0 and out.stdout.decode().strip().endswith("bad=0"), out.stdout.decode()[-2000:] + out.stderr.decode()[-500:]
