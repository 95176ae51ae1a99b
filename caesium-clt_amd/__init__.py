"""caesium-clt_amd -- MI355X-native replacement for caesium-clt's per-image hot path.

The product is the C-ABI shared library `libcaesium_hip.so` (sources in csrc/, interface in
include/caesium_hip.h).  This package is only the thin ctypes loader the tests, bench.py and
__graft_entry__ use; it has no compute of its own and no CPU fallback: loading fails loudly if the
HIP library has not been built.
"""
from .binding import (CByteArray, CCSParameters, CCSResult, CaesiumHip, CaesiumError, Timing, default_parameters,  # noqa: F401
                      library_path, load)
