"""What a cold process pays for device and pinned memory: hipMalloc / hipHostMalloc / hipFree by size (ctypes on libamdhip64, no torch)."""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
t0 = time.time(); n = C.c_int(); hip.hipGetDeviceCount(C.byref(n)); hip.hipSetDevice(0); p = C.c_void_p(); hip.hipMalloc(C.byref(p), 4096); print(f"runtime start + first hipMalloc: {(time.time() - t0) * 1e3:.1f} ms")
for mb in (1, 64, 1024, 4096, 4096, 16384):
    t0 = time.time(); q = C.c_void_p(); rc = hip.hipMalloc(C.byref(q), mb << 20); t1 = time.time()
    hip.hipMemset(q, 0, mb << 20); hip.hipDeviceSynchronize(); t2 = time.time()
    hip.hipFree(q); t3 = time.time()
    print(f"hipMalloc {mb:6d} MiB: rc {rc} {(t1 - t0) * 1e3:8.2f} ms, first memset {(t2 - t1) * 1e3:8.2f} ms, hipFree {(t3 - t2) * 1e3:8.2f} ms")
t0 = time.time()
ptrs = []
for i in range(80):
    q = C.c_void_p(); hip.hipMalloc(C.byref(q), 80 << 20); ptrs.append(q)
print(f"80 x hipMalloc 80 MiB: {(time.time() - t0) * 1e3:.1f} ms")
t0 = time.time()
for q in ptrs: hip.hipFree(q)
print(f"80 x hipFree: {(time.time() - t0) * 1e3:.1f} ms")
for mb in (16, 256, 1024):
    t0 = time.time(); q = C.c_void_p(); rc = hip.hipHostMalloc(C.byref(q), mb << 20, 0); t1 = time.time(); hip.hipHostFree(q); t2 = time.time()
    print(f"hipHostMalloc {mb:5d} MiB: rc {rc} {(t1 - t0) * 1e3:8.2f} ms, hipHostFree {(t2 - t1) * 1e3:8.2f} ms")
