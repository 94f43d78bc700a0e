#!/usr/bin/env python3
"""What a fresh device allocation costs on this box: hipMalloc / first touch (hipMemset) / second touch / hipFree for 1, 4 and 24 GiB."""
import ctypes as C
import time
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipDeviceSynchronize()
for gib in (1, 4, 24, 24):
    n = gib << 30
    p = C.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), n); hip.hipDeviceSynchronize(); t1 = time.perf_counter()
    hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
    hip.hipFree(p); hip.hipDeviceSynchronize(); t4 = time.perf_counter()
    print("ALLOC %2d GiB rc=%d: hipMalloc %.1f ms, first memset %.1f ms, second memset %.1f ms, hipFree %.1f ms" % (gib, rc, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)
