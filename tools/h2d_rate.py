#!/usr/bin/env python3
"""Host -> device and device -> host rate of a 2 GiB pageable buffer (what invertQuda's source and solution are at 48^3 x 96): one hipMemcpy against
the same bytes cut into chunks copied by several host threads, and against a pinned (hipHostMalloc) buffer."""
import ctypes as C
import threading
import time

import numpy as np

hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
H2D, D2H = 1, 2
n = 2 << 30
host = np.ones(n // 8)
hp = host.ctypes.data
dev = C.c_void_p()
hip.hipMalloc(C.byref(dev), n)
hip.hipMemcpy(dev, hp, 1 << 20, H2D)


def timed(kind, nthreads):
    def work(k):
        off = k * (n // nthreads)
        if kind == H2D:
            hip.hipMemcpy(dev.value + off, hp + off, n // nthreads, H2D)
        else:
            hip.hipMemcpy(hp + off, dev.value + off, n // nthreads, D2H)
    best = 1e9
    for _ in range(3):
        ts = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        hip.hipDeviceSynchronize()
        best = min(best, time.perf_counter() - t0)
    return n / best * 1e-9


for nt in (1, 2, 4, 8):
    print("pageable, %d thread(s): H2D %.1f GB/s, D2H %.1f GB/s" % (nt, timed(H2D, nt), timed(D2H, nt)), flush=True)
pin = C.c_void_p()
hip.hipHostMalloc(C.byref(pin), n, 0)
t0 = time.perf_counter(); hip.hipMemcpy(dev, pin, n, H2D); hip.hipDeviceSynchronize(); t1 = time.perf_counter()
hip.hipMemcpy(pin, dev, n, D2H); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
print("pinned: H2D %.1f GB/s, D2H %.1f GB/s" % (n / (t1 - t0) * 1e-9, n / (t2 - t1) * 1e-9))
t0 = time.perf_counter(); C.memmove(pin, hp, n); t1 = time.perf_counter()
print("host memcpy pageable -> pinned, one thread: %.1f GB/s" % (n / (t1 - t0) * 1e-9))
