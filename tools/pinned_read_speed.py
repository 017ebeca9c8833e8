#!/usr/bin/env python3
"""is page-locked host memory (bsx_pinned_alloc = hipHostMalloc) as fast to READ from the CPU as ordinary memory?  (the command line's
formatter reads the reads it formats from the page-locked upload buffers)"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bsmap_amd as B
L = B.lib()
L.bsx_pinned_alloc.restype = C.c_void_p; L.bsx_pinned_alloc.argtypes = [C.c_size_t]; L.bsx_pinned_free.argtypes = [C.c_void_p]
L.bsx_thread_device(0)
n = 1 << 28
p = L.bsx_pinned_alloc(n)
pin = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p))
ordinary = np.zeros(n, np.uint8)
pin[:] = 1; ordinary[:] = 1
dst = np.empty(n, np.uint8)
out = {}
for name, src in (("ordinary", ordinary), ("pinned", pin)):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); np.copyto(dst, src); ts.append(time.perf_counter() - t0)
    out[name + "_read_GBps"] = n / min(ts) / 1e9
    t0 = time.perf_counter(); s = int(src[::64].sum()); out[name + "_strided_line_touch_ns"] = (time.perf_counter() - t0) / (n / 64) * 1e9
print(json.dumps(out))
L.bsx_pinned_free(p)
